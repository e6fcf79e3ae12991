"""Micro-benchmark of the vocoder stage alone: python tools/bench_hift.py [frames] [reps] -- ms per call of HiftEngine.inference on a
[1, 80, frames] mel (500 frames = 10 s of audio), after warm-up."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cosyvoice2-eu_amd'))
import torch
from cv2amd import synth
from cv2amd.hift import HiftEngine
T = int(sys.argv[1]) if len(sys.argv) > 1 else 500
R = int(sys.argv[2]) if len(sys.argv) > 2 else 20
eng = HiftEngine(synth.make_hift(), 'cuda:0', max_frames=max(512, T))
mel = (torch.randn(1, 80, T, device='cuda:0') * 2 - 4).clamp(-11.5, 2)
for _ in range(3):
    eng.inference(mel, None, seed=1)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(R):
    eng.inference(mel, None, seed=1)
e1.record()
torch.cuda.synchronize()
print(f'HiFT {T} frames: {e0.elapsed_time(e1) / R:.3f} ms per call')
