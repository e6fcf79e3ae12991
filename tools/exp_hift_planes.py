"""Round 6: the vocoder's plane products -- three bf16 planes per operand (six products, the default) against TWO planes (three products,
cv2_hift_debug_precision(2) / CV2_HIFT_PLANES=2) and the fp32 matrix-core kernels: ms per call and the waveform distance to the three-plane
result.  python tools/exp_hift_planes.py [frames ...]   (accuracy against the reference fixtures: tests/test_hift_gpu.py::test_precision_modes)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cosyvoice2-eu_amd'))
import torch
from cv2amd import synth, lib as L
from cv2amd.hift import HiftEngine
frames = [int(a) for a in sys.argv[1:]] or [500, 250, 90]
eng = HiftEngine(synth.make_hift(), 'cuda:0', max_frames=max(512, max(frames)))
lib = L.lib()
for T in frames:
    g = torch.Generator().manual_seed(T)
    mel = (torch.randn(1, 80, T, generator=g) * 2 - 4).clamp(-11.5, 2).to('cuda:0')
    nz = torch.randn(1, 480 * T, 9, generator=g).to('cuda:0')          # (injected noise: no graph replay, the mode of THIS call counts)
    ref = None
    for mode, name in ((0, 'three planes'), (2, 'two planes'), (1, 'fp32 MFMA'), (0, 'three planes'), (2, 'two planes')):
        L.check(lib.cv2_hift_debug_precision(mode))
        for _ in range(3):
            wav, src = eng.inference(mel, None, noise=nz)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            wav, src = eng.inference(mel, None, noise=nz)
        e1.record()
        torch.cuda.synchronize()
        if ref is None:
            ref = wav.clone()
        print(f'{T:4d} frames, {name:12s}: {e0.elapsed_time(e1) / 20:7.3f} ms per call; max |wav - three-plane wav| = {float((wav - ref).abs().max()):.3e} '
              f'(absmax {float(ref.abs().max()):.3f})', flush=True)
L.check(lib.cv2_hift_debug_precision(-1))
