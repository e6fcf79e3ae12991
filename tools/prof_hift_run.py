import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cosyvoice2-eu_amd'))
import torch
from cv2amd import synth
from cv2amd.hift import HiftEngine
eng = HiftEngine(synth.make_hift(), 'cuda:0', max_frames=512)
mel = torch.randn(1, 80, 500, device='cuda:0') * 0.5
for _ in range(3):
    eng.inference(mel, None, seed=1)
torch.cuda.synchronize()
