"""Phase timing inside the HiFT convolutions (diagnostic build only: ./build.sh -DCV2_STAMPS -o cosyvoice2-eu_amd/cv2amd/libcv2amd_dbg.so;
CV2_AMD_LIB=... python tools/dbg_stamps_hift.py).  Block (0,0,0) of every k_conv6 launch stamps s_memtime at: start, line buffer staged (last
chunk), MFMAs issued (last chunk), epilogue done; the last 56 launches of one 500-frame call grouped by (taps, dil, C_in, frames per block, C_out, blocks)."""
import ctypes as C, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cosyvoice2-eu_amd'))
import torch
from cv2amd import synth, lib as L
from cv2amd.hift import HiftEngine
eng = HiftEngine(synth.make_hift(), 'cuda:0', max_frames=512)
mel = (torch.randn(1, 80, 500, device='cuda:0') * 2 - 4).clamp(-11.5, 2)
for _ in range(2):
    eng.inference(mel, None, seed=1)
torch.cuda.synchronize()
buf = (C.c_ulonglong * (64 * 8))()
L.check(L.lib().cv2_debug_stamps_hift(buf))
groups = collections.defaultdict(list)
for k in range(56):
    t = [buf[k * 8 + i] for i in range(8)]
    if not t[0]: continue
    key = (t[6] >> 48, (t[6] >> 32) & 0xffff, t[6] & 0xffffffff, t[7] >> 48, (t[7] >> 32) & 0xffff, t[7] & 0xffffffff)
    groups[key].append([t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4], t[5]])
for key, rows in sorted(groups.items()):
    n = len(rows)
    avg = [sum(r[i] for r in rows) / n for i in range(5)]
    print(f'taps {key[0]:2d} dil {key[1]} C_in {key[2]:4d} frames/block {key[3]:3d} C_out {key[4]:4d} blocks {key[5]:5d} launches {n:2d}: to staged(last chunk) {avg[0]:7.0f}  MFMAs {avg[1]:7.0f}  epilogue {avg[2]:7.0f}  total {sum(avg[:3]):7.0f} cycles; staging over all chunks: load issue {avg[3]:6.0f}, wait + activation + planes {avg[4]:6.0f}')
