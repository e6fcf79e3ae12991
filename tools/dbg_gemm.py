"""Error pattern of cv2_gemm_bf16 against fp64 for one shape: python tools/dbg_gemm.py M N K"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cosyvoice2-eu_amd'))
import torch
from cv2amd import lib as L, weights as W, flow as F
lib = L.lib(); F._bind(lib)
m, n, k = (int(x) for x in sys.argv[1:4])
g = torch.Generator().manual_seed(0)
a = torch.randn(m, k, generator=g); w = torch.randn(n, k, generator=g) / k ** 0.5; b = torch.randn(n, generator=g)
ab = a.to(torch.bfloat16).cuda(); wp = W.pack_bf16(w.cuda()); bd = b.cuda()
out = torch.full((m, n), float('nan'), device='cuda')
L.check(lib.cv2_gemm_bf16(ab.data_ptr(), k, wp.data_ptr(), bd.data_ptr(), out.data_ptr(), n, m, n, k, L.stream_ptr()))
torch.cuda.synchronize()
ref = ab.cpu().double() @ W.bf16_round(w).double().T + b.double()
e = (out.cpu().double() - ref).abs()
e[~torch.isfinite(e)] = 1e9
print('max err', e.max().item())
bad = e > 1e-3
print('bad rows (mod 16):', sorted(set((bad.any(1).nonzero().flatten() % 16).tolist())))
print('bad row blocks:', sorted(set((bad.any(1).nonzero().flatten() // 16).tolist()))[:20])
print('bad cols (mod 16):', sorted(set((bad.any(0).nonzero().flatten() % 16).tolist())))
print('bad col tiles:', sorted(set((bad.any(0).nonzero().flatten() // 16).tolist())))
print(out[:2, :8].cpu(), ref[:2, :8])
