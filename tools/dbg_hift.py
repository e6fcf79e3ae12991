import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cosyvoice2-eu_amd'))
import torch, torch.nn.functional as F
from cv2amd import synth
from cv2amd.hift import HiftEngine
from oracle import hift as OH
sd = synth.make_hift()
eng = HiftEngine(sd, 'cuda:0', max_frames=256)
eng.lib.cv2_hift_debug_buffer.restype = C.c_void_p
eng.lib.cv2_hift_debug_buffer.argtypes = [C.c_void_p, C.c_int32]
def buf(which, n):
    p = eng.lib.cv2_hift_debug_buffer(eng.handle, which)
    off = p - eng.workspace.data_ptr()
    return eng.workspace[off:off + 4 * n].view(torch.float32).cpu()
def noise(seed, T):
    g = torch.Generator().manual_seed(seed); return torch.rand(1, 9, generator=g), torch.randn(1, 480 * T, 9, generator=g)
T = 130
g = torch.Generator().manual_seed(5)
mel = (torch.randn(1, 80, T, generator=g) * 2 - 4).clamp(-11.5, 2)
cs = torch.zeros(1, 1, 0)
ri, nz = noise(77, T)
wav, src = eng.inference(mel.cuda(), cs, noise=nz)
torch.cuda.synchronize()
f0 = OH.f0_predictor(sd, mel)
print('f0 err', (buf(1, T) - f0[0]).abs().max().item(), 'f0 max', f0.max().item())
so = OH.source(sd, f0, ri, nz)
sr, si = OH.stft(so.squeeze(1)); sst = torch.cat([sr, si], 1)[0].t()
Fn = sst.shape[0]
e = (buf(2, Fn * 18).view(Fn, 18) - sst).abs().max(1).values
print('s_stft err', e.max().item(), 'at frame', e.argmax().item())
xpre = F.conv1d(mel, OH.wn(sd, 'conv_pre'), sd['conv_pre.bias'], padding=3)[0].t()
e = (buf(3, T * 512).view(T, 512) - xpre).abs().max(1).values
print('conv_pre err', e.max().item(), 'at', e.argmax().item())
pre = OH.decode(sd, mel, so, return_pre=True)[0].t()
e = (buf(5, Fn * 18).view(Fn, 18) - pre).abs().max(1).values
print('conv_post err', e.max().item(), 'at row', e.argmax().item(), 'rows>1e-4:', (e > 1e-4).nonzero().flatten()[:8].tolist(), (e > 1e-4).sum().item())
gp = buf(5, Fn * 18).view(Fn, 18).t()[None]
w_from_gpu_post = torch.clamp(OH.istft(torch.exp(gp[:, :9]), torch.sin(gp[:, 9:])), -0.99, 0.99)
wo = OH.decode(sd, mel, so)
print('wav err total', (wav.cpu() - wo).abs().max().item(), ' istft-only err (GPU post -> torch istft vs GPU wav):', (wav.cpu() - w_from_gpu_post).abs().max().item(),
      ' torch-istft(GPU post) vs oracle wav', (w_from_gpu_post - wo).abs().max().item())
e = (wav.cpu() - w_from_gpu_post).abs()[0]
print('istft err argmax', e.argmax().item(), 'count>1e-4', (e > 1e-4).sum().item())
i = e.argmax().item()
print('around', wav.cpu()[0, i-3:i+4], w_from_gpu_post[0, i-3:i+4])
fr = i // 4
print('post rows', gp[0, :, fr+2-2:fr+2+3].t())
