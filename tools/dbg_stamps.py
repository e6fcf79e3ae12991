"""Phase timing inside the skinny decode kernels (diagnostic build only):
    ./build.sh -DCV2_STAMPS -o cosyvoice2-eu_amd/cv2amd/libcv2amd_dbg.so
    CV2_AMD_LIB=$PWD/cosyvoice2-eu_amd/cv2amd/libcv2amd_dbg.so python tools/dbg_stamps.py
Prints, for layer 1's four weight-streaming kernels of the last decode step, s_memtime deltas (100 MHz ticks -> us) of
block 0 / wave 0 between the phase boundaries marked SK_STAMP in csrc/skinny.h."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cosyvoice2-eu_amd'))
import torch
from cv2amd import synth, lib as L
from cv2amd.llm import LLMEngine

sd = synth.make_llm(layers=24)
eng = LLMEngine(sd, 'cuda:0', max_seqs=1, max_pos=2048, max_out=2048)
inp = synth.synthetic_inputs(seed=0, text_len=50, prompt_len=255)
x = eng.build_lm_input(inp['text'], inp['prompt_text'], inp['prompt_token'])
eng.add_request(0, x, 5000, 5000, force_len=True)
names = ['qkv', 'o-proj', 'gate/up', 'down']
phases = ['issue loads', 'x arrive+rms', 'stage B', 'weights+mfma', 'reduce']
for rep in range(3):
    eng.step(1, 64)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (64 * 8))()
    L.check(L.lib().cv2_debug_stamps(buf))
    print('rep', rep)
    for k in range(4):
        t = [buf[k * 8 + i] for i in range(6)]
        d = [(t[i + 1] - t[i]) / 100.0 if t[i + 1] and t[i] else float('nan') for i in range(5)]
        print(f'  {names[k]:8s} ' + '  '.join(f'{p}={v:.2f}' for p, v in zip(phases, d)) + f'   total={(t[5]-t[0])/100.0:.2f} us')
    t0 = buf[0]; 
    print('  qkv start -> down end:', (buf[3 * 8 + 5] - t0) / 100.0, 'us;  kernel starts:', [(buf[k * 8] - t0) / 100.0 for k in range(4)])
