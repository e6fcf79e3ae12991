"""Timeline of the one-row chain launch (diagnostic build only):
    ./build.sh -DCV2_STAMPS -o cosyvoice2-eu_amd/cv2amd/libcv2amd_dbg.so
    CV2_AMD_LIB=$PWD/cosyvoice2-eu_amd/cv2amd/libcv2amd_dbg.so python tools/dbg_chain.py
Per role (O projection, gate/up, down) of layer 1's k_chain in the last decode step: block start, operand ready, result
published, in microseconds after the first block started (s_memrealtime, 100 MHz)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cosyvoice2-eu_amd'))
import numpy as np
import torch
from cv2amd import synth, lib as L
from cv2amd.llm import LLMEngine

sd = synth.make_llm(layers=24)
eng = LLMEngine(sd, 'cuda:0', max_seqs=1, max_pos=2048, max_out=2048)
inp = synth.synthetic_inputs(seed=0, text_len=50, prompt_len=255)
x = eng.build_lm_input(inp['text'], inp['prompt_text'], inp['prompt_token'])
eng.add_request(0, x, 2000, 2000, mode=1, seed=7, force_len=True)
nO, nGU = 56, 304
for rep in range(3):
    eng.step(1, 64)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (1024 * 4))()
    L.check(L.lib().cv2_debug_chain(buf))
    t = np.array(buf, dtype=np.int64).reshape(1024, 4)[:nO + nGU + 4 * nO].astype(np.float64)
    t0 = t[:, 0].min()
    us = (t - t0) / 100.0
    print('rep', rep)
    for name, sl in (('O', slice(0, nO)), ('gate/up', slice(nO, nO + nGU)), ('down', slice(nO + nGU, None))):
        r = us[sl]
        f = lambda a: f'min {a.min():6.2f} med {np.median(a):6.2f} max {a.max():6.2f}'
        print(f'  {name:8s} start [{f(r[:, 0])}]  operand [{f(r[:, 3])}]  result [{f(r[:, 1])}]  published [{f(r[:, 2])}]')
