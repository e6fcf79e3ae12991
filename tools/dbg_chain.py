"""Timeline of one layer inside the one-launch decode step k_step (diagnostic build only):
    ./build.sh -DCV2_STAMPS -o cosyvoice2-eu_amd/cv2amd/libcv2amd_dbg.so
    CV2_AMD_LIB=$PWD/cosyvoice2-eu_amd/cv2amd/libcv2amd_dbg.so python tools/dbg_chain.py
Per role (Q, A, O, gate/up, down) of layer 12 in the last decode step: block start, operand ready, result, published, in
microseconds after the layer's first block started (s_memrealtime, 100 MHz)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cosyvoice2-eu_amd'))
import numpy as np
import torch
from cv2amd import synth, lib as L
from cv2amd.llm import LLMEngine

sd = synth.make_llm(layers=24)
MAXPOS = int(os.environ.get('CV2_MAXPOS', '2048'))
NT = (MAXPOS + 127) // 128
eng = LLMEngine(sd, 'cuda:0', max_seqs=1, max_pos=MAXPOS, max_out=MAXPOS)
inp = synth.synthetic_inputs(seed=0, text_len=50, prompt_len=255)
x = eng.build_lm_input(inp['text'], inp['prompt_text'], inp['prompt_token'])
eng.add_request(0, x, MAXPOS - 400, MAXPOS - 400, mode=1, seed=7, force_len=True)
step1 = int(os.environ.get('CV2_STEP1', '0'))        # the one-row form that runs (llm.hip get_graph): bit 0 QA blocks, bit 1 two-pair gate/up blocks
nQ, nA, nO, nGU, nD = (8 if step1 & 1 else 36), (NT * 16 if step1 & 1 else NT * 2), 56, (152 if step1 & 2 else 304), 112
DUMP = sys.argv[sys.argv.index('--dump') + 1] if '--dump' in sys.argv else None
roles = (('KV' if step1 & 1 else 'Q', nQ), ('QA' if step1 & 1 else 'A', nA), ('O', nO), ('gate/up', nGU), ('down', nD))
for rep in range(3):
    eng.step(1, 64)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (1024 * 8))()
    L.check(L.lib().cv2_debug_chain(buf))
    t = np.array(buf, dtype=np.int64).reshape(1024, 8).astype(np.float64)
    live = t[:, 2] > 0
    # the layer's time origin: when its Q blocks received their operand is the end of the previous layer
    print('rep', rep)
    base = 0
    t0 = None
    for name, n in roles:
        r = t[base:base + n]
        m = live[base:base + n]
        base += n
        if not m.any():
            continue
        r = r[m]
        if t0 is None:
            t0 = r[:, 3].min() if (r[:, 3] > 0).any() else r[:, 0].min()
        us = (r - t0) / 100.0
        if DUMP and rep == 2:                     # every block of the role: index in the role, XCD of its grid position, stamps
            idx = np.nonzero(m)[0]
            with open(DUMP, 'a') as fdump:
                for i, u in zip(idx, us):
                    fdump.write(f'{name} {i:3d} start {u[0]:7.2f} operand {u[3]:7.2f} staged {u[4]:7.2f} mfma {u[5]:7.2f} reduced {u[6]:7.2f} result {u[1]:7.2f} published {u[2]:7.2f}\n')
        f = lambda a: f'min {a.min():7.2f} med {np.median(a):7.2f} max {a.max():7.2f}'
        print(f'  {name:8s} n={len(r):3d} start [{f(us[:, 0])}]  operand [{f(us[:, 3])}]  result [{f(us[:, 1])}]  published [{f(us[:, 2])}]')
        if name in ('A', 'QA'):
            g = lambda a: f'{np.median(a):5.2f}/{a.max():5.2f}'
            print(f'           q arrived [{f(us[:, 3])}]  q->scores {g(us[:, 4] - us[:, 3])}  scores->softmax {g(us[:, 5] - us[:, 4])}  softmax->pv {g(us[:, 6] - us[:, 5])}  pv->published {g(us[:, 2] - us[:, 6])}')
        elif (r[:, 4] > 0).all():
            g = lambda a: f'{np.median(a):5.2f}/{a.max():5.2f}'
            print(f'           operand->staged {g(us[:, 4] - us[:, 3])}  staged->mfma {g(us[:, 5] - us[:, 4])}  mfma->reduced {g(us[:, 6] - us[:, 5])}  reduced->result {g(us[:, 1] - us[:, 6])} (med/max us)')
