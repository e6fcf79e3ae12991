"""Which hand-off values differ between the one-row step's forms (CV2_STEP1 = 0 / 1 / 2 / 3)?  One engine per form (the variable is read when a
decode graph is captured), the same request, one decode step; layer 0's granule buffers and the logits are compared bit for bit."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cosyvoice2-eu_amd'))
import numpy as np
import torch
from cv2amd import synth, lib as L
from cv2amd.llm import LLMEngine

sd = synth.make_llm(layers=3)
inp = synth.synthetic_inputs(seed=3, text_len=9, prompt_len=150, prompt_text_len=4)
res = {}
for mode in ('0', '1', '2', '3'):
    os.environ['CV2_STEP1'] = mode
    eng = LLMEngine(sd, 'cuda:0', max_seqs=1, max_pos=512, max_out=64)
    x = eng.build_lm_input(inp['text'], inp['prompt_text'], inp['prompt_token'])
    eng.add_request(0, x, 40, 40, force_len=True)
    eng.step(1, 1)
    torch.cuda.synchronize()
    p = (C.c_uint64 * 16)()
    L.check(eng.lib.cv2_llm_debug_ptrs(eng.handle, p))
    gran, gl, off_dg, off_qg, off_kv, off_ag, off_hg = p[8], p[9], p[10], p[11], p[12], p[13], p[14]
    n = int(gl) * 3
    buf = (C.c_uint64 * n)()
    from torch.cuda import cudart
    t = torch.empty(n, dtype=torch.int64, device='cuda:0')
    assert C.CDLL('libamdhip64.so').hipMemcpy(C.c_void_p(t.data_ptr()), C.c_void_p(gran), C.c_size_t(n * 8), C.c_int(3)) == 0
    g = t.cpu().numpy().view(np.uint64)
    vals = (g & 0xffffffff).astype(np.uint32).view(np.float32)
    tags = (g >> 32).astype(np.uint32)
    res[mode] = dict(vals=vals, tags=tags, logits=eng.logits[0, :eng.vocab].cpu().numpy().copy(), offs=(int(gl), int(off_dg), int(off_qg), int(off_kv), int(off_ag), int(off_hg)))
gl, off_dg, off_qg, off_kv, off_ag, off_hg = res['0']['offs']
regions = (('x_mid', 0, off_dg), ('down partials', off_dg, off_qg), ('q', off_qg, off_kv), ('k/v new', off_kv, off_ag), ('att tiles', off_ag, off_hg), ('h', off_hg, gl))
ep = res['0']['tags'].max()
for mode in ('1', '2', '3'):
    print('== CV2_STEP1 =', mode, 'against 0: logits equal', np.array_equal(res[mode]['logits'], res['0']['logits']),
          'max diff', np.abs(res[mode]['logits'] - res['0']['logits']).max())
    for layer in range(3):
        for name, a, b in regions:
            va, vb = res['0']['vals'][layer * gl + a:layer * gl + b], res[mode]['vals'][layer * gl + a:layer * gl + b]
            live = (res['0']['tags'][layer * gl + a:layer * gl + b] == ep) & (res[mode]['tags'][layer * gl + a:layer * gl + b] == res[mode]['tags'].max())
            ne = (va.view(np.uint32) != vb.view(np.uint32)) & live
            print(f'   layer {layer} {name:14s} live {int(live.sum()):5d} differing {int(ne.sum()):5d}' + (f'  first at {int(np.nonzero(ne)[0][0])}: {va[ne][0]!r} vs {vb[ne][0]!r}' if ne.any() else ''))
