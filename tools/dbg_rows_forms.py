"""How far apart are the decode step's forms by row count?  The same three requests as rows 0 .. 2 of steps of n rows (launches), logits
after 7 steps against the 3-row run, relative to the largest logit.  CV2_AMD_LIB selects the library (A/B against an older build)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cosyvoice2-eu_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from cv2amd import synth
from cv2amd.llm import LLMEngine
from test_llm_gpu import _requests
sd = synth.make_llm(layers=int(os.environ.get('LAYERS', '3')))
eng = LLMEngine(sd, 'cuda:0', max_seqs=32, max_pos=512, max_out=64)
reqs = _requests(32, seed=900)
xs = [eng.build_lm_input(*r) for r in reqs]
ref = None
for shared in (True, False):
    for n in (3, 8, 12, 16, 17, 20, 32):
        if not shared and n > 24:
            continue
        eng.park()
        eng.add_requests(list(range(n)), xs[:n], [(12, 12)] * n, 0, 0, True)
        eng.step(n, 6, shared=shared)
        torch.cuda.synchronize()
        st, toks = eng.read(n)
        lg = eng.logits[:3, :eng.vocab].clone()
        if ref is None:
            ref = (toks[:3], lg)
        print(f'{"launches" if shared else "one launch"} n={n:2d}: ids equal {toks[:3] == ref[0]}, logits max rel diff vs 3-row launches {(lg - ref[1]).abs().max().item() / ref[1].abs().max().item():.3e}')
