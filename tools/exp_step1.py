"""One-row decode step (k_step<false>) at the configs[1] context: python tools/exp_step1.py  (A/B switches: CV2_CHAIN_PF=0/1 ...)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cosyvoice2-eu_amd'))
import torch
from cv2amd import synth
from cv2amd.llm import LLMEngine
eng = LLMEngine(synth.make_llm(layers=24), 'cuda:0', max_seqs=1, max_pos=968, max_out=600)
inp = synth.synthetic_inputs(seed=0, text_len=50, prompt_len=255, prompt_text_len=20)
x = eng.build_lm_input(inp['text'], inp['prompt_text'], inp['prompt_token'])
res = []
for rep in range(3):
    eng.add_request(0, x, 500, 500, mode=1, seed=7, force_len=True)
    eng.step(1, 16); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); eng.step(1, 232); e1.record(); torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) / 232 * 1e3)
toks = eng.read(1)[1][0]
print(f'one-row step, positions 343 .. 575: {min(res):.1f} us (runs: {[round(r, 1) for r in res]}); checksum of ids {sum(toks) % 100003}')
