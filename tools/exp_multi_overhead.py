"""Where k_step<true> loses against the one-row kernel at one row: python tools/exp_multi_overhead.py (run with / without CV2_CHAIN_FORCE_MULTI=1)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cosyvoice2-eu_amd'))
import torch
from cv2amd import synth
from cv2amd.llm import LLMEngine
eng = LLMEngine(synth.make_llm(layers=24), 'cuda:0', max_seqs=8, max_pos=2048, max_out=2048)
for b in range(8):
    inp = synth.synthetic_inputs(seed=b, text_len=50, prompt_len=255)
    eng.add_request(b, eng.build_lm_input(inp['text'], inp['prompt_text'], inp['prompt_token']), 2000, 2000, force_len=True)


def t(fn, n=64):
    fn(16); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(n); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print(f'step(1)            {t(lambda n: eng.step(1, n)):7.1f} us   (slot 0; k_step<false> unless CV2_CHAIN_FORCE_MULTI)')
print(f'step_rows([3])     {t(lambda n: eng.step_rows([3], n)):7.1f} us   (k_step<true>, row -> slot map)')
print(f'step(2)            {t(lambda n: eng.step(2, n)):7.1f} us   (k_step<true>, identity slots)')
print(f'step_rows([2, 5])  {t(lambda n: eng.step_rows([2, 5], n)):7.1f} us   (k_step<true>, row -> slot map)')
print(f'step(4)            {t(lambda n: eng.step(4, n)):7.1f} us')
print(f'step_rows([1,3,5,7]) {t(lambda n: eng.step_rows([1, 3, 5, 7], n)):7.1f} us')
