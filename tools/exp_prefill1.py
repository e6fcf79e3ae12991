"""One prompt's prefill alone, 12 times (289 rows, 24 layers), for a kernel trace: python tools/exp_prefill1.py
(CV2_PREFILL_SPLITK=0: the down projection as one GEMM of 35 blocks instead of four K slices)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cosyvoice2-eu_amd'))
import torch
from cv2amd import synth
from cv2amd.llm import LLMEngine

eng = LLMEngine(synth.make_llm(layers=24), 'cuda:0', max_seqs=1, max_pos=2048, max_out=2048)
inp = synth.synthetic_inputs(seed=0, text_len=12, prompt_len=255, prompt_text_len=20)
x = eng.build_lm_input(inp['text'], inp['prompt_text'], inp['prompt_token'])
for _ in range(12):
    eng.park(); eng.add_requests([0], [x], [(10, 10)], 1, 0, True)       # the GEMM path (cv2_llm_prefill_batch), as the model's _llm_start
torch.cuda.synchronize()
print('rows', x.shape[0], 'first id', eng.read(1)[1][0][:1])
