"""Where k_sample spends its time: runs decode steps with the sampler's diagnostic exit points (mode 2/3/4) and the full RAS
path (mode 1) / greedy (mode 0); post-process the kernel trace by call order.  python tools/dbg_sample.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cosyvoice2-eu_amd'))
import torch
from cv2amd import synth
from cv2amd.llm import LLMEngine
sd = synth.make_llm(layers=1)
eng = LLMEngine(sd, 'cuda:0', max_seqs=1, max_pos=512, max_out=256)
inp = synth.synthetic_inputs(text_len=20, prompt_len=30, prompt_text_len=5)
req = [(inp['text'], inp['prompt_text'], inp['prompt_token'])]
for mode in (0, 1, 2, 3, 4, 1):
    eng.generate_fixed(req, 101, mode=mode, seed=5)
    torch.cuda.synchronize()
print('done')
