"""Decode steps at a fixed number of live rows, for the profilers: python tools/prof_rows.py <rows> [steps]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cosyvoice2-eu_amd'))
import torch
from cv2amd import synth
from cv2amd.llm import LLMEngine

n = int(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 16
eng = LLMEngine(synth.make_llm(layers=24), 'cuda:0', max_seqs=32, max_pos=2048, max_out=2048)
for b in range(n):
    inp = synth.synthetic_inputs(seed=b, text_len=50, prompt_len=255)
    eng.add_request(b + (32 - n), eng.build_lm_input(inp['text'], inp['prompt_text'], inp['prompt_token']), 2000, 2000, force_len=True)
eng.step_rows(list(range(32 - n, 32)), steps)
torch.cuda.synchronize()
