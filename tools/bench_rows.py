"""Decode step time by number of rows (slots): python tools/bench_rows.py [rows ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cosyvoice2-eu_amd'))
import torch
from cv2amd import synth
from cv2amd.llm import LLMEngine

rows = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 16, 32]
dev = torch.device('cuda:0')
eng = LLMEngine(synth.make_llm(), dev, max_seqs=32, max_pos=1024, max_out=512)
inp = synth.synthetic_inputs(text_len=50, prompt_len=255, prompt_text_len=20)
req = (inp['text'], inp['prompt_text'], inp['prompt_token'])
for n in rows:
    ids = eng.generate([req] * n, force_len=40)          # warm: graphs for this row count
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    xs = [eng.build_lm_input(*req) for _ in range(n)]
    eng.add_requests(list(range(n)), xs, [(300, 300)] * n, 0, 1, True)
    torch.cuda.synchronize()
    e0.record()
    eng.step(n, 200)
    e1.record()
    torch.cuda.synchronize()
    print(f'rows {n:2d}: {e0.elapsed_time(e1) / 200 * 1e3:7.1f} us per step', flush=True)
    eng.park()
