"""Micro-benchmark of the decode step (not the headline bench): python tools/bench_llm.py [batch] [steps]"""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cosyvoice2-eu_amd'))
import torch
from cv2amd import synth
from cv2amd.llm import LLMEngine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
P = int(sys.argv[3]) if len(sys.argv) > 3 else 0
t0 = time.time()
sd = synth.make_llm(layers=24)
print('synth', time.time() - t0)
MAXPOS = int(os.environ.get('CV2_MAXPOS', '2048'))      # (the product's engine at BASELINE configs[1]: 968 -> 8 attention tiles)
eng = LLMEngine(sd, 'cuda:0', max_seqs=32, max_pos=MAXPOS, max_out=MAXPOS)
print('engine', time.time() - t0, 'weight MB', eng.weight_bytes / 1e6)
for b in range(B):
    inp = synth.synthetic_inputs(seed=b, text_len=50, prompt_len=P)
    x = eng.build_lm_input(inp['text'], inp['prompt_text'], inp['prompt_token'])
    torch.cuda.synchronize(); t = time.time()
    eng.add_request(b, x, 5000, 5000, force_len=True)
    torch.cuda.synchronize()
    if b == 0: print('prefill rows', x.shape[0], 'ms', (time.time() - t) * 1e3)
eng.step(B, 20)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
eng.step(B, steps)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / steps
print(f'B={B} decode step {ms*1e3:.1f} us  -> {eng.weight_bytes/ms/1e6:.1f} GB/s weight stream, {B/ms*1e3:.0f} tok/s')
