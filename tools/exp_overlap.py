"""Does a many-row decode burst overlap with a flow batch on another stream?  python tools/exp_overlap.py [rows] [utts] [steps]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cosyvoice2-eu_amd'))
import torch
from cv2amd import synth
from cv2amd.llm import LLMEngine
from cv2amd.flow import FlowEngine

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 16
utts_n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 120
dev = 'cuda:0'
eng = LLMEngine(synth.make_llm(layers=24), dev, max_seqs=32, max_pos=2048, max_out=2048)
for b in range(rows):
    inp = synth.synthetic_inputs(seed=b, text_len=50, prompt_len=255)
    eng.add_request(b, eng.build_lm_input(inp['text'], inp['prompt_text'], inp['prompt_token']), 2000, 2000, force_len=True)
flow = FlowEngine(synth.make_flow(), dev, max_utts=utts_n, max_len=2 * (320 + 512))
inp = synth.synthetic_inputs(seed=1986, text_len=50, prompt_len=255, prompt_text_len=20)
utts = [dict(token=torch.randint(0, 6561, (1, 250), dtype=torch.int32), prompt_token=inp['prompt_token'].to(dev),
             prompt_feat=inp['prompt_feat'].to(dev), embedding=inp['embedding'].to(dev)) for _ in range(utts_n)]
s_llm = torch.cuda.Stream(dev)


def t(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3


def dec():
    with torch.cuda.stream(s_llm):
        eng.step(rows, steps, shared=True)


def fl():
    flow.inference_batch(utts, streaming=False, finalize=True)


for _ in range(2):
    dec(); fl(); torch.cuda.synchronize()
a = t(dec); b = t(fl)
c = t(lambda: (dec(), fl()))
print(f'rows {rows} x {steps} steps: decode alone {a:.1f} ms, flow({utts_n}) alone {b:.1f} ms, sum {a + b:.1f}, both streams {c:.1f} ms')
