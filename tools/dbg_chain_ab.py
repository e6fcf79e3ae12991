"""k_step (one launch per decode step) against the five-launches-per-layer path: greedy ids of the same request, several depths."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cosyvoice2-eu_amd'))
import torch
from cv2amd import synth
from cv2amd.llm import LLMEngine

inp = synth.synthetic_inputs(text_len=30, prompt_len=40, prompt_text_len=10)
req = [(inp['text'], inp['prompt_text'], inp['prompt_token'])]
for layers, max_pos in ((3, 512), (6, 1024), (12, 1024), (24, 1024), (24, 2048)):
    sd = synth.make_llm(layers=layers)
    out = {}
    for chain in ('0', '1'):
        os.environ['CV2_LLM_CHAIN'] = chain
        eng = LLMEngine(sd, 'cuda:0', max_seqs=4, max_pos=max_pos, max_out=256)
        out[chain] = eng.generate(req, force_len=60)[0]
        del eng
    same = out['0'] == out['1']
    first = next((i for i, (a, b) in enumerate(zip(out['0'], out['1'])) if a != b), -1)
    print(f'layers {layers:2d} max_pos {max_pos}: same={same} first diff {first}  {out["0"][:6]} vs {out["1"][:6]}', flush=True)
