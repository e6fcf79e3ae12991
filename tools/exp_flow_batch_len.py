"""Flow stage of a coalesced batch of n utterances against their length (all equal): python tools/exp_flow_batch_len.py n N [N ...]
(P = 255 prompt tokens, N generated tokens).  Shows the tile-quantisation steps of the batch kernels (k_tail_rows2: one 64-row block per CU)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cosyvoice2-eu_amd'))
import torch
from cv2amd import synth
from cv2amd.flow import FlowEngine
dev = 'cuda:0'
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
Ns = [int(a) for a in sys.argv[2:]] or [250, 300, 400]
flow = FlowEngine(synth.make_flow(), dev, max_utts=n, max_len=2 * (320 + 800))
inp = synth.synthetic_inputs(seed=1986, text_len=50, prompt_len=255, prompt_text_len=20)
for N in Ns:
    utts = [dict(token=torch.randint(0, 6561, (1, N), dtype=torch.int32), prompt_token=inp['prompt_token'].to(dev),
                 prompt_feat=inp['prompt_feat'].to(dev), embedding=inp['embedding'].to(dev)) for _ in range(n)]
    for _ in range(2):
        flow.inference_batch(utts, streaming=False, finalize=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        flow.inference_batch(utts, streaming=False, finalize=True)
    e1.record(); torch.cuda.synchronize()
    T = 2 * (255 + N)
    rows = n * 2 * ((T + 8 + 127) // 128 * 128)
    print(f'{n} utterances, N = {N:4d} (T = {T:5d} frames, {rows} rows = {rows // 64} tiles of 64): {e0.elapsed_time(e1) / 3:7.2f} ms = {e0.elapsed_time(e1) / 3 / n / T * 1e3:6.2f} us per frame', flush=True)
