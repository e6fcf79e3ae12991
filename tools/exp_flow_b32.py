"""Flow stage alone on a configs[2]-like batch (32 ragged utterances), the estimator attention with / without LDS DMA staging of its
key / value tiles (test hook cv2_flow_debug_attn_dma):  python tools/exp_flow_b32.py [n_utts]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cosyvoice2-eu_amd'))
import torch
from cv2amd import synth, lib as L
from cv2amd.flow import FlowEngine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = 'cuda:0'
flow = FlowEngine(synth.make_flow(), dev, max_utts=n, max_len=2 * (320 + 512))
utts = []
for i in range(n):
    inp = synth.synthetic_inputs(seed=500 + i, text_len=50, prompt_len=150 + (37 * i) % 100, prompt_text_len=20)
    g = torch.Generator().manual_seed(i)
    utts.append(dict(token=torch.randint(0, 6561, (1, 150 + (53 * i) % 250), generator=g, dtype=torch.int32), prompt_token=inp['prompt_token'].to(dev),
                     prompt_feat=inp['prompt_feat'].to(dev), embedding=inp['embedding'].to(dev)))


def timed(reps=3):
    flow.inference_batch(utts, streaming=False, finalize=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = flow.inference_batch(utts, streaming=False, finalize=True)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out


res = {}
for on in (1, 0, 1, 0):
    L.check(L.lib().cv2_flow_debug_attn_dma(on))
    ms, out = timed()
    res.setdefault(on, []).append(ms)
    print(f'attention tiles by {"LDS DMA" if on else "registers"}: flow {ms:7.1f} ms per batch of {n}', flush=True)
    if on == 1: keep = [m.clone() for m in out]
    else: print('   max |diff| / range vs the DMA form:', max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(keep, out)))
L.check(L.lib().cv2_flow_debug_attn_dma(-1))
print('DMA', min(res[1]), 'registers', min(res[0]))
