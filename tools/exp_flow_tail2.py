"""Flow stage alone on a configs[2]-like batch (32 ragged utterances): the batch tail kernel k_tail_rows<4> (CV2_FLOW_TAIL_ROWS2=0) against
k_tail_rows2 (1) and k_tail_rows2 with the next block's QKV projection chained on (2, the default).  The variable is read once per
process: this script runs itself three times.  python tools/exp_flow_tail2.py [n_utts]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if 'TAIL2_CHILD' not in os.environ:
    n = sys.argv[1] if len(sys.argv) > 1 else '32'
    for mode in ('0', '1', '2', '0', '2'):
        env = dict(os.environ, TAIL2_CHILD='1', CV2_FLOW_TAIL_ROWS2=mode)
        subprocess.run([sys.executable, os.path.abspath(__file__), n], env=env, check=False)
    sys.exit(0)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cosyvoice2-eu_amd'))
import torch
from cv2amd import synth
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
out = flow.inference_batch(utts, streaming=False, finalize=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    out = flow.inference_batch(utts, streaming=False, finalize=True)
e1.record()
torch.cuda.synchronize()
import hashlib
h = hashlib.sha256(b''.join(m.cpu().numpy().tobytes() for m in out)).hexdigest()[:16]
finite = all(bool(torch.isfinite(m).all()) for m in out)
print(f'CV2_FLOW_TAIL_ROWS2={os.environ["CV2_FLOW_TAIL_ROWS2"]}: flow {e0.elapsed_time(e1) / 3:7.1f} ms per batch of {n}; mels sha {h} finite {finite}', flush=True)
