"""Where the LLM stage's wall time of a 32-request batch goes: prefill, decode bursts (GPU time between events), and the gaps between
bursts (host polls of the state records).  python tools/exp_generate_bubbles.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cosyvoice2-eu_amd'))
import torch
import bench
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
m = bench.build_model(dev, 32)
g = torch.Generator().manual_seed(1986)
reqs = [bench.request(1986 + b, bench.P_TOK if b % 2 == 0 else bench.P_TOK_DE, bench.TEXT_LEN, dev) for b in range(32)]
forces = [int(torch.randint(150, 501, (1,), generator=g)) for _ in range(32)]
rq = [(r['text'], r['prompt_text'], r['llm_prompt_speech_token']) for r in reqs]
llm = m.llm
st = bench.StepTimer(llm)
orig_add = llm.add_requests
pre = []
def add(*a, **k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = orig_add(*a, **k); e1.record(); pre.append((e0, e1)); return r
llm.add_requests = add
for rep in range(3):
    st.rec.clear(); pre.clear(); st.on = True
    torch.cuda.synchronize(); t0 = time.perf_counter()
    toks = llm.generate(rq, mode=1, seed=rep + 1, force_len=forces, return_errors=True)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    dec, steps = st.result()
    pf = sum(a.elapsed_time(b) for a, b in pre)
    print(f'generate(32 requests): wall {1e3 * (t1 - t0):7.1f} ms = prefill {pf:6.1f} + decode bursts {dec:7.1f} ({steps} steps in {len(st.rec)} bursts) + gaps {1e3 * (t1 - t0) - pf - dec:6.1f}', flush=True)
