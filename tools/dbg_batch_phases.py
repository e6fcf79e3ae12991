"""Where does a coalesced batch of 32 spend the time that is not decode / flow / HiFT kernels?  Host timestamps around the
phases of CosyVoice2Model._run_batch (with a device synchronisation after each, so phases do not overlap):
    python tools/dbg_batch_phases.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cosyvoice2-eu_amd'))
import torch
import bench

dev = torch.device('cuda:0')
model = bench.build_model(dev, 32)
g = torch.Generator().manual_seed(1986)
reqs = [bench.request(1986 + b, bench.P_TOK if b % 2 == 0 else bench.P_TOK_DE, bench.TEXT_LEN, dev) for b in range(32)]
forces = [int(torch.randint(150, 501, (1,), generator=g)) for _ in range(32)]
model.coalesce_ms = 50.0
log = []


def wrap(obj, name, label):
    orig = getattr(obj, name)

    def f(*a, **k):
        torch.cuda.synchronize()
        t = time.perf_counter()
        r = orig(*a, **k)
        torch.cuda.synchronize()
        log.append((label, (time.perf_counter() - t) * 1e3))
        return r
    setattr(obj, name, f)


wrap(model, '_run_batch', 'run_batch')
wrap(model.llm, 'generate', ' llm.generate')
wrap(model.llm, 'add_requests', '  prefill (add_requests)')
wrap(model.llm, 'step', '  decode (step)')
wrap(model.llm, 'step_rows', '  decode (step_rows)')
wrap(model.llm, 'build_lm_input', '  build_lm_input')
wrap(model.llm, 'read', '  read')
wrap(model.flow, 'inference_batch', ' flow')
wrap(model.hift_pool, 'inference_many', ' hift')
bench.run_calls(model, reqs, forces)
del log[:]
t0 = time.perf_counter()
bench.run_calls(model, reqs, forces)
total = (time.perf_counter() - t0) * 1e3
agg = {}
for k, v in log:
    agg.setdefault(k, [0, 0.0])
    agg[k][0] += 1; agg[k][1] += v
print(f'run_calls total {total:.1f} ms')
for k, (n, v) in agg.items():
    print(f'{k:28s} n={n:4d}  {v:8.1f} ms')
