"""First chunk of 8 concurrent streams (bench.py's extra.streaming.streams_8 workload) with the scheduler's log: served prompt, then a new prompt.
python tools/exp_newprompt.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cosyvoice2-eu_amd'))
import torch
import bench as B
dev = torch.device('cuda:0')
model = B.build_model(dev, 32)
sreq = B.request(1986, B.P_TOK, 12, dev)
B.run_calls(model, [sreq] * 8, [None] * 8, stream=True)
B.run_calls(model, [sreq] * 8, [None] * 8, stream=True)
for new in (False, True, True):
    if new:
        model._prompt_caches.clear()
    model._sched_log = []
    t0 = time.perf_counter()
    _, first = B.run_calls(model, [sreq] * 8, [None] * 8, stream=True)
    log, model._sched_log = model._sched_log, None
    print(f'--- {"new" if new else "served"} prompt: first chunks (ms) {sorted(round(x * 1e3) for x in first)}')
    for t, kind, info in sorted(log)[:6]:
        print(f'{(t - t0) * 1e3:8.2f} ms  {kind:6s} {info}')
    time.sleep(0.3)
