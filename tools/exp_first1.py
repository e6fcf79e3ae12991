"""Where the first chunk of ONE stream goes (bench.py's extra.streaming.streams_1): python tools/exp_first1.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cosyvoice2-eu_amd'))
import torch
import bench as B
dev = torch.device('cuda:0')
model = B.build_model(dev, 32)
sreq = B.request(1986, B.P_TOK, 12, dev)
for _ in range(3):
    B.run_calls(model, [sreq], [None], stream=True)
for rep in range(3):
    model._sched_log = []
    t0 = time.perf_counter()
    marks = []
    g = model.tts(**sreq, stream=True)
    first = next(g)
    t1 = time.perf_counter()
    for _ in g:
        pass
    log, model._sched_log = model._sched_log, None
    print(f'first chunk {1e3 * (t1 - t0):.1f} ms; chunk rounds: ' + ', '.join(f'start {(t - t0) * 1e3:.1f} ms took {i["ms"]} ms' for t, k, i in sorted(log) if k == 'chunks')[:200])
