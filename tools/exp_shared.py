"""8 concurrent streams x 250 forced tokens (bench.py's extra.streaming.streams_8 forced250 leg) and the 8 generator-text streams:
throughput and chunk gaps; run with / without CV2_SHARED_ONE_LAUNCH=1 (decode bursts beside flow + HiFT as one launch or as launches)."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cosyvoice2-eu_amd'))
import torch
import bench as B
dev = torch.device('cuda:0')
model = B.build_model(dev, 32)
sreq = B.request(1986, B.P_TOK, 12, dev)
for n in (1, 8):
    B.run_calls(model, [sreq] * n, [250] * n, stream=True)
    gaps, audio, dts = [], 0.0, 0.0
    for _ in range(2):
        ct = [[] for _ in range(n)]
        t0 = time.perf_counter()
        wavs, _ = B.run_calls(model, [sreq] * n, [250] * n, stream=True, chunk_times=ct)
        dts += time.perf_counter() - t0
        audio += sum(w.shape[1] for w in wavs) / 24000.0
        for c in ct:
            gaps += [b - a for a, b in zip(c[:-2], c[1:-1])]
    gaps.sort()
    print(f'{n} stream(s) x 250 tokens: {audio / dts:6.1f} audio-s/s, chunk gap p50 {gaps[len(gaps) // 2] * 1e3:.1f} ms')
