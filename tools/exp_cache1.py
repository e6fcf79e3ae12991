"""One stream x 250 tokens: per-chunk recompute (default for a lone short stream) against the cached flow from the second chunk on."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cosyvoice2-eu_amd'))
import torch
import bench as B
dev = torch.device('cuda:0')
model = B.build_model(dev, 32)
sreq = B.request(1986, B.P_TOK, 12, dev)
for mg in (2, 1):
    model.flow_cache_min_group = mg
    B.run_calls(model, [sreq], [250], stream=True)
    gaps, audio, dts = [], 0.0, 0.0
    for _ in range(3):
        ct = [[]]
        t0 = time.perf_counter()
        wavs, _ = B.run_calls(model, [sreq], [250], stream=True, chunk_times=ct)
        dts += time.perf_counter() - t0
        audio += wavs[0].shape[1] / 24000.0
        gaps += [b - a for a, b in zip(ct[0][:-2], ct[0][1:-1])]
    gaps.sort()
    print(f'flow_cache_min_group={mg}: 1 stream x 250 tokens: {audio / dts:5.1f} audio-s/s, chunk gap p50 {gaps[len(gaps) // 2] * 1e3:.1f} ms, last gaps {[round(g * 1e3, 1) for g in gaps[-3:]]}')
