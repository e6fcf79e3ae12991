"""One streaming call (P=255, 250 forced tokens) on the bench's model, three times: for a kernel trace of the chunk path.
  rocprofv3 --kernel-trace --stats --output-format csv -d out -- python3 tools/prof_stream.py [n_streams]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench  # noqa: E402
import torch  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device('cuda:0')
model = bench.build_model(dev, 8)
req = bench.request(1986, bench.P_TOK, 12, dev)
bench.run_calls(model, [req] * n, [250] * n, stream=True)
torch.cuda.synchronize()
for _ in range(2):
    ct = [[] for _ in range(n)]
    t0 = time.perf_counter()
    bench.run_calls(model, [req] * n, [250] * n, stream=True, chunk_times=ct)
    print('total %.1f ms; chunk times (stream 0): %s' % ((time.perf_counter() - t0) * 1e3, ' '.join('%.0f' % (t * 1e3) for t in ct[0])))
