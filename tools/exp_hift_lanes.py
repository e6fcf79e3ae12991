"""How much the vocoder gains from running several utterances as workspace LANES of one set of launches (cv2_hift_inference_batch: gridDim.z)
instead of one call per utterance on the pool's HIP streams: n utterances of T frames each, ms per utterance.
python tools/exp_hift_lanes.py [T] [n]   (T > 160 needs a build with -DHG_MAX_T=1024: the batch entry stages through the graph buffers)
Measured (round 5): 8 x 600 frames: one after the other 3.54 ms each, pool of 4 streams 2.78, 8 lanes 2.45; 8 x 400: 2.87 / 2.09 / 1.75."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cosyvoice2-eu_amd'))
import torch
from cv2amd import synth
from cv2amd.hift import HiftEngine, HiftPool
T = int(sys.argv[1]) if len(sys.argv) > 1 else 600
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = 'cuda:0'
sd = synth.make_hift()
pool = HiftPool(sd, dev, max_frames=1024, n=4)
lanes = HiftEngine(sd, dev, max_frames=1024, share_weights_with=pool.engines[0], lanes=n)
mels = [(torch.randn(1, 80, T, device=dev) * 2 - 4).clamp(-11.5, 2) for _ in range(n)]


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


t_serial = timeit(lambda: [pool.engines[0].inference(m, None, seed=1) for m in mels])
t_pool = timeit(lambda: pool.inference_many(mels))
t_lanes = timeit(lambda: lanes.inference_batch(mels, [None] * n, seeds=list(range(1, n + 1))))
print(f'{n} utterances x {T} frames: one after the other {t_serial / n:.3f} ms each; pool of 4 streams {t_pool / n:.3f}; {n} lanes of one launch set {t_lanes / n:.3f}')
