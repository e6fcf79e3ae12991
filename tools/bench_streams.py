"""First-chunk latency of N concurrent streaming calls on ONE model (BASELINE config 5: streaming, B = 8):
python tools/bench_streams.py [streams] [rounds] [--stagger MS] [--fresh-threads] [--trace].  Each round starts N calls at once (or within MS
milliseconds of each other); reports p50 / max of the time from a call to its first chunk."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cosyvoice2-eu_amd'))
import torch
from cv2amd import synth
from cosyvoice.cli.model import CosyVoice2Model

_a = [x for i, x in enumerate(sys.argv[1:]) if not x.startswith('--') and sys.argv[i] != '--stagger']
N = int(_a[0]) if len(_a) > 0 else 8
R = int(_a[1]) if len(_a) > 1 else 4
m = CosyVoice2Model(synth.make_llm(), synth.make_flow(), synth.make_hift(), max_text=128, max_prompt_tokens=320, max_new_tokens=1100,
                    max_batch=max(8, N))
m.stream_live_rows = os.environ.get('CV2_STREAM_LOCKSTEP', '0') != '1'      # A/B switch
inp = synth.synthetic_inputs(seed=1986, text_len=12, prompt_len=255, prompt_text_len=20)     # 12 text tokens -> <= 240 speech tokens
kw = dict(text=inp['text'], prompt_text=inp['prompt_text'], llm_prompt_speech_token=inp['prompt_token'],
          flow_prompt_speech_token=inp['prompt_token'], prompt_speech_feat=inp['prompt_feat'], flow_embedding=inp['embedding'],
          llm_embedding=inp['embedding'])


import random
STAGGER = float(sys.argv[sys.argv.index('--stagger') + 1]) if '--stagger' in sys.argv else 0.0      # --stagger MS: random start offsets
_rng = random.Random(1986)
POOL = '--fresh-threads' not in sys.argv       # callers are the persistent workers of a server's pool (default); --fresh-threads: new threads per
_pool = None                                   # round, whose first device copy costs ~20 ms of per-thread HIP initialisation on this image


class _Pool:
    def __init__(self, n):
        import queue
        self.q = [queue.Queue() for _ in range(n)]
        self.done = queue.Queue()
        self.ths = [threading.Thread(target=self._loop, args=(i,), daemon=True) for i in range(n)]
        [t.start() for t in self.ths]

    def _loop(self, i):
        while True:
            fn = self.q[i].get()
            try:
                fn(i)
            finally:
                self.done.put(i)

    def run(self, fn, n):
        for i in range(n):
            self.q[i].put(fn)
        for _ in range(n):
            self.done.get()


def run(n):
    global _pool
    first, total = [None] * n, [0.0] * n
    if POOL and _pool is None:
        _pool = _Pool(N)
    t0 = time.perf_counter()

    def work(i):
        t_call = t0
        if STAGGER > 0:                            # calls that do NOT start together: each waits a random 0 .. STAGGER ms first
            time.sleep(_rng.random() * STAGGER * 1e-3)
            t_call = time.perf_counter()
        for out in m.tts(**kw, stream=True):
            if first[i] is None:
                first[i] = time.perf_counter() - t_call
            total[i] += out['tts_speech'].shape[1] / 24000.0
    if POOL:
        _pool.run(work, n)
    else:
        ths = [threading.Thread(target=work, args=(i,)) for i in range(n)]
        [t.start() for t in ths]
        [t.join() for t in ths]
    return first, sum(total), time.perf_counter() - t0


run(1); run(N)                                   # warm-up (graphs for 1..N slots)
if '--trace' in sys.argv:                        # one round of N streams with a prompt the model has not seen, with the scheduler's log
    for new_prompt in (False, True):
        if new_prompt:
            m._prompt_caches.clear()
        m._sched_log = []
        t0 = time.perf_counter()
        f, audio, dt = run(N)
        log, m._sched_log = m._sched_log, None
        print(f'--- {N} streams, {"new" if new_prompt else "served"} prompt: first chunks (ms) {sorted(round(x * 1e3) for x in f)}')
        for t, kind, info in sorted(log):
            print(f'{(t - t0) * 1e3:8.2f} ms  {kind:6s} {info}')
    sys.exit(0)
for n in (1, N):
    firsts = []
    for _ in range(R):
        f, audio, dt = run(n)
        firsts += f
    firsts.sort()
    print(f'{n} concurrent stream(s): first chunk p50 {firsts[len(firsts) // 2] * 1e3:.1f} ms, max {firsts[-1] * 1e3:.1f} ms; '
          f'last round {audio:.1f} s of audio in {dt:.2f} s = {audio / dt:.1f} audio-s/s')
