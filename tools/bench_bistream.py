"""BASELINE configs[4] as worded: N concurrent tts(text=<generator>, stream=True) calls on one model, with the scheduler's own log of
the hub rounds (feeds / bursts) and chunk rounds (flow + HiFT batches): python tools/bench_bistream.py [streams] [rounds] [--trace].
The same workload as bench.py's extra.streaming.bistream_8 (120 text tokens in pieces of 5, P = 255, greedy harness, EOS / fill biases)."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cosyvoice2-eu_amd'))
import torch
from cv2amd import synth
from cv2amd.llm import MODE_GREEDY
from cosyvoice.cli.model import CosyVoice2Model

args = [a for a in sys.argv[1:] if not a.startswith('--')]
N = int(args[0]) if len(args) > 0 else 8
R = int(args[1]) if len(args) > 1 else 3
TRACE = '--trace' in sys.argv
m = CosyVoice2Model(synth.make_llm(), synth.make_flow(), synth.make_hift(), max_text=128, max_prompt_tokens=320, max_new_tokens=1100, max_batch=8)
bd = m.llm.bdec
bd[6563] += 6.0; bd[6561] += 24.0; bd[6562] = -30.0
m.sampling_mode = MODE_GREEDY
inp = synth.synthetic_inputs(seed=1986, text_len=120, prompt_len=255, prompt_text_len=20)
pieces = [inp['text'][:, i:i + 5] for i in range(0, 120, 5)]
kw = dict(prompt_text=inp['prompt_text'], llm_prompt_speech_token=inp['prompt_token'], flow_prompt_speech_token=inp['prompt_token'],
          prompt_speech_feat=inp['prompt_feat'], flow_embedding=inp['embedding'], llm_embedding=inp['embedding'])


def run(n):
    first, total, times = [None] * n, [0.0] * n, [[] for _ in range(n)]
    t0 = time.perf_counter()

    def work(i):
        for out in m.tts(text=(p for p in pieces), **kw, stream=True):
            times[i].append(time.perf_counter() - t0)
            if first[i] is None:
                first[i] = times[i][-1]
            total[i] += out['tts_speech'].shape[1] / 24000.0
    ths = [threading.Thread(target=work, args=(i,)) for i in range(n)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    return first, sum(total), time.perf_counter() - t0, times, t0


run(N)
for r in range(R):
    m._sched_log = [] if TRACE and r == R - 1 else None
    f, audio, dt, times, t0 = run(N)
    f.sort()
    print(f'{N} generator-text streams: first chunk p50 {f[len(f) // 2] * 1e3:.1f} ms, max {f[-1] * 1e3:.1f} ms; {audio:.1f} s of audio in {dt * 1e3:.0f} ms = '
          f'{audio / dt:.1f} audio-s/s; chunk times of stream 0 (ms): {[round(t * 1e3) for t in times[0]]}')
if TRACE:
    for t, kind, info in sorted(m._sched_log):
        print(f'{(t - t0) * 1e3:8.2f} ms  {kind:6s} {info}')
