"""Where does the first-chunk latency of N concurrent streams go?  Host timestamps (relative to the common start) of every prefill
batch, decode burst and chunk round on one model: python tools/dbg_stream_phases.py [streams]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cosyvoice2-eu_amd'))
import torch
from cv2amd import synth
from cosyvoice.cli.model import CosyVoice2Model

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
FORCE = int(sys.argv[2]) if len(sys.argv) > 2 else None      # forced token count (None: EOS live)
UPTO = float(sys.argv[3]) if len(sys.argv) > 3 else None     # print the log up to this time (ms) instead of up to the first chunks
m = CosyVoice2Model(synth.make_llm(), synth.make_flow(), synth.make_hift(), max_text=128, max_prompt_tokens=320, max_new_tokens=1100, max_batch=8)
inp = synth.synthetic_inputs(seed=1986, text_len=12, prompt_len=255, prompt_text_len=20)
kw = dict(text=inp['text'], prompt_text=inp['prompt_text'], llm_prompt_speech_token=inp['prompt_token'],
          flow_prompt_speech_token=inp['prompt_token'], prompt_speech_feat=inp['prompt_feat'], flow_embedding=inp['embedding'], llm_embedding=inp['embedding'])
log, t0 = [], [0.0]


def wrap(obj, name, label, sync=None):
    orig = getattr(obj, name)

    def f(*a, **k):
        ta = time.perf_counter() - t0[0]
        r = orig(*a, **k)
        if sync is not None:
            sync()
        log.append((label, ta * 1e3, (time.perf_counter() - t0[0]) * 1e3, a[0] if (a and isinstance(a[0], int)) else (len(a[0]) if a and hasattr(a[0], '__len__') else '')))
        return r
    setattr(obj, name, f)


wrap(m.llm, 'add_requests', 'prefill', m.llm_stream.synchronize)
wrap(m.llm, 'step', 'decode', m.llm_stream.synchronize)
wrap(m, '_run_chunks', 'chunks', torch.cuda.synchronize)
wrap(m.flow, 'inference_batch', ' flow', torch.cuda.synchronize)
wrap(m.flow, 'inference_chunk_batch', ' flow$', torch.cuda.synchronize)
for k_, e_ in enumerate(m.hift_pool.engines):
    wrap(e_, 'inference', f'  hift{k_}', None)


def run(n):
    first = [None] * n
    t0[0] = time.perf_counter()

    def work(i):
        for out in m.tts(**kw, stream=True, force_len=FORCE):
            if first[i] is None:
                first[i] = (time.perf_counter() - t0[0]) * 1e3
    ths = [threading.Thread(target=work, args=(i,)) for i in range(n)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    return first


run(1); run(N)
del log[:]
first = run(N)
print('first chunks ms:', ' '.join(f'{f:.0f}' for f in sorted(first)))
for lab, a, b, n in [e for e in log if e[1] < (UPTO or max(first) + 5)][:(400 if UPTO else 40)]:
    print(f'{lab:8s} start {a:7.1f}  end {b:7.1f}  ({b - a:6.1f} ms)  n={n}')
