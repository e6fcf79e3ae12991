"""Throughput of ONE CosyVoice2Model called from several threads (the evaluation harness pattern,
evaluation/cosyvoice_synthesizer.py:219,260): python tools/bench_threads.py [threads] [calls_per_thread]
Non-streaming calls are coalesced into batches by the scheduler; prints audio-s/s for 1 thread and for N threads."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cosyvoice2-eu_amd'))
import torch
from cv2amd import synth
from cosyvoice.cli.model import CosyVoice2Model

NT = int(sys.argv[1]) if len(sys.argv) > 1 else 8
NC = int(sys.argv[2]) if len(sys.argv) > 2 else 3
m = CosyVoice2Model(synth.make_llm(), synth.make_flow(), synth.make_hift(), max_text=128, max_prompt_tokens=320, max_new_tokens=1100,
                    max_batch=8)
inp = synth.synthetic_inputs(seed=1986, text_len=50, prompt_len=255, prompt_text_len=20)
kw = dict(text=inp['text'], prompt_text=inp['prompt_text'], llm_prompt_speech_token=inp['prompt_token'],
          flow_prompt_speech_token=inp['prompt_token'], prompt_speech_feat=inp['prompt_feat'], flow_embedding=inp['embedding'],
          llm_embedding=inp['embedding'])


def run(nthreads, ncalls):
    samples = [0] * nthreads

    def work(i):
        for _ in range(ncalls):
            for out in m.tts(**kw, stream=False):
                samples[i] += out['tts_speech'].shape[1]
    ths = [threading.Thread(target=work, args=(i,)) for i in range(nthreads)]
    t0 = time.perf_counter()
    [t.start() for t in ths]
    [t.join() for t in ths]
    dt = time.perf_counter() - t0
    return sum(samples) / 24000.0, dt


run(1, 1)                                   # warm-up (graphs, allocator)
n0 = len(m.batch_sizes)
a1, t1 = run(1, NC)
n1 = len(m.batch_sizes)
aN, tN = run(NT, NC)
print(f'1 thread : {a1:.1f} s of audio in {t1:.2f} s = {a1 / t1:.1f} audio-s/s')
print(f'{NT} threads: {aN:.1f} s of audio in {tN:.2f} s = {aN / tN:.1f} audio-s/s; batch sizes {m.batch_sizes[n1:]}')
