"""First-chunk latency of streaming synthesis (BASELINE configs[4], single stream): python tools/bench_stream.py [runs]
Time from the tts() call to the first yielded chunk: LLM prefill + the 48 tokens the first chunk needs (hop 25 + pad 20 +
look-ahead 3 at P=255, cli/model.py:353-357) + chunk-masked flow at T=600 + HiFT on 90 frames.  The reference polls every
100 ms (cli/model.py:355); this scheduler does not sleep."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cosyvoice2-eu_amd'))
import numpy as np
import torch
from cv2amd import synth
from cosyvoice.cli.model import CosyVoice2Model

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 12
m = CosyVoice2Model(synth.make_llm(), synth.make_flow(), synth.make_hift(), max_text=128, max_prompt_tokens=320, max_new_tokens=1100)
inp = synth.synthetic_inputs(text_len=50, prompt_len=255, prompt_text_len=20)
kw = dict(text=inp['text'], flow_embedding=inp['embedding'], llm_embedding=inp['embedding'], prompt_text=inp['prompt_text'],
          llm_prompt_speech_token=inp['prompt_token'], flow_prompt_speech_token=inp['prompt_token'], prompt_speech_feat=inp['prompt_feat'])
lat, chunk_s = [], []
for i in range(runs + 2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g = m.tts(**kw, stream=True)
    first = next(g)
    t1 = time.perf_counter()
    g.close()
    if i >= 2:
        lat.append((t1 - t0) * 1e3)
        chunk_s.append(first['tts_speech'].shape[1] / 24000)
print(f'streaming first-chunk latency over {runs} runs: p50 {np.percentile(lat, 50):.1f} ms, p90 {np.percentile(lat, 90):.1f} ms, min {min(lat):.1f} ms; '
      f'first chunk = {chunk_s[0]:.2f} s of audio')
