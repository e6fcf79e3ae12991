"""What would a bf16 KV cache do to the decode parity?  ("evaluate a bf16 KV cache ... ids
re-checked against the oracle".)  CPU experiment on the oracle, configs[1] shape (24 layers, P=255, 50 text tokens, 250 forced greedy
steps, bf16-rounded weights as on the device): the same run with every cached key / value rounded to bf16 after the step that
produced it.  Prints how many of the 250 ids change and the logit error against the fp32-cache run.
    python tools/eval_bf16_kv.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cosyvoice2-eu_amd'))
import torch  # noqa: E402
from cv2amd import synth, weights as W  # noqa: E402
from oracle import llm as OL  # noqa: E402

torch.set_num_threads(int(os.environ.get('CV2_THREADS', '8')))
sd = W.round_llm_sd(synth.make_llm())
inp = synth.synthetic_inputs(text_len=50, prompt_len=255, prompt_text_len=20)
req = (inp['text'], inp['prompt_text'], inp['prompt_token'])
ref, logp_ref = OL.inference(sd, *req, force_len=250, return_logp=True)

step = OL.qwen2_step


def step_bf16(sd_, d, x, cache):
    y = step(sd_, d, x, cache)
    for i, kv in enumerate(cache):
        if kv is not None:
            cache[i] = (kv[0].bfloat16().float(), kv[1].bfloat16().float())
    return y


OL.qwen2_step = step_bf16
got, logp = OL.inference(sd, *req, force_len=250, return_logp=True)
OL.qwen2_step = step
first = next((i for i, (a, b) in enumerate(zip(ref, got)) if a != b), None)
n_diff = sum(a != b for a, b in zip(ref, got))
print(f'ids that differ: {n_diff} of {len(ref)}; first divergence at step {first}')
upto = first if first is not None else len(ref)
errs = [float((a - b)[torch.isfinite(a) & torch.isfinite(b)].abs().max()) for a, b in zip(logp_ref[:upto + 1], logp[:upto + 1])]
margins = [float(lp.topk(2).values[0] - lp.topk(2).values[1]) for lp in logp_ref[:upto + 1]]
print(f'max |dlogp| over the common prefix: {max(errs):.3e} (median {sorted(errs)[len(errs) // 2]:.3e}); '
      f'min top-1 margin there: {min(margins):.3e}')
