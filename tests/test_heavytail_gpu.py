"""Parity off the Gaussian (VERDICT round 2, weak point 1).  Every other parity test runs on N(0, s) weights with norm gains near 1.
Real checkpoints are not like that: Qwen2 has RMSNorm gains from 0.05 to 30, a few 'massive activation' residual channels hundreds
of times larger than the rest, outlier q / k biases; trained flow LayerNorm gains and convolution rows spread over an order of
magnitude.  `cv2amd.synth.heavy_tail_llm / heavy_tail_flow` give the synthetic checkpoints those statistics; the bars are the same
as on the Gaussian ones: greedy ids bit-exact against the oracle on the same bf16-rounded weights (through BOTH decode paths: the
one-launch step at one row and the launches at three rows), logits within 2e-4 relative after the prefill, and the flow within what
bf16 operand rounding explains (x1.25).  Margins and errors go to gpurun_out/r3_heavytail.json."""
import json
import os

import numpy as np
import pytest
from _bars import bar
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'gpurun_out')


def _record(name, **vals):
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, 'r3_heavytail.jsonl'), 'a') as f:
        f.write(json.dumps(dict(test=name, **vals)) + '\n')


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'needs a GPU'
    return torch.device('cuda:0')


@pytest.fixture(scope='module')
def heavy_llm(dev):
    from cv2amd import synth, weights as W
    from cv2amd.llm import LLMEngine
    sd = synth.heavy_tail_llm(synth.make_llm(layers=24))
    return W.round_llm_sd(sd), LLMEngine(sd, dev, max_seqs=4, max_pos=512, max_out=128)


def test_heavy_tailed_llm_activations_are_heavy_tailed(heavy_llm):
    """The fixture does what it says: in the oracle's residual stream the massive channels are >= 50x the median channel, and the
    RMSNorm gains span [0.05, 30]."""
    from cv2amd import synth
    from oracle import llm as OL
    sdr, _ = heavy_llm
    inp = synth.synthetic_inputs(seed=11, text_len=8, prompt_len=10, prompt_text_len=2)
    d = OL.LLMDims(sdr)
    x = OL.build_lm_input(sdr, inp['text'], inp['prompt_text'], inp['prompt_token'])
    taps = []
    orig = OL.rmsnorm if hasattr(OL, "rmsnorm") else None
    if orig is None:
        pytest.skip("oracle has no rmsnorm hook")

    def tap(x_, w, eps):
        taps.append(x_.detach().abs().amax(dim=0))
        return orig(x_, w, eps)
    OL.rmsnorm = tap
    try:
        OL.qwen2_step(sdr, d, x, [None] * d.layers)
    finally:
        OL.rmsnorm = orig
    late = torch.stack(taps[4:]).amax(dim=0)
    ratio = float(late.max() / late.median())
    _record('llm_activation_spread', max_over_median=ratio)
    assert ratio > 50.0
    gains = torch.cat([v.abs() for k, v in sdr.items() if 'layernorm.weight' in k])
    assert float(gains.min()) <= 0.06 and float(gains.max()) >= 25.0


@pytest.mark.parametrize('n_req', [1, 2, 3, 4])          # one-row k_step; one pair (k_step2); three chains (k_step<true>); two pairs
def test_heavy_tailed_llm_greedy_ids_vs_oracle(heavy_llm, n_req):
    from cv2amd import synth
    from oracle import llm as OL
    sdr, eng = heavy_llm
    reqs = []
    for i in range(n_req):
        inp = synth.synthetic_inputs(seed=300 + i, text_len=20 + 3 * i, prompt_len=60 + 5 * i, prompt_text_len=6)
        reqs.append((inp['text'], inp['prompt_text'], inp['prompt_token']))
    got = eng.generate(reqs, force_len=60)
    for i, (req, ids) in enumerate(zip(reqs, got)):
        want, logps = OL.inference(sdr, *req, force_len=60, return_logp=True)
        margins = np.array([float(lp.topk(2).values[0] - lp.topk(2).values[1]) for lp in logps])
        first = next((k for k, (a, b) in enumerate(zip(ids, want)) if a != b), -1)
        _record(f'llm_heavy_ids_{n_req}req_{i}', path={1: 'k_step', 2: 'k_step2 (one pair)', 3: 'k_step<true> (three chains)'}.get(n_req, 'k_step2'), min_margin=float(margins.min()),
                median_margin=float(np.median(margins)), first_difference=first)
        assert ids == want, f'request {i}: first difference at step {first}, margin there {margins[first]:.2e}, min margin {margins.min():.2e}'


def test_heavy_tailed_llm_logits_after_prefill(heavy_llm, dev):
    from cv2amd import synth
    from oracle import llm as OL
    import torch.nn.functional as F
    sdr, eng = heavy_llm
    inp = synth.synthetic_inputs(seed=77, text_len=25, prompt_len=70, prompt_text_len=5)
    x = eng.build_lm_input(inp['text'], inp['prompt_text'], inp['prompt_token'])
    eng.add_request(0, x, 10, 10)
    torch.cuda.synchronize()
    got = eng.logits[(x.shape[0] - 1) % 32, :eng.vocab].cpu()
    d = OL.LLMDims(sdr)
    y = OL.qwen2_step(sdr, d, OL.build_lm_input(sdr, inp['text'], inp['prompt_text'], inp['prompt_token']), [None] * d.layers)
    want = F.linear(y[-1], sdr['llm_decoder.weight'], sdr['llm_decoder.bias'])
    err = float((got - want).abs().max() / want.abs().max())
    _record('llm_heavy_prefill_logits', rel_err=err, logit_range=float(want.abs().max()))
    assert err < 2e-4, f'logits rel err {err:.3e}'


@pytest.mark.parametrize('streaming', [False, True])
def test_heavy_tailed_flow_error_is_explained_by_operand_rounding(dev, streaming):
    from cv2amd import synth
    from cv2amd.flow import FlowEngine
    from oracle import flow as OF
    sd = synth.heavy_tail_flow(synth.make_flow())
    eng = FlowEngine(sd, dev, max_utts=2, max_len=256)
    T = 150
    g = torch.Generator().manual_seed(19)
    x, mu, cond = (torch.randn(2, 80, T, generator=g) for _ in range(3))
    spks = torch.randn(2, 80, generator=g)
    t = torch.full((2,), 0.41)
    y = eng.forward_estimator(x.to(dev).contiguous(), torch.ones(2, 1, T, device=dev), mu.to(dev), t.to(dev), spks.to(dev), cond.to(dev), streaming).cpu()
    ref32 = OF.estimator(sd, x, torch.ones(2, 1, T), mu, t, spks, cond, streaming)
    with OF.rounded_operands():
        refr = OF.estimator(sd, x, torch.ones(2, 1, T), mu, t, spks, cond, streaming)
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())           # noqa: E731
    mrel = lambda a, b: float((a - b).abs().mean() / b.abs().mean())        # noqa: E731
    hm, hn, rm, rn = rel(y, ref32), mrel(y, ref32), rel(refr, ref32), mrel(refr, ref32)
    _record('flow_heavy_estimator_' + ('chunk' if streaming else 'full'), hip_vs_fp32_max=hm, hip_vs_fp32_mean=hn, rounded_vs_fp32_max=rm,
            rounded_vs_fp32_mean=rn, out_range=float(ref32.abs().max()))
    assert torch.isfinite(y).all()
    # HIP-vs-fp32 within 1.25 x (mean) / 1.3 x (max) of what bf16 operand rounding alone produces in the oracle (measured round 4: full context
    # 0.94 / 1.00, chunk masks 1.14 / 1.29; round 5: 0.99 / 0.83 and 1.10 / 1.18 -- the max is one element of a field this checkpoint
    # amplifies, and it moves with every re-association).  What the max bar certifies is therefore ONE set of summation orders
    # (csrc/gemm.h, csrc/flow.hip): FF2 as four chains of 8 k-steps combined pairwise (k_tail_panel / k_tail_rows2, every split), the O
    # projection and FF1 as one chain over K, the attention's key tiles in ascending order with the four key groups of a one-utterance
    # block merged in group order, LayerNorm sums over a row by the 16-lane tree of the epilogue, activations with v_rcp (round 5).
    # A change to any of them needs this bar re-measured (profiles/r5_bars.jsonl holds the values), not loosened blindly.
    bar(f'heavy-tail flow ({"chunk" if streaming else "full"}): HIP-vs-fp32 max / rounded-operand max', hm / rm, 1.3)
    bar(f'heavy-tail flow ({"chunk" if streaming else "full"}): HIP-vs-fp32 mean / rounded-operand mean', hn / rn, 1.25)
    assert hn < 1.25 * rn and hm < 1.3 * rm, f'HIP vs fp32 (max {hm:.3e}, mean {hn:.3e}) exceeds what operand rounding explains (max {rm:.3e}, mean {rn:.3e})'
