"""GPU parity of stage 2 (csrc/flow.hip + gemm.h through the C ABI) against the CPU oracle and the golden vectors
captured from the real reference (tests/golden/make_golden.py).  Run with -m gpu on an MI355X.

Tolerances: the HIP path multiplies bf16 operands (weights and activations rounded to bf16, fp32 accumulate, fp32
softmax / LayerNorm / residual stream), the oracle is fp32.  Per GEMM the rounding error is ~2^-9 relative per operand;
over the 56-block estimator it accumulates to ~1e-2 of the output range, so whole-network checks use
max|err| <= 2e-2 * max|ref| (measured 1.2e-2 .. 1.5e-2; every bar goes through tests/_bars.py, which records the measured value
beside its limit in gpurun_out/bars.jsonl) and a mean-relative bound, while single-GEMM checks against fp64 on the SAME rounded
operands are tight (1e-5).  The reference's own precedent for this network is rtol 1e-2 (bin/export_onnx.py:133).
"""
import ctypes as C

import numpy as np
import pytest
from _bars import bar
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'needs a GPU'
    return torch.device('cuda:0')


@pytest.fixture(scope='module')
def flow_sd():
    from cv2amd import synth
    return synth.make_flow()


@pytest.fixture(scope='module')
def eng(dev, flow_sd):
    from cv2amd.flow import FlowEngine
    return FlowEngine(flow_sd, dev, max_utts=4, max_len=512)


def rel(got, ref):
    return (got - ref).abs().max().item() / max(ref.abs().max().item(), 1e-9)


def test_gemm_bf16_matches_fp64(dev):
    from cv2amd import lib as L, weights as W
    from cv2amd import flow as F
    lib = L.lib()
    F._bind(lib)
    g = torch.Generator().manual_seed(0)
    # the last three shapes reach the large-grid tile configurations (k_gemm<64,256,2,4>, <128,128,2,4>, <128,128,4,4>) that the
    # one-utterance tests never select; (256, 256, 256) and (128, 256, 768) go through the row-panel kernel
    for m, n, k in ((128, 128, 64), (256, 256, 256), (384, 1536, 256), (128, 256, 768), (256, 1024, 1024), (128, 512, 2560),
                    (16384, 256, 512), (6400, 1024, 256), (3072, 1536, 256)):
        a = torch.randn(m, k, generator=g)
        w = torch.randn(n, k, generator=g) / k ** 0.5
        b = torch.randn(n, generator=g)
        ab = a.to(torch.bfloat16).to(dev)
        wp = W.pack_bf16(w.to(dev))
        bd = b.to(dev)
        out = torch.full((m, n), float('nan'), device=dev)
        L.check(lib.cv2_gemm_bf16(ab.data_ptr(), k, wp.data_ptr(), bd.data_ptr(), out.data_ptr(), n, m, n, k, L.stream_ptr()))
        torch.cuda.synchronize()
        ref = ab.cpu().double() @ W.bf16_round(w).double().T + b.double()
        err = (out.cpu().double() - ref).abs().max().item()
        assert err < 2e-5 * (1 + ref.abs().max().item()), f'm={m} n={n} k={k} err={err:.3e}'


@pytest.mark.parametrize('T', [16, 50, 101])
@pytest.mark.parametrize('tag,streaming', [('full', False), ('chunk', True)])
def test_estimator_vs_reference_golden(golden, eng, dev, T, tag, streaming):
    gd = golden('flow_estimator.npz')
    g = torch.Generator().manual_seed(100 + T)
    x = torch.randn(2, 80, T, generator=g)
    mu = torch.randn(2, 80, T, generator=g)
    mu[1] = 0
    cond = torch.randn(2, 80, T, generator=g)
    cond[1] = 0
    spks = torch.randn(2, 80, generator=g)
    spks[1] = 0
    xd = x.to(dev).contiguous()
    y = eng.forward_estimator(xd, torch.ones(2, 1, T, device=dev), mu.to(dev), torch.full((2,), 0.3, device=dev), spks.to(dev),
                              cond.to(dev), streaming)
    torch.cuda.synchronize()
    ref = torch.from_numpy(gd[f'y_T{T}_{tag}'])
    got = y.cpu()
    assert torch.isfinite(got).all()
    # bars = what is measured (max 1.3e-2, mean 1.0e-2 of bf16 operand rounding through 56 blocks, gpurun_out/flow_rounded_operand_errors.jsonl)
    # plus margin; the x1.25 rounded-operand budget test below is the tight guard
    _record(f'estimator_golden_T{T}_{tag}', max=rel(got, ref), mean=_mrel(got, ref))
    assert rel(got, ref) < 2e-2, f'rel max err {rel(got, ref):.3e}'
    assert _mrel(got, ref) < 1.2e-2


@pytest.mark.parametrize('T', [28, 53])
def test_encoder_vs_reference_golden(golden, eng, T):
    gd = golden('flow_encoder.npz')
    g = torch.Generator().manual_seed(200 + T)
    xs = torch.randn(1, T, 512, generator=g)
    ctx = torch.randn(1, 3, 512, generator=g)
    for tag, streaming, c in (('full', False, None), ('chunk', True, None), ('chunk_ctx', True, ctx)):
        h = eng.encoder(xs, c, streaming)
        torch.cuda.synchronize()
        ref = torch.from_numpy(gd[f'h_T{T}_{tag}'])
        got = h[0, :, ::8].cpu()
        _record(f'encoder_golden_T{T}_{tag}', max=rel(got, ref), mean=_mrel(got, ref))
        assert rel(got, ref) < 2e-2, f'{tag}: rel max err {rel(got, ref):.3e}'


@pytest.mark.parametrize('tag,streaming,finalize', [('full', False, True), ('stream', True, True),
                                                    ('stream_nonfinal', True, False)])
def test_flow_inference_vs_reference_golden(golden, eng, tag, streaming, finalize):
    from cv2amd import synth
    gd = golden('flow_e2e.npz')
    inp = synth.synthetic_inputs(prompt_len=int(gd['prompt_len']))
    tok = torch.from_numpy(gd['token'])
    mel, _ = eng.inference(tok, None, inp['prompt_token'], None, inp['prompt_feat'], None, inp['embedding'], streaming, finalize)
    torch.cuda.synchronize()
    ref = torch.from_numpy(gd['mel_' + tag])
    got = mel.cpu()
    assert got.shape == ref.shape
    assert torch.isfinite(got).all()
    # 10 Euler steps through the bf16 estimator: error relative to the mel range (measured 5.8e-3 max, 7.6e-3 mean)
    _record(f'flow_e2e_golden_{tag}', max=rel(got, ref), mean=_mrel(got, ref))
    assert rel(got, ref) < 1.5e-2, f'rel max err {rel(got, ref):.3e}'
    assert _mrel(got, ref) < 1.2e-2


def test_ragged_batch_equals_single(eng):
    """Packing several utterances of different lengths changes nothing: rows are independent (masks, causal padding)."""
    from cv2amd import synth
    utts = []
    for i, (p, n) in enumerate(((20, 31), (7, 90), (33, 12))):
        inp = synth.synthetic_inputs(seed=50 + i, prompt_len=p)
        g = torch.Generator().manual_seed(i)
        utts.append(dict(token=torch.randint(0, 6561, (1, n), generator=g, dtype=torch.int32), prompt_token=inp['prompt_token'],
                         prompt_feat=inp['prompt_feat'], embedding=inp['embedding']))
    single = [eng.inference_batch([u])[0].clone() for u in utts]
    batch = eng.inference_batch(utts)
    torch.cuda.synchronize()
    for s, b in zip(single, batch):
        assert s.shape == b.shape
        assert torch.equal(s, b)


def test_repeated_shapes_replay_one_graph_bit_identically(eng):
    """cv2_flow_inference captures a call's launches into a hipGraph at the second use of a shape and replays it from then on (flow.hip,
    `launches` / FlowGraph): first call = the launches, second = capture + replay, third and fourth = replays -- with DIFFERENT inputs of
    the same shape (the callers' pointers travel through device tables, not through kernel arguments) and the results of a batch and of a
    single utterance.  Every replay must equal the direct launches bit for bit."""
    from cv2amd import synth

    def utt(seed, p, n):
        inp = synth.synthetic_inputs(seed=seed, prompt_len=p)
        g = torch.Generator().manual_seed(seed)
        return dict(token=torch.randint(0, 6561, (1, n), generator=g, dtype=torch.int32), prompt_token=inp['prompt_token'],
                    prompt_feat=inp['prompt_feat'], embedding=inp['embedding'])
    from cv2amd import lib as L
    a, b = utt(301, 21, 47), utt(302, 21, 47)                       # same shape, different content
    L.check(L.lib().cv2_flow_debug_graph(1))                        # (off by default: CV2_FLOW_GRAPH=1; measured slower, profiles/r6_flow_graph_ab.txt)
    try:
        outs = [eng.inference_batch([x])[0].clone() for x in (a, b, a, b, a)]
        torch.cuda.synchronize()
        assert torch.equal(outs[0], outs[2]) and torch.equal(outs[0], outs[4]) and torch.equal(outs[1], outs[3])
        assert not torch.equal(outs[0], outs[1])
        pair = [utt(303, 9, 60), utt(304, 30, 22)]
        first = [o.clone() for o in eng.inference_batch(pair)]
        again = [[o.clone() for o in eng.inference_batch(pair)] for _ in range(3)]
        torch.cuda.synchronize()
        for r in again:
            assert torch.equal(r[0], first[0]) and torch.equal(r[1], first[1])
        assert torch.equal(eng.inference_batch([a])[0], outs[0])     # the single-utterance graph is still there and still right
        L.check(L.lib().cv2_flow_debug_graph(0))
        assert torch.equal(eng.inference_batch([a])[0], outs[0])     # ... and equals the plain launches
    finally:
        L.check(L.lib().cv2_flow_debug_graph(-1))


def test_large_batch_tile_configurations_agree_with_single(dev, flow_sd):
    """Eight ~10 s utterances packed together select the large-grid kernels (k_gemm<64,256,2,4>, <128,128,2,4>, k_attn_est<2,4>);
    one utterance alone runs on the row-panel GEMM, the 64x128 tiles and the key-split attention that the golden-vector tests
    cover.  Same arithmetic class (bf16 operands, fp32 accumulate), different summation order: agreement to bf16 round-off
    after the 10 Euler steps, not bit equality."""
    from cv2amd import synth
    from cv2amd.flow import FlowEngine
    big = FlowEngine(flow_sd, dev, max_utts=8, max_len=1100)
    utts = []
    for i in range(8):
        inp = synth.synthetic_inputs(seed=70 + i, prompt_len=40 + 3 * i)
        g = torch.Generator().manual_seed(100 + i)
        utts.append(dict(token=torch.randint(0, 6561, (1, 440 + 7 * i), generator=g, dtype=torch.int32), prompt_token=inp['prompt_token'],
                         prompt_feat=inp['prompt_feat'], embedding=inp['embedding']))
    batch = [m.clone() for m in big.inference_batch(utts)]
    for i in (0, 3, 7):
        single = big.inference_batch([utts[i]])[0]
        torch.cuda.synchronize()
        assert single.shape == batch[i].shape and torch.isfinite(batch[i]).all()
        bar(f'flow batch-of-8 vs single, utterance {i}', rel(batch[i].cpu(), single.cpu()), 1e-2)        # measured 3.4e-3 (other tile shapes at other row counts)


def test_attention_with_dma_staged_tiles_agrees_with_the_register_staged_form(dev, flow_sd):
    """Batches of >= 4096 / 8192 rows run the estimator attention with its key / value tiles staged by LDS DMA (k_attn_est_dma<1> / <2>:
    swizzled 128-byte rows, three stages, inline-asm DMA with hand-counted waits, the keys of a tile permuted among the score rows so that
    the V^T operand is one 16-byte read).  Same products as the register-staged form (test hook cv2_flow_debug_attn_dma), another order
    inside a matrix-core k group: the mels agree to round-off amplified by the 10 Euler steps -- the class of the batch-vs-single test
    above -- in full context, under chunk masks, and in the cached streaming form (keys / values from the streams' caches, second call at
    pos0 > 0)."""
    from cv2amd import synth, lib as L
    from cv2amd.flow import FlowEngine
    big = FlowEngine(flow_sd, dev, max_utts=16, max_len=1200)

    def utt(i, n, p):
        inp = synth.synthetic_inputs(seed=170 + i, prompt_len=p)
        g = torch.Generator().manual_seed(300 + i)
        return dict(token=torch.randint(0, 6561, (1, n), generator=g, dtype=torch.int32), prompt_token=inp['prompt_token'],
                    prompt_feat=inp['prompt_feat'], embedding=inp['embedding'])

    def both(fn):
        out = []
        for on in (1, 0):
            L.check(L.lib().cv2_flow_debug_attn_dma(on))
            try:
                out.append(fn())
                torch.cuda.synchronize()
            finally:
                L.check(L.lib().cv2_flow_debug_attn_dma(-1))
        return out

    for n_utts in (8, 3, 1):                                # 8 x 2 x ~1000 rows: the 128-row query tiles; 3: the 64-row ones; 1 utterance of
        # 1 110 frames (2 304 rows = 288 blocks of the four-group form: round 5 sends full-context calls above 256 blocks to the 64-row DMA form)
        utts = [utt(i, 520 if n_utts == 1 else 430 + 11 * i, 35 + 4 * i) for i in range(n_utts)]
        for streaming in (False, True):
            a, b = both(lambda: [m.clone() for m in big.inference_batch(utts, streaming=streaming)])
            assert all(torch.isfinite(x).all() for x in a)
            bar(f'flow DMA-staged vs register-staged attention, {n_utts} utterances, {"chunk" if streaming else "full"}',
                max(rel(x.cpu(), y.cpu()) for x, y in zip(a, b)), 1e-2)
    for n_streams in (16, 10):
        P, N = 37, 90
        us = [utt(40 + i, N, P) for i in range(n_streams)]
        calls = _stream_calls(P, N)[:2]

        def run():
            caches = [big.new_cache(2 * (P + N)) for _ in range(n_streams)]
            res = []
            for n, off, fin in calls:
                cur = [dict(u, token=u['token'][:, :n]) for u in us]
                res += [g.clone() for g, _ in big.inference_chunk_batch(cur, caches, finalize=fin)]
            return res
        a, b = both(run)
        assert len(a) == 2 * n_streams and all(torch.isfinite(x).all() for x in a)
        bar(f'flow DMA-staged vs register-staged attention, {n_streams} cached streams', max(rel(x.cpu(), y.cpu()) for x, y in zip(a, b)), 1e-2)


def test_oracle_full_size_estimator_random(eng, dev, flow_sd):
    """Against the oracle itself (not only the stored vectors) at a tile-crossing length, weights bf16-rounded on both sides."""
    from oracle import flow as OF
    T = 150
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 80, T, generator=g)
    mu = torch.randn(2, 80, T, generator=g)
    cond = torch.randn(2, 80, T, generator=g)
    spks = torch.randn(2, 80, generator=g)
    t = torch.full((2,), 0.62)
    y = eng.forward_estimator(x.to(dev).contiguous(), torch.ones(2, 1, T, device=dev), mu.to(dev), t.to(dev), spks.to(dev), cond.to(dev))
    torch.cuda.synchronize()
    ref = OF.estimator(flow_sd, x, torch.ones(2, 1, T), mu, t, spks, cond, False)
    bar('flow estimator T=150 vs oracle (max of range)', rel(y.cpu(), ref), 2e-2)                    # measured 1.5e-2: the estimator family's bar


def _record(name, **vals):
    import json
    import os
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, 'flow_rounded_operand_errors.jsonl'), 'a') as f:
        f.write(json.dumps(dict(test=name, **vals)) + '\n')


def _mrel(a, b):
    return ((a - b).abs().mean() / b.abs().mean()).item()


@pytest.mark.parametrize('streaming', [False, True])
def test_estimator_error_is_explained_by_operand_rounding(eng, dev, flow_sd, streaming):
    """VERDICT r1 item 7.  Against the fp32 oracle the absolute bar has to be 4e-2 (bf16 operand rounding through 56 blocks), which
    could hide a small bug.  The oracle's rounded-operand mode rounds every matrix-product operand to bf16 where the HIP path does;
    it is a second, independent realisation of the same rounding noise (any upstream ulp flips later roundings, so the two do not
    converge to each other: measured HIP-vs-rounded 1.2e-2 == HIP-vs-fp32 1.2e-2 == rounded-vs-fp32 1.4e-2).  What CAN be held
    tightly is the magnitude: the HIP path's distance from the fp32 oracle must not exceed the distance that operand rounding
    alone produces (x1.25), max and mean.  All three distances are recorded in gpurun_out/flow_rounded_operand_errors.jsonl."""
    from oracle import flow as OF
    T = 150
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 80, T, generator=g)
    mu = torch.randn(2, 80, T, generator=g)
    cond = torch.randn(2, 80, T, generator=g)
    spks = torch.randn(2, 80, generator=g)
    t = torch.full((2,), 0.62)
    y = eng.forward_estimator(x.to(dev).contiguous(), torch.ones(2, 1, T, device=dev), mu.to(dev), t.to(dev), spks.to(dev), cond.to(dev), streaming).cpu()
    ref32 = OF.estimator(flow_sd, x, torch.ones(2, 1, T), mu, t, spks, cond, streaming)
    with OF.rounded_operands():
        refr = OF.estimator(flow_sd, x, torch.ones(2, 1, T), mu, t, spks, cond, streaming)
    hip_max, hip_mean, rnd_max, rnd_mean = rel(y, ref32), _mrel(y, ref32), rel(refr, ref32), _mrel(refr, ref32)
    _record('estimator_T150_' + ('chunk' if streaming else 'full'), hip_vs_fp32_max=hip_max, hip_vs_fp32_mean=hip_mean, rounded_vs_fp32_max=rnd_max,
            rounded_vs_fp32_mean=rnd_mean, hip_vs_rounded_max=rel(y, refr), hip_vs_rounded_mean=_mrel(y, refr))
    assert hip_mean < 1.25 * rnd_mean and hip_max < 1.25 * rnd_max, \
        f'HIP vs fp32 (max {hip_max:.3e}, mean {hip_mean:.3e}) exceeds what bf16 operand rounding explains (max {rnd_max:.3e}, mean {rnd_mean:.3e})'


def test_flow_inference_error_is_explained_by_operand_rounding(golden, eng, flow_sd):
    from cv2amd import synth
    from oracle import flow as OF
    gd = golden('flow_e2e.npz')
    inp = synth.synthetic_inputs(prompt_len=int(gd['prompt_len']))
    tok = torch.from_numpy(gd['token'])
    mel, _ = eng.inference(tok, None, inp['prompt_token'], None, inp['prompt_feat'], None, inp['embedding'], False, True)
    got = mel.cpu()
    with OF.rounded_operands():
        refr = OF.inference(flow_sd, tok, inp['prompt_token'], inp['prompt_feat'], inp['embedding'], False, True)
    ref32 = torch.from_numpy(gd['mel_full'])                     # the reference's own output
    hip_max, hip_mean, rnd_max, rnd_mean = rel(got, ref32), _mrel(got, ref32), rel(refr, ref32), _mrel(refr, ref32)
    _record('flow_e2e_full', hip_vs_fp32_max=hip_max, hip_vs_fp32_mean=hip_mean, rounded_vs_fp32_max=rnd_max, rounded_vs_fp32_mean=rnd_mean,
            hip_vs_rounded_max=rel(got, refr), hip_vs_rounded_mean=_mrel(got, refr))
    assert hip_mean < 1.25 * rnd_mean and hip_max < 1.25 * rnd_max, \
        f'HIP vs reference (max {hip_max:.3e}, mean {hip_mean:.3e}) exceeds what bf16 operand rounding explains (max {rnd_max:.3e}, mean {rnd_mean:.3e})'


def _stream_calls(P, N, hop=25, la=3):
    """Token counts of the flow calls of one streaming utterance (cli/model.py:351-381): (n_tokens_given, token_offset, finalize)."""
    pad = -(-P // hop) * hop - P
    calls, off = [], 0
    while True:
        this = hop + pad if off == 0 else hop
        if N - off >= this + la:
            calls.append((off + this + la, off, False))
            off += this
        else:
            break
    calls.append((N, off, True))
    return calls


def test_cached_chunks_equal_the_recompute_of_the_whole_prefix(eng):
    """cv2_flow_inference_chunk (per-stream K/V + conv-tail cache) against the reference's scheme, which re-runs flow.inference over the
    whole prefix for every chunk and keeps mel[:, :, token_offset*2:] (cli/model.py:300-311).  Same kernels and arithmetic class, other
    row counts (so other tile shapes / summation orders): agreement to bf16 round-off, the bound of the batch-vs-single test above."""
    from cv2amd import synth
    P, N = 37, 150
    inp = synth.synthetic_inputs(seed=91, prompt_len=P)
    tok = torch.randint(0, 6561, (1, N), generator=torch.Generator().manual_seed(5), dtype=torch.int32)
    calls = _stream_calls(P, N)
    assert len(calls) >= 5 and calls[-1][2]
    cache = eng.new_cache(2 * (P + N))
    worst = 0.0
    for n, off, fin in calls:
        u = dict(token=tok[:, :n], prompt_token=inp['prompt_token'], prompt_feat=inp['prompt_feat'], embedding=inp['embedding'])
        ref = eng.inference_batch([u], streaming=True, finalize=fin)[0][:, :, 2 * off:].clone()
        (got, first), = eng.inference_chunk_batch([u], [cache], finalize=fin)
        torch.cuda.synchronize()
        assert first == 2 * off, (first, off)
        assert got.shape == ref.shape and torch.isfinite(got).all()
        worst = max(worst, rel(got.cpu(), ref.cpu()))
        bar(f'flow cached chunk vs recompute, offset {off}', rel(got.cpu(), ref.cpu()), 2e-3)        # measured 0.0 (bit-identical)
    assert cache.n_cached == 2 * (P + N) and cache.gen == len(calls)
    # a cache that sits out some calls (the scheduler only uses it when that pays) stays valid for the frames it holds: the next
    # cached call computes everything after them
    lazy = eng.new_cache(2 * (P + N))
    for k, (n, off, fin) in enumerate(calls[:-1]):
        if k not in (1, 4):
            continue
        u = dict(token=tok[:, :n], prompt_token=inp['prompt_token'], prompt_feat=inp['prompt_feat'], embedding=inp['embedding'])
        full = eng.inference_batch([u], streaming=True, finalize=fin)[0].clone()
        before = lazy.n_cached
        (got, first), = eng.inference_chunk_batch([u], [lazy], finalize=fin)
        torch.cuda.synchronize()
        assert first == max(before - 2 * P, 0) and first <= 2 * off and got.shape[2] == full.shape[2] - first
        bar('flow cached chunk vs full recompute (regrown cache)', rel(got.cpu(), full[:, :, first:].cpu()), 2e-3)
    _record('cached_chunks_vs_recompute', worst_rel=worst, calls=len(calls))


def test_cached_chunks_of_streams_in_different_phases_share_a_batch(eng):
    """Three streams whose calls are batched although one is at its first chunk, one in the middle and one just started later: every
    stream gets what it gets alone; a repeated call (same gen, as after a failure elsewhere in the batch) reproduces the result."""
    from cv2amd import synth
    specs = [(20, 120, 61), (44, 95, 62), (25, 140, 63)]
    streams = []
    for P, N, seed in specs:
        inp = synth.synthetic_inputs(seed=seed, prompt_len=P)
        tok = torch.randint(0, 6561, (1, N), generator=torch.Generator().manual_seed(seed), dtype=torch.int32)
        streams.append(dict(inp=inp, tok=tok, calls=_stream_calls(P, N), P=P, N=N))

    def utt(st, n):
        return dict(token=st['tok'][:, :n], prompt_token=st['inp']['prompt_token'], prompt_feat=st['inp']['prompt_feat'],
                    embedding=st['inp']['embedding'])
    # alone
    alone = []
    for st in streams:
        c = eng.new_cache(2 * (st['P'] + st['N']))
        outs = []
        for n, off, fin in st['calls']:
            (m, first), = eng.inference_chunk_batch([utt(st, n)], [c], finalize=fin)
            outs.append(m.clone())
        alone.append(outs)
    # batched, stream k starts k rounds late; non-final calls only share a batch with non-final calls
    caches = [eng.new_cache(2 * (st['P'] + st['N'])) for st in streams]
    pos = [0, 0, 0]
    rnd = 0
    while any(p < len(st['calls']) for p, st in zip(pos, streams)):
        ready = [k for k, st in enumerate(streams) if rnd >= k and pos[k] < len(st['calls'])]
        for fin in (False, True):
            grp = [k for k in ready if streams[k]['calls'][pos[k]][2] == fin]
            if not grp:
                continue
            us = [utt(streams[k], streams[k]['calls'][pos[k]][0]) for k in grp]
            if rnd == 2 and not fin:                                          # a repeated call must be harmless
                saved = [(caches[k].n_cached, caches[k].gen) for k in grp]
                eng.inference_chunk_batch(us, [caches[k] for k in grp], finalize=fin)
                for k, (nc, g) in zip(grp, saved):
                    caches[k].n_cached, caches[k].gen = nc, g
            outs = eng.inference_chunk_batch(us, [caches[k] for k in grp], finalize=fin)
            torch.cuda.synchronize()
            for k, (m, first) in zip(grp, outs):
                ref = alone[k][pos[k]]
                assert m.shape == ref.shape
                bar(f'flow cached chunks of 3 streams, stream {k} call {pos[k]}', rel(m.cpu(), ref.cpu()), 2e-3)
                pos[k] += 1
        rnd += 1


def test_stream_started_from_a_prompt_cache(eng):
    """A prompt that another call has already run: its whole chunks (those whose look-ahead lies inside the prompt) are computed once
    (`prompt_cache`), copied into the new stream's cache of another capacity (`clone_cache` -> cv2_flow_cache_copy), and the stream's
    first chunk computes only the frames after them.  Every chunk equals the stream that started from an empty cache."""
    from cv2amd import synth
    P, N = 87, 120                                        # 87 - 3 = 84 -> 75 prompt tokens = 150 frames come from the prompt cache
    inp = synth.synthetic_inputs(seed=93, prompt_len=P)
    tok = torch.randint(0, 6561, (1, N), generator=torch.Generator().manual_seed(8), dtype=torch.int32)
    calls = _stream_calls(P, N)[:-1]
    pc = eng.prompt_cache(inp['prompt_token'], inp['prompt_feat'], inp['embedding'])
    assert pc is not None and pc.n_cached == 150 and pc.gen == 1
    a = eng.new_cache(2 * (P + N))
    b = eng.clone_cache(pc, 2 * (P + N) + 64)
    assert b.n_cached == 150 and b.frames != pc.frames
    for n, off, fin in calls:
        u = dict(token=tok[:, :n], prompt_token=inp['prompt_token'], prompt_feat=inp['prompt_feat'], embedding=inp['embedding'])
        (ma, fa), = eng.inference_chunk_batch([u], [a], finalize=False)
        ma = ma.clone()
        (mb, fb), = eng.inference_chunk_batch([u], [b], finalize=False)
        torch.cuda.synchronize()
        assert fa == fb and ma.shape == mb.shape
        bar(f'flow prompt-cache start vs recompute, offset {off}', rel(mb.cpu(), ma.cpu()), 2e-3)
    assert a.n_cached == b.n_cached
    assert eng.prompt_cache(inp['prompt_token'][:, :20], inp['prompt_feat'][:, :40], inp['embedding']) is None
