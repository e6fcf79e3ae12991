"""GPU parity of stage 1 (csrc/llm.hip through the C ABI) against the CPU oracle.  Run with -m gpu on an MI355X."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'needs a GPU'
    return torch.device('cuda:0')


def test_skinny_gemm_matches_fp64(dev):
    import ctypes as C
    from cv2amd import lib as L, weights as W
    lib = L.lib()
    g = torch.Generator().manual_seed(0)
    for rows, n, k in ((1, 64, 896), (3, 1152, 896), (16, 896, 896), (17, 896, 1024), (32, 6576, 896)):
        w = torch.randn(n, k, generator=g) / k ** 0.5
        x = torch.randn(rows, k, generator=g)
        b = torch.randn(n, generator=g)
        wp = W.pack_bf16(w.to(dev))
        xd, bd = x.to(dev), b.to(dev)           # keep alive: ctypes holds raw pointers only
        out = torch.full((rows, n), float('nan'), device=dev)
        L.check(lib.cv2_skinny_gemm(L.ptr(wp), L.ptr(bd), L.ptr(xd), L.ptr(out), rows, n, k, L.stream_ptr()))
        torch.cuda.synchronize()
        ref = x.double() @ W.bf16_round(w).double().T + b.double()
        err = (out.cpu().double() - ref).abs().max().item()
        # split-bf16 activations keep ~16 mantissa bits: error ~ 2^-17 * sum|w x|
        bound = 3e-5 * (x.abs().double() @ W.bf16_round(w).abs().double().T).max().item() + 1e-6
        assert err < bound, f'rows={rows} n={n} k={k} err={err:.3e} bound={bound:.3e}'


def _requests(n, seed=1986):
    from cv2amd import synth
    reqs = []
    for i in range(n):
        inp = synth.synthetic_inputs(seed=seed + i, text_len=5 + i, prompt_len=9 + 3 * i, prompt_text_len=3 if i % 2 == 0 else 0)
        ptok = inp['prompt_token'] if i % 2 == 0 else torch.zeros(1, 0, dtype=torch.int32)
        reqs.append((inp['text'], inp['prompt_text'], ptok))
    return reqs


@pytest.fixture(scope='module')
def small(dev):
    from cv2amd import synth, weights as W
    from cv2amd.llm import LLMEngine
    sd = synth.make_llm(layers=3)
    return sd, W.round_llm_sd(sd), LLMEngine(sd, dev, max_seqs=32, max_pos=512, max_out=256)


def test_greedy_ids_bit_exact_vs_oracle(small):
    """Harness-defined greedy ids: HIP path == oracle run on the same (bf16-rounded) weights, prefill + 60 steps."""
    from oracle import llm as OL
    sd, sdr, eng = small
    reqs = _requests(3)
    got = eng.generate(reqs, force_len=None, max_ratio=10)
    for (text, ptxt, ptok), ids in zip(reqs, got):
        want, logps = OL.inference(sdr, text, ptxt, ptok, max_ratio=10, return_logp=True)
        margins = [float(lp.topk(2).values[0] - lp.topk(2).values[1]) for lp in logps]
        assert ids == want, f'min top-1 margin on this fixture {min(margins):.2e}'


def test_batched_gemm_prefill_equals_chunked_prefill_ids(small):
    """cv2_llm_prefill_batch (MFMA GEMM over all prompt rows, hi/lo operand split) vs cv2_llm_prefill (32-row skinny passes):
    same greedy ids as the oracle either way, and near-identical first-step logits."""
    from oracle import llm as OL
    sd, sdr, eng = small
    reqs = _requests(4, seed=33)
    a = eng.generate(reqs, max_ratio=6, batch_prefill=True)
    la = eng.logits[:4, :eng.vocab].clone()
    b = eng.generate(reqs, max_ratio=6, batch_prefill=False)
    assert a == b
    for (text, ptxt, ptok), ids in zip(reqs, a):
        assert ids == OL.inference(sdr, text, ptxt, ptok, max_ratio=6)


def test_logits_close_to_oracle(small):
    from oracle import llm as OL
    import torch.nn.functional as F
    sd, sdr, eng = small
    text, ptxt, ptok = _requests(1)[0]
    lm_input = eng.build_lm_input(text, ptxt, ptok)
    eng.add_request(0, lm_input, 10, 50)
    torch.cuda.synchronize()
    row = (lm_input.shape[0] - 1) % 32
    got = eng.logits[row, :eng.vocab].cpu()
    d = OL.LLMDims(sdr)
    y = OL.qwen2_step(sdr, d, OL.build_lm_input(sdr, text, ptxt, ptok), [None] * d.layers)
    want = F.linear(y[-1], sdr['llm_decoder.weight'], sdr['llm_decoder.bias'])
    assert (got - want).abs().max().item() < 2e-4 * want.abs().max().item()


def test_forced_length_batch(small):
    from oracle import llm as OL
    sd, sdr, eng = small
    reqs = _requests(5, seed=7)
    got = eng.generate(reqs, force_len=24)
    for (text, ptxt, ptok), ids in zip(reqs, got):
        assert len(ids) == 24 and max(ids) < 6561
        assert ids == OL.inference(sdr, text, ptxt, ptok, force_len=24)


@pytest.mark.parametrize('max_pos', [1536, 4400])
def test_long_context_attention_splits(dev, small, max_pos):
    """Larger KV capacities change the decode attention's key splits: 128-key splits with two wave groups per block (max_pos 1536),
    512-key splits with four groups and a second round of tiles (max_pos 4400).  A 300-token prompt crosses the tile and round
    boundaries; greedy ids must still equal the oracle's."""
    from cv2amd import synth
    from cv2amd.llm import LLMEngine
    from oracle import llm as OL
    sd, sdr, _ = small
    eng = LLMEngine(sd, dev, max_seqs=2, max_pos=max_pos, max_out=64)
    reqs = []
    for i, plen in enumerate((300, 41)):
        inp = synth.synthetic_inputs(seed=400 + i, text_len=9, prompt_len=plen, prompt_text_len=3)
        reqs.append((inp['text'], inp['prompt_text'], inp['prompt_token']))
    got = eng.generate(reqs, force_len=12)
    for (text, ptxt, ptok), ids in zip(reqs, got):
        assert ids == OL.inference(sdr, text, ptxt, ptok, force_len=12)


def test_one_launch_step_at_a_long_context(dev, small):
    """The one-launch step (k_step) far into the cache: a 1500-token prompt = 12 attention tiles of 128 keys per kv head, merged by the O
    role in three sweeps of four; the new token's own (q.k, v) partial is folded first.  Greedy ids == the oracle's, and the engine
    really takes the one-launch step."""
    from cv2amd import synth
    from cv2amd.llm import LLMEngine
    from oracle import llm as OL
    sd, sdr, _ = small
    eng = LLMEngine(sd, dev, max_seqs=1, max_pos=2048, max_out=64)
    assert eng.lib.cv2_llm_one_launch_step(eng.handle) == 1
    inp = synth.synthetic_inputs(seed=77, text_len=9, prompt_len=1500, prompt_text_len=3)
    req = (inp['text'], inp['prompt_text'], inp['prompt_token'])
    got = eng.generate([req], force_len=14)
    assert got[0] == OL.inference(sdr, *req, force_len=14)


def test_one_launch_step_with_more_head_blocks_than_layer_blocks(dev, monkeypatch):
    """Small dims the one-launch gate accepts (hidden 256, inter 512, 4 / 2 heads: ~100 blocks per layer) with the 6564-id vocabulary
    (411 head blocks): the head's blocks follow the layers' in the grid and must index their logits rows by their distance from the
    last layer block, not modulo the per-layer count.  ids of k_step == the launches (CV2_LLM_CHAIN=0) == the oracle; every logit row
    of the last step agrees between the two paths."""
    from cv2amd import synth, weights as W
    from cv2amd.llm import LLMEngine
    from oracle import llm as OL
    sd = synth.make_llm(layers=2, hidden=256, inter=512, n_q=4, n_kv=2, vocab=2048)
    sdr = W.round_llm_sd(sd)
    inp = synth.synthetic_inputs(seed=5, text_len=7, prompt_len=20, prompt_text_len=3)
    req = (inp['text'] % 2048, inp['prompt_text'] % 2048, inp['prompt_token'])
    want = OL.inference(sdr, *req, force_len=20)
    eng = LLMEngine(sd, dev, max_seqs=2, max_pos=256, max_out=64)
    assert eng.lib.cv2_llm_one_launch_step(eng.handle) == 1
    got = eng.generate([req], force_len=20)[0]
    lg1 = eng.logits[0, :eng.vocab].cpu()
    monkeypatch.setenv('CV2_LLM_CHAIN', '0')
    eng0 = LLMEngine(sd, dev, max_seqs=2, max_pos=256, max_out=64)
    assert eng0.lib.cv2_llm_one_launch_step(eng0.handle) == 0
    got0 = eng0.generate([req], force_len=20)[0]
    lg0 = eng0.logits[0, :eng0.vocab].cpu()
    assert got == want and got0 == want
    assert torch.isfinite(lg1).all() and (lg1 - lg0).abs().max().item() < 2e-4 * lg0.abs().max().item()


def test_hand_off_timeout_commits_nothing_and_the_steps_are_repeated_on_the_launches(dev, small, caplog):
    """A hand-off inside the one-launch step that never arrives (test hook: one Q-role block of layer 1 keeps its q values to itself):
    every wait behind it is bounded (0.2 s, csrc/chain.h), the step reports CV2_ST_ERR = 3 and k_sample commits NOTHING for it or for
    the steps enqueued behind it.  The host clears the flag, repeats the steps on the launches (the engine stays on them for a back-off period) and the
    request finishes with the ids of an undisturbed run; another engine in the process is not affected."""
    import logging
    from cv2amd import lib as L
    from cv2amd.llm import LLMEngine
    sd, sdr, _ = small
    eng = LLMEngine(sd, dev, max_seqs=2, max_pos=512, max_out=64)
    assert eng.lib.cv2_llm_one_launch_step(eng.handle) == 1
    req = _requests(1, seed=11)[0]
    want = eng.generate([req], force_len=20)[0]
    assert eng.handoff_recoveries == 0 and not eng.chain_broken
    L.check(eng.lib.cv2_llm_debug_skip_publish(eng.handle, 1, 3))
    try:
        # the prefill draws token 0; the first burst's k_step times out: state as after the prefill, error flag set
        eng.add_requests([0], [eng.build_lm_input(*req)], [(20, 20)], 0, 0, True)
        eng.step(1, 3)
        torch.cuda.synchronize()
        st = eng.state[0].cpu()
        assert int(st[L.ST_ERR]) == 3 and int(st[L.ST_STEP]) == 1 and int(st[L.ST_NOUT]) == 1 and int(st[L.ST_DONE]) == 0
        with caplog.at_level(logging.WARNING):
            got = eng.generate([req], force_len=20)[0]
    finally:
        L.check(eng.lib.cv2_llm_debug_skip_publish(eng.handle, -1, 0))
    assert got == want
    assert eng.handoff_recoveries == 1 and eng.chain_broken and any('hand-off' in r.message for r in caplog.records)
    assert eng.generate([req], force_len=20)[0] == want          # stays on the launches for now, still correct
    eng._chain_off_until = 0.0                                   # ... and returns to the one-launch step once the back-off period (30 s, doubling) is over
    assert not eng.chain_broken and eng.generate([req], force_len=20)[0] == want and eng.handoff_recoveries == 1


def test_hand_off_timeout_in_a_two_row_step_fails_no_request(dev, small):
    """The same broken hand-off inside k_step2 (two requests = one pair of rows per block): the waits behind it time out, BOTH rows'
    slots report CV2_ST_ERR = 3 and commit nothing; generate() clears the flags, repeats the steps on the launches and both requests
    finish with the ids of an undisturbed run."""
    from cv2amd import lib as L
    from cv2amd.llm import LLMEngine
    sd, sdr, _ = small
    eng = LLMEngine(sd, dev, max_seqs=2, max_pos=512, max_out=64)
    reqs = _requests(2, seed=21)
    want = eng.generate(reqs, force_len=18)
    assert eng.handoff_recoveries == 0
    L.check(eng.lib.cv2_llm_debug_skip_publish(eng.handle, 1, 2))
    try:
        eng.add_requests([0, 1], [eng.build_lm_input(*r) for r in reqs], [(18, 18)] * 2, 0, 0, True)
        eng.step(2, 2)
        torch.cuda.synchronize()
        st = eng.state[:2].cpu()
        assert st[:, L.ST_ERR].tolist() == [3, 3] and st[:, L.ST_STEP].tolist() == [1, 1] and st[:, L.ST_NOUT].tolist() == [1, 1]
        got = eng.generate(reqs, force_len=18)
    finally:
        L.check(eng.lib.cv2_llm_debug_skip_publish(eng.handle, -1, 0))
    assert got == want and eng.chain_broken and eng.handoff_recoveries >= 1


@pytest.mark.parametrize('n,q_block', [(3, 2), (9, 5), (11, 0)])
def test_hand_off_timeout_flags_every_row_of_the_launch(dev, small, n, q_block):
    """The broken hand-off in the other one-launch forms: 3 rows (k_step<true>, one chain per row) and 9 / 11 rows (k_step4: four rows per
    block, the last chain of 11 holds three).  A block that gives up flags EVERY row its chain serves (csrc/chain.h Gran::errs) -- its
    consumers see valid tags on whatever it published and would never time out themselves -- so every row reports CV2_ST_ERR = 3 and
    commits nothing (step, output count and position as after the prefill); generate() repeats the steps on the launches and all requests
    finish with the ids of an undisturbed run.  The hook's key is (layer, Q-role block) in every form."""
    from cv2amd import lib as L
    from cv2amd.llm import LLMEngine
    sd, sdr, _ = small
    eng = LLMEngine(sd, dev, max_seqs=12, max_pos=512, max_out=64)
    reqs = _requests(n, seed=60 + n)
    want = eng.generate(reqs, force_len=12)
    assert eng.handoff_recoveries == 0
    xs = [eng.build_lm_input(*r) for r in reqs]
    eng.park()
    eng.add_requests(list(range(n)), xs, [(12, 12)] * n, 0, 0, True)
    torch.cuda.synchronize()
    before = eng.state[:n].cpu().clone()
    L.check(eng.lib.cv2_llm_debug_skip_publish(eng.handle, 1, q_block))
    try:
        eng.step(n, 2)
        torch.cuda.synchronize()
        st = eng.state[:n].cpu()
        assert st[:, L.ST_ERR].tolist() == [3] * n
        for col in (L.ST_STEP, L.ST_NOUT, L.ST_POS, L.ST_DONE):
            assert st[:, col].tolist() == before[:, col].tolist(), f'state column {col} moved in a step that reported a hand-off time-out'
        got = eng.generate(reqs, force_len=12)
    finally:
        L.check(eng.lib.cv2_llm_debug_skip_publish(eng.handle, -1, 0))
    assert got == want and eng.chain_broken and eng.handoff_recoveries >= 1


@pytest.mark.parametrize('mode', ['1', '2', '3'])
def test_one_row_step_forms_give_the_same_ids(dev, small, monkeypatch, mode):
    """k_step (the default one-row step) against the round-5 experiments kept behind CV2_STEP1 (k_step1; 1: QA blocks that compute their
    query head themselves and keep the tile's cache rows in LDS, 2: gate/up blocks of two tile pairs, 3: both): every sum of a row runs in
    k_step's order in all of them, so ids AND logits are equal bit for bit (profiles/r5_decode_step_experiments.txt has their timings)."""
    from cv2amd.llm import LLMEngine, MODE_RAS
    from oracle import llm as OL
    sd, sdr, _ = small
    req = _requests(3, seed=77)[2]
    eng = LLMEngine(sd, dev, max_seqs=2, max_pos=512, max_out=64)
    ids = eng.generate([req], force_len=40)[0]
    lg = eng.logits[0, :eng.vocab].cpu().clone()
    ras = eng.generate([req], force_len=40, mode=MODE_RAS, seed=5)[0]
    assert ids == OL.inference(sdr, *req, force_len=40)
    monkeypatch.setenv('CV2_STEP1', mode)
    eng2 = LLMEngine(sd, dev, max_seqs=2, max_pos=512, max_out=64)
    ids2 = eng2.generate([req], force_len=40)[0]
    lg2 = eng2.logits[0, :eng2.vocab].cpu().clone()
    assert ids2 == ids and torch.equal(lg, lg2)
    assert eng2.generate([req], force_len=40, mode=MODE_RAS, seed=5)[0] == ras


@pytest.mark.parametrize('n', [2, 3, 4, 7, 8, 9, 11, 16, 22])
def test_one_launch_rows_equal_the_launches_and_the_oracle(small, n):
    """Decode steps of 2 .. 24 rows as ONE launch (up to 8 rows k_step2: the rows in pairs as two MFMA columns per block, an odd count
    leaves the last pair half empty; 3 rows: k_step<true>, one chain of blocks per row; from 9 rows k_step4: four columns per block, the
    last chain may hold 1 .. 3 rows) against the same steps on the launches (CV2_DECODE_SHARED)
    and the oracle: prompts of different lengths (every row has its own position, KV cache and attention-tile count), greedy and RAS.
    Per row every sum of the one-launch forms runs in the one-row kernel's order; the launches of up to 16 rows fold and normalise their
    operands the same way, those of 17 .. 32 rows (n = 22) from operands prepared once per row (another rounding of the normalised value):
    there the agreement is by margin, which these seeds have."""
    from cv2amd.llm import MODE_RAS
    from oracle import llm as OL
    sd, sdr, eng = small
    reqs = _requests(n, seed=40 + n)
    xs = [eng.build_lm_input(*r) for r in reqs]
    out = {}
    for shared in (False, True):
        for mode in (0, MODE_RAS):
            eng.park()
            eng.add_requests(list(range(n)), xs, [(24, 24)] * n, mode, 99, True)
            eng.step(n, 23, shared=shared)
            st, toks = eng.read(n)
            assert bool(st[:, 3].all()) and all(len(t) == 24 for t in toks)
            out[(shared, mode)] = toks
    assert out[(False, 0)] == out[(True, 0)] and out[(False, MODE_RAS)] == out[(True, MODE_RAS)]
    for r, ids in list(zip(reqs, out[(False, 0)]))[:3]:
        assert ids == OL.inference(sdr, *r, force_len=24)
    eng.park()


def test_many_row_decode_path(small):
    """More than 16 sequences per step take the prepared-operand kernels (k_prep + PRE variants): ids still equal the oracle's."""
    from oracle import llm as OL
    sd, sdr, eng = small
    reqs = _requests(19, seed=3)
    got = eng.generate(reqs, force_len=8)
    for (text, ptxt, ptok), ids in zip(reqs, got):
        assert ids == OL.inference(sdr, text, ptxt, ptok, force_len=8)


def test_many_row_launches_normalise_like_the_other_forms(small):
    """The 17 .. 32-row launches prepare every row's operand once (k_prep).  Since round 5 that preparation is the DEFERRED RMSNorm of every
    other form of the decode step -- planes of g . x, the consumer scales its outputs by the row's rstd -- instead of planes of the
    normalised value g . (x rstd), an extra hi / lo rounding of the operand that no other form has.  The forms still are different
    kernels (matrix-core attention up to 8 rows, the scalar one above; tile merges of the one-launch forms): the same three requests as
    rows 0 .. 2 of a 3-row and of a 20-row step, logits after 7 steps, relative to the largest logit -- measured (tools/dbg_rows_forms.py)
    3 vs 8 rows 0, vs 12 / 16 rows 4.6e-6 (the attention kernels), vs 17 .. 32 rows 4.5e-6 (6.1e-6 with the old preparation), launches vs
    one-launch forms 5.1 - 5.4e-6.  Greedy ids agree across the forms by margin (top-1 margins: profiles/r3_llm_margin_histogram.json),
    not by construction; the bar holds the many-row form to the class of the others."""
    from _bars import bar
    sd, sdr, eng = small
    reqs = _requests(20, seed=900)
    xs = [eng.build_lm_input(*r) for r in reqs]
    out = {}
    for n in (3, 20):
        eng.park()
        eng.add_requests(list(range(n)), xs[:n], [(12, 12)] * n, 0, 0, True)
        eng.step(n, 6, shared=True)
        torch.cuda.synchronize()
        st, toks = eng.read(n)
        out[n] = (toks[:3], eng.logits[:3, :eng.vocab].clone())
    assert out[3][0] == out[20][0]
    scale = out[3][1].abs().max().item()
    bar('llm 20-row vs 3-row launches, logits after 7 steps (rel to max)', (out[3][1] - out[20][1]).abs().max().item() / scale, 5.5e-6)
    eng.park()


def test_many_row_launches_with_fused_combines_are_reproducible(dev, small, monkeypatch):
    """The measured-and-rejected form of the 17 .. 32-row step, kept behind CV2_PRE_FUSE=1 for the A/B (llm.hip, run_layers_pre): 5
    launches per layer -- the attention's last split of a (row, kv head) combines the splits and leaves the O projection's operand; the
    down projection's last K-slice block of a column tile finishes the residual stream and leaves the next QKV operand (arrival counters,
    results read back with agent-coherent loads).  Prompts of different lengths (every row has its own split count).  Ids equal the
    oracle's and the default form's; two runs agree bit for bit in ids AND logits (a stale read of another block's result would not)."""
    from cv2amd.llm import LLMEngine, MODE_RAS
    from oracle import llm as OL
    sd, sdr, eng0 = small
    monkeypatch.setenv('CV2_PRE_FUSE', '1')
    eng = LLMEngine(sd, dev, max_seqs=32, max_pos=512, max_out=64)
    monkeypatch.delenv('CV2_PRE_FUSE')
    n = 30
    reqs = _requests(n, seed=700)
    xs = [eng.build_lm_input(*r) for r in reqs]
    runs = []
    for e, mode in ((eng, 0), (eng, 0), (eng, MODE_RAS), (eng0, 0), (eng0, MODE_RAS)):
        e.park()
        e.add_requests(list(range(n)), xs, [(48, 48)] * n, mode, 99, True)
        e.step(n, 47, shared=True)
        st, toks = e.read(n)
        assert bool(st[:, 3].all()) and all(len(t) == 48 for t in toks)
        runs.append((toks, e.logits[:n, :e.vocab].clone()))
    assert runs[0][0] == runs[1][0] and torch.equal(runs[0][1], runs[1][1])
    assert runs[0][0] == runs[3][0] and runs[2][0] == runs[4][0]
    for r, ids in list(zip(reqs, runs[0][0]))[::7]:
        assert ids == OL.inference(sdr, *r, force_len=48)
    eng0.park()


def test_live_row_decode_equals_lockstep_and_oracle(small):
    """Requests of one batch end at different lengths: generate() drops the finished slots from the decode rows at every poll
    (cv2_llm_decode_rows, rows != slots, passing through the 17..32-row, 2..16-row and one-row kernels).  The ids equal those of the
    lock-step run over all slots and the oracle's; RAS draws are keyed by slot, not by row."""
    from cv2amd.llm import MODE_RAS
    from oracle import llm as OL
    sd, sdr, eng = small
    reqs = _requests(19, seed=5)
    fl = [6 + (7 * b) % 23 for b in range(19)]
    got = eng.generate(reqs, force_len=fl, sync_every=4)
    assert [len(g) for g in got] == fl
    assert got == eng.generate(reqs, force_len=fl, sync_every=4, compact=False)
    for (text, ptxt, ptok), ids, n in zip(reqs, got, fl):
        assert ids == OL.inference(sdr, text, ptxt, ptok, force_len=n)
    ras = eng.generate(reqs, mode=MODE_RAS, seed=77, force_len=fl, sync_every=4)
    assert ras == eng.generate(reqs, mode=MODE_RAS, seed=77, force_len=fl, sync_every=4, compact=False)
    # slot 0 is not among the survivors here (the shortest request): the last rows run the launches, not k_step
    assert fl[0] == min(fl)


def test_ras_sampling_matches_oracle_with_same_noise(small):
    from cv2amd import philox
    from cv2amd.llm import MODE_RAS
    from oracle import llm as OL
    sd, sdr, eng = small
    reqs = _requests(2, seed=21)
    seed = 0x1234ABCD5
    got = eng.generate(reqs, mode=MODE_RAS, seed=seed, max_ratio=8)
    for b, ((text, ptxt, ptok), ids) in enumerate(zip(reqs, got)):
        want = OL.inference(sdr, text, ptxt, ptok, mode='ras', max_ratio=8,
                            uniforms=lambda step, trial: philox.uniforms(b, step, trial, seed))
        assert ids == want


def test_sampler_decisions_vs_reference_fixture(small):
    """k_sample on its own (test hook cv2_llm_debug_sample) replays tests/golden/sampler_ras.npz: logp, decoded window and ignore_eos of 64
    cases go into the engine's logits / out_tokens / state, the slot's (seed, step) are the ones the fixture's table of uniforms was made
    with (Philox: the device draws the table itself), and the id drawn -- or the 100-re-draw error -- must be what the REFERENCE's
    `TransformerLM.sampling_ids` + `ras_sampling` returned under that table (llm/llm.py:235-250, utils/common.py:111-139).  Both candidate
    selection paths (mode 1, and 5 = the extract-max path)."""
    import os
    from cv2amd import lib as L
    from cv2amd.llm import MODE_RAS
    sd, sdr, eng = small
    gd = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'sampler_ras.npz'))
    n = len(gd['top'])
    for mode in (MODE_RAS, 5):
        for i in range(n):                                    # (the tables are drawn for slot 0: one case per launch)
            step, seed, wl = int(gd['step'][i]), int(gd['seed'][i]), int(gd['window_len'][i])
            st = torch.zeros(L.STATE_STRIDE, dtype=torch.int32)
            st[L.ST_POS], st[L.ST_STEP], st[L.ST_NOUT] = 40, step, wl
            st[L.ST_MINLEN] = step + 1 if gd['ignore_eos'][i] else 0             # ignore_eos = step < min_len (llm.py:697)
            st[L.ST_MAXLEN], st[L.ST_MODE] = 10_000, mode
            st[L.ST_SEED_LO], st[L.ST_SEED_HI] = seed & 0x7FFFFFFF, 0
            toks = torch.zeros(eng.max_out, dtype=torch.int32)
            toks[:wl] = torch.from_numpy(gd['window'][i][:wl]).int()
            lg = torch.zeros(eng.logits.shape[1])
            lg[:6564] = torch.from_numpy(gd['logp'][i])
            eng.state[0].copy_(st); eng.out_tokens[0].copy_(toks); eng.logits[0].copy_(lg)
            L.check(eng.lib.cv2_llm_debug_sample(eng.handle, 1, L.stream_ptr()))
            got = eng.state[0].cpu()
            want = int(gd['top'][i])
            if want < 0:
                assert int(got[L.ST_ERR]) == 1, f'case {i}: expected the 100-re-draw error'
            else:
                assert int(got[L.ST_ERR]) == 0 and int(got[L.ST_LAST]) == want, \
                    f'case {i} (kind {int(gd["kind"][i])}, mode {mode}): device {int(got[L.ST_LAST])}, reference {want}'
                assert int(got[L.ST_STEP]) == step + 1
    eng.park()


def test_ras_candidate_selection_paths_agree(small):
    """The nucleus candidates come from a bound + rank-by-counting selection; massively tied logits fall back to 25 extract-max
    rounds.  Sampler mode 5 forces that fallback: both must draw the same ids from the same Philox stream."""
    from cv2amd.llm import MODE_RAS
    sd, sdr, eng = small
    reqs = _requests(3, seed=33)
    a = eng.generate(reqs, mode=MODE_RAS, seed=99, max_ratio=8)
    b = eng.generate(reqs, mode=5, seed=99, max_ratio=8)
    assert a == b and all(len(x) > 0 for x in a)


def test_eos_guard_raises(dev):
    """A model that always prefers EOS: the sampler re-draws 100 times then the host raises RuntimeError (llm.py:249)."""
    from cv2amd import synth
    from cv2amd.llm import LLMEngine, MODE_RAS
    sd = synth.make_llm(layers=1)
    sd['llm_decoder.bias'] = sd['llm_decoder.bias'].clone()
    sd['llm_decoder.bias'][6561] = 1e4
    eng = LLMEngine(sd, dev, max_seqs=1, max_pos=128, max_out=64)
    text, ptxt, ptok = _requests(1)[0]
    with pytest.raises(RuntimeError):
        # step 0 masks EOS, step 1 is still below min_len -> guard fires
        eng.generate([(text, ptxt, ptok)], mode=MODE_RAS, seed=1)
