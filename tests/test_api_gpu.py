"""GPU tests of the drop-in API layer and the synthesis scheduler (cosyvoice/cli/model.py of this build) against the
reference's scheduling logic restated on the CPU oracle.  Run with -m gpu on an MI355X."""
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TEXT_IDS = {'bonjour': [11, 22, 33, 44, 55, 66, 77, 88], 'guten tag': [9, 8, 7, 6, 5, 4, 3, 2, 1, 10, 20, 30]}


@pytest.fixture(scope='module')
def cv():
    assert torch.cuda.is_available(), 'needs a GPU'
    from cv2amd import synth
    from cosyvoice.cli.cosyvoice import CosyVoice2
    from cosyvoice.cli.frontend import PrecomputedFrontEnd
    inp = synth.synthetic_inputs(prompt_len=37, prompt_text_len=4)
    spk = {'prompt_text': inp['prompt_text'], 'prompt_text_len': torch.tensor([4]), 'llm_prompt_speech_token': inp['prompt_token'],
           'llm_prompt_speech_token_len': torch.tensor([37]), 'flow_prompt_speech_token': inp['prompt_token'],
           'flow_prompt_speech_token_len': torch.tensor([37]), 'prompt_speech_feat': inp['prompt_feat'],
           'prompt_speech_feat_len': torch.tensor([74]), 'llm_embedding': inp['embedding'], 'flow_embedding': inp['embedding']}
    fe = PrecomputedFrontEnd(lambda t: TEXT_IDS.get(t.rstrip('.'), [1, 2, 3]), {'fr': spk})     # split_paragraph closes a segment with '.'
    m = CosyVoice2('unused', final=True, frontend=fe, state_dicts=(synth.make_llm(layers=2), synth.make_flow(), synth.make_hift()))
    m.model.sampling_mode = 0            # harness-defined greedy: deterministic tokens
    return m


def test_non_stream_yield_contract(cv):
    outs = list(cv.inference_zero_shot('bonjour', 'salut', None, zero_shot_spk_id='fr', stream=False))
    assert len(outs) == 1
    wav = outs[0]['tts_speech']
    assert wav.dtype == torch.float32 and wav.device.type == 'cpu' and wav.dim() == 2 and wav.shape[0] == 1
    assert wav.shape[1] % 960 == 0 and wav.shape[1] >= 960 * 16        # min_len = 2 x 8 text tokens, 2 mel frames x 480 samples per token
    assert torch.isfinite(wav).all() and wav.abs().max() <= 0.99
    assert cv.model.tts_speech_token_dict == {} and cv.model.hift_cache_dict == {}      # per-call state released (model.py:395-398)


def test_cross_lingual_drops_llm_prompt(cv):
    a = list(cv.inference_cross_lingual('bonjour', None, zero_shot_spk_id='fr'))[0]['tts_speech']
    b = list(cv.inference_zero_shot('bonjour', 'salut', None, zero_shot_spk_id='fr'))[0]['tts_speech']
    assert a.shape[1] > 0 and (a.shape != b.shape or not torch.equal(a, b))


def test_instruct2_runs_without_llm_prompt_tokens(cv):
    """cli/cosyvoice.py:284-295 + frontend.py:533-537: the LLM sees [sos, prompt text, text, task] only, the flow keeps its prompt."""
    out = list(cv.inference_instruct2('bonjour', 'parle lentement', None, zero_shot_spk_id='fr'))
    cl = list(cv.inference_cross_lingual('bonjour', None, zero_shot_spk_id='fr'))
    zs = list(cv.inference_zero_shot('bonjour', 'salut', None, zero_shot_spk_id='fr'))
    assert len(out) == 1 and out[0]['tts_speech'].dtype == torch.float32 and torch.isfinite(out[0]['tts_speech']).all()
    assert out[0]['tts_speech'].shape[1] % 480 == 0 and out[0]['tts_speech'].shape[1] > 0
    assert len(cl) == 1 and len(zs) == 1


def test_inference_sft_uses_the_stored_embedding_and_no_prompt(cv):
    """cli/cosyvoice.py:81-90 + frontend.py:485-489: `inference_sft` feeds tts() with the text and the speaker's stored embedding only -- no
    prompt text, no prompt speech tokens, an EMPTY flow prompt (the defaults of CosyVoice2Model.tts).  The LLM then sees what cross-lingual
    sees; the token count must be the oracle's, the flow and the vocoder run on a sequence without prompt frames.  An unknown speaker id is
    the reference's KeyError (a CosyVoice2 model dir ships no spk2info)."""
    from cv2amd import synth, weights as W
    from oracle import llm as OL
    inp = synth.synthetic_inputs(prompt_len=37, prompt_text_len=4)
    cv.frontend.spk2info['sft_spk'] = {'embedding': inp['embedding']}
    try:
        out = list(cv.inference_sft('bonjour', 'sft_spk', stream=False))
        assert len(out) == 1
        wav = out[0]['tts_speech']
        assert wav.dtype == torch.float32 and wav.device.type == 'cpu' and torch.isfinite(wav).all() and wav.abs().max() <= 0.99
        e0 = torch.zeros(1, 0, dtype=torch.int32)
        text = torch.tensor([TEXT_IDS['bonjour']], dtype=torch.int32)
        want = OL.inference(W.round_llm_sd(synth.make_llm(layers=2)), text, e0, e0)
        assert wav.shape[1] == 960 * len(want), (wav.shape, len(want))           # 2 mel frames x 480 samples per token, no prompt frames dropped
        chunks = list(cv.inference_sft('bonjour', 'sft_spk', stream=True))
        assert sum(c['tts_speech'].shape[1] for c in chunks) == wav.shape[1]
        with pytest.raises(KeyError):
            list(cv.inference_sft('bonjour', 'nobody'))
    finally:
        cv.frontend.spk2info.pop('sft_spk', None)
    assert cv.model.tts_speech_token_dict == {} and cv.model.hift_cache_dict == {}


def test_streaming_scheduler_matches_reference_logic(cv):
    """Chunking / caches / cross-fade of model.py:300-381 restated on the CPU oracle, fed with the SAME flow mels and the same
    injected noise the device run used: waveform chunks must agree to 1e-3; the flow mels themselves are checked against the
    oracle flow (bf16 tolerance) in test_flow_gpu.py."""
    from cv2amd import synth
    from oracle import hift as OH
    mdl = cv.model
    gen = torch.Generator().manual_seed(5)
    noises = []

    def hook(T):
        nz = torch.randn(1, 480 * T, 9, generator=gen)
        noises.append(nz)
        return nz
    mdl._noise_hook, mdl._trace = hook, []
    try:
        chunks = [o['tts_speech'] for o in cv.inference_zero_shot('guten tag', 'salut', None, zero_shot_spk_id='fr', stream=True)]
        trace = mdl._trace
    finally:
        mdl._noise_hook, mdl._trace = None, None
    assert len(chunks) >= 2 and len(chunks) == len(trace)
    P, hop, la = 37, 25, 3
    pad = int(np.ceil(P / hop) * hop - P)
    offs = [t[1] for t in trace]
    assert offs[0] == 0 and offs[1] == hop + pad and all(b - a == hop for a, b in zip(offs[1:-1], offs[2:]))
    assert [t[2] for t in trace] == [False] * (len(trace) - 1) + [True]
    # CPU restatement of token2wav's tail (model.py:311-334)
    hsd = synth.make_hift()
    win = torch.from_numpy(np.hamming(2 * 3840)).float()
    cache, ref_chunks = None, []
    ri = torch.zeros(1, 9)
    for (mel, off, fin, nz, _uuid) in trace:
        mel = mel[:, :, off * 2:]
        if cache is not None:
            mel = torch.cat([cache['mel'], mel], dim=2)
            cs = cache['source']
        else:
            cs = torch.zeros(1, 1, 0)
        speech, src = OH.inference(hsd, mel, cs, ri, nz)
        if cache is not None:
            speech[..., :3840] = speech[..., :3840] * win[:3840] + cache['speech'][..., -3840:] * win[3840:]
        if not fin:
            cache = {'mel': mel[:, :, -8:], 'source': src[:, :, -3840:], 'speech': speech[:, -3840:]}
            speech = speech[:, :-3840]
        ref_chunks.append(speech)
    for i, (a, b) in enumerate(zip(chunks, ref_chunks)):
        assert a.shape == b.shape, f'chunk {i}: {a.shape} vs {b.shape}'
        assert (a - b).abs().max().item() < 1e-3, f'chunk {i}: max abs err {(a - b).abs().max().item():.3e}'
    total = sum(c.shape[1] for c in chunks)
    n_tok = trace[-1][0].shape[2] // 2
    assert total == 960 * n_tok


def test_concurrent_calls_are_coalesced(cv):
    """The evaluation harness calls one model from several threads (evaluation/cosyvoice_synthesizer.py:219,260): concurrent
    non-streaming calls must all succeed, and they run as ONE batch (ragged: two different texts) instead of one after the other."""
    ref = {t: list(cv.inference_zero_shot(t, 'salut', None, zero_shot_spk_id='fr'))[0]['tts_speech'] for t in ('bonjour', 'guten tag')}
    texts = ['bonjour', 'guten tag', 'bonjour', 'guten tag']
    res, errs = [None] * len(texts), []
    n0 = len(cv.model.batch_sizes)
    old = cv.model.coalesce_ms
    cv.model.coalesce_ms = 100.0          # generous window: the threads below start within microseconds of each other

    def work(i):
        try:
            res[i] = list(cv.inference_zero_shot(texts[i], 'salut', None, zero_shot_spk_id='fr'))[0]['tts_speech']
        except Exception as e:      # noqa: BLE001
            errs.append(e)
    try:
        ths = [threading.Thread(target=work, args=(i,)) for i in range(len(texts))]
        [t.start() for t in ths]
        [t.join(300) for t in ths]
    finally:
        cv.model.coalesce_ms = old
    assert not errs, errs
    sizes = cv.model.batch_sizes[n0:]
    assert sum(sizes) == len(texts) and max(sizes) >= 2, sizes
    for t, r in zip(texts, res):
        assert r.dtype == torch.float32 and r.device.type == 'cpu' and torch.isfinite(r).all()
        assert r.shape == ref[t].shape    # greedy tokens are deterministic; HiFT noise differs per call (device Philox)
    assert not cv.model.tts_speech_token_dict and not cv.model.llm_end_dict and not cv.model._pending


def test_stream_and_batch_calls_share_the_device(cv):
    """A streaming call holds the device while it runs; a concurrent non-streaming call waits and then completes."""
    out = {}

    def streamer():
        out['s'] = [c['tts_speech'] for c in cv.inference_zero_shot('guten tag', 'salut', None, zero_shot_spk_id='fr', stream=True)]

    def batcher():
        out['b'] = list(cv.inference_zero_shot('bonjour', 'salut', None, zero_shot_spk_id='fr'))[0]['tts_speech']
    ths = [threading.Thread(target=streamer), threading.Thread(target=batcher)]
    [t.start() for t in ths]
    [t.join(300) for t in ths]
    assert len(out['s']) >= 1 and all(torch.isfinite(c).all() for c in out['s']) and torch.isfinite(out['b']).all()


def test_concurrent_streams_share_decode_steps(cv):
    """BASELINE config 5 runs several streams at once: each streaming call owns one LLM slot and the calls share the decode steps
    (chunks interleave).  Every stream must deliver exactly its own audio: greedy tokens are deterministic, so each stream's total
    length equals the length of the same text synthesised alone."""
    import time
    texts = ['bonjour', 'guten tag', 'bonjour', 'guten tag']
    alone = {t: sum(c['tts_speech'].shape[1] for c in cv.inference_zero_shot(t, 'salut', None, zero_shot_spk_id='fr', stream=True))
             for t in set(texts)}
    out, first, errs = [None] * len(texts), [None] * len(texts), []
    t0 = time.perf_counter()

    def work(i):
        try:
            chunks = []
            for c in cv.inference_zero_shot(texts[i], 'salut', None, zero_shot_spk_id='fr', stream=True):
                if first[i] is None:
                    first[i] = time.perf_counter() - t0
                chunks.append(c['tts_speech'])
            out[i] = chunks
        except Exception as e:      # noqa: BLE001
            errs.append(e)
    ths = [threading.Thread(target=work, args=(i,)) for i in range(len(texts))]
    [t.start() for t in ths]
    [t.join(300) for t in ths]
    assert not errs, errs
    for t, chunks in zip(texts, out):
        assert len(chunks) >= 1 and all(torch.isfinite(c).all() for c in chunks)
        assert sum(c.shape[1] for c in chunks) == alone[t]
    assert sorted(cv.model._slot_free) == list(range(cv.model.max_batch)) and not cv.model._active_slots
    assert not cv.model.tts_speech_token_dict and not cv.model.hift_cache_dict


def test_abandoned_stream_releases_its_slot(cv):
    g = cv.inference_zero_shot('guten tag', 'salut', None, zero_shot_spk_id='fr', stream=True)
    next(g)
    g.close()                                   # consumer walks away after the first chunk
    assert sorted(cv.model._slot_free) == list(range(cv.model.max_batch))
    again = list(cv.inference_zero_shot('bonjour', 'salut', None, zero_shot_spk_id='fr'))
    assert torch.isfinite(again[0]['tts_speech']).all()


def test_speed_changes_length(cv):
    a = list(cv.inference_zero_shot('bonjour', 'salut', None, zero_shot_spk_id='fr', speed=1.0))[0]['tts_speech']
    b = list(cv.inference_zero_shot('bonjour', 'salut', None, zero_shot_spk_id='fr', speed=2.0))[0]['tts_speech']
    assert abs(b.shape[1] - a.shape[1] // 2) <= 480


def test_evaluation_harness_contract(cv):
    """SURVEY §8(c) last row: `CosyVoice2` driven the way `evaluation/cosyvoice_synthesizer.py:183-302` drives the reference (pattern
    restated in tests/_harness_replay.py): `_ensure_prompt_cached` registers the prompt ONCE through add_zero_shot_spk, a warm-up call
    "warmup.", a pool of 8 workers over 12 samples with the language hint, one sample whose frontend raises -> ITS row carries `error`
    and no audio while the other eleven succeed (the coalesced batch it rode in is not poisoned), result keys and per-utterance rtf
    as the pipeline computes it, every per-call state released.  Then a caller that stops waiting (`future.result(timeout)` -- the
    reference's as_completed loop only ever sees finished futures, so its FuturesTimeout row needs a waiter outside that loop): the
    abandoned call keeps its slot, finishes on its own, and a second batch runs on a model with every slot free again."""
    import math
    from concurrent.futures import ThreadPoolExecutor, TimeoutError as FuturesTimeout
    from _harness_replay import SynthesizerReplay, rtf
    fe = cv.frontend
    calls = {'extract': 0}
    real_zero_shot = fe.frontend_zero_shot

    def zero_shot(tts_text, prompt_text, prompt_speech_16k, resample_rate, zero_shot_spk_id):
        if zero_shot_spk_id == '':             # "feature extraction" of the prompt audio: the pre-extracted dict, counted
            calls['extract'] += 1
            return real_zero_shot(tts_text, prompt_text, prompt_speech_16k, resample_rate, 'fr')
        return real_zero_shot(tts_text, prompt_text, prompt_speech_16k, resample_rate, zero_shot_spk_id)
    real_tok = fe.tokenize

    def tokenize(t):
        if 'kaputt' in t:
            raise KeyError('tokenizer: unknown piece in ' + t)
        return real_tok(t.replace('<|fr|><|endofprompt|> ', ''))
    fe.frontend_zero_shot, fe.tokenize = zero_shot, tokenize
    try:
        h = SynthesizerReplay(cv, prompt_speech=torch.zeros(1, 16000))
        cfg = {'method': 'zero_shot', 'prompt_text': 'salut', 'zero_shot_spk_id': 'eval_cached_prompt', 'warmup': True, 'workers': 8,
               'timeout_s': 30, 'add_language_hint': True, 'language': 'fr', 'text_frontend': False, 'speed': 1.0}
        texts = ['bonjour', 'guten tag', 'bonjour', 'autre', 'guten tag', 'kaputt', 'bonjour', 'guten tag', 'autre', 'bonjour', 'guten tag', 'autre']
        samples = [{'utterance_id': f'utt_{i:03d}', 'text': t} for i, t in enumerate(texts)]
        rows = h.synthesize_batch(samples, cfg)
        assert calls['extract'] == 1 and h.cached_spk_id == 'eval_cached_prompt' and 'eval_cached_prompt' in cv.list_available_spks()
        assert [r['utterance_id'] for r in rows] == [s['utterance_id'] for s in samples]
        for i, r in enumerate(rows):
            if texts[i] == 'kaputt':
                assert set(r) == {'utterance_id', 'audio_tensor', 'audio_path', 'sample_rate', 'synthesis_time', 'error'}
                assert r['audio_tensor'] is None and r['sample_rate'] is None and r['synthesis_time'] == 0.0 and 'unknown piece' in r['error']
                assert math.isnan(rtf(r))
            else:
                assert set(r) == {'utterance_id', 'audio_tensor', 'audio_path', 'sample_rate', 'synthesis_time'}
                w = r['audio_tensor']
                assert w.dtype == torch.float32 and w.device.type == 'cpu' and w.shape[0] == 1 and w.shape[1] % 960 == 0 and w.shape[1] > 0
                assert r['sample_rate'] == 24000 and r['synthesis_time'] > 0 and 0 < rtf(r) < 50
        # the same text gives the same audio whichever batch it rode in (greedy ids; flow / HiFT noise is per call and seeded)
        same = [r['audio_tensor'].shape for r, t in zip(rows, texts) if t == 'bonjour']
        assert len(set(same)) == 1
        m = cv.model
        assert m.tts_speech_token_dict == {} and m.hift_cache_dict == {} and len(m._slot_free) == m.max_batch and not m._active_slots
        # a waiter that gives up: its call goes on, holds its slot until it finishes, then releases it
        with ThreadPoolExecutor(max_workers=2) as ex:
            fut = ex.submit(h.synthesize_single, 'guten tag', cfg)
            try:
                fut.result(timeout=1e-4)
                timed_out = False
            except FuturesTimeout:
                timed_out = True
            others = h.synthesize_batch(samples[:4], dict(cfg, warmup=False, workers=4))
            late = fut.result(timeout=60)
        assert timed_out and late.shape[1] > 0 and all(r['audio_tensor'] is not None for r in others)
        assert m.tts_speech_token_dict == {} and len(m._slot_free) == m.max_batch and not m._active_slots
    finally:
        fe.frontend_zero_shot, fe.tokenize = real_zero_shot, real_tok
        cv.frontend.spk2info.pop('eval_cached_prompt', None)


def test_sharded_path_over_rccl_world_of_one():
    """configs[3]'s code path on the one GPU of this box: `CV2_BENCH_FORCE_SHARDED=1 bench.py --gpus 1` spawns ONE rank under
    torch.distributed.run with backend `nccl` (= RCCL), so cv2amd/shard.py's device branch runs for real: broadcast_prompt on device
    tensors, scatter_texts from a device list, 4 utterances through tts(device_output=True), gather_waves device to device and ONE
    copy to the host.  bench.py itself asserts the gathered waveforms' lengths against the forced-length rule
    (the harness pattern: evaluation/cosyvoice_synthesizer.py:219,260)."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, CV2_BENCH_FORCE_SHARDED='1', CV2_BENCH_PER_GPU='4', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'CV2_BENCH_BACKEND', 'CV2_BENCH_FAKE_SYNTH'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '1', '--steps', '1', '--warmup', '1'], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out['backend'].startswith('nccl = RCCL') and out['ranks_seen'] == 1 and out['n_gpus'] == 1
    assert 'configs[3]' in out['config']['workload'] and out['config']['batch_per_gpu'] == 4 and out['data'] == 'synthetic'
    assert out['per_rank'][0]['utts_per_step'] == 4 and out['per_rank'][0]['device'].startswith('cuda:') and out['value'] > 0
