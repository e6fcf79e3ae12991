"""Generate the golden fixtures in this directory from the REAL reference modules.

Run in the build container only (needs /root/reference):   python tests/golden/make_golden.py
The fixtures are DATA (inputs + the reference's outputs); weights are not stored — they are the seeded
synthetic checkpoints of cv2amd/synth.py, loaded into the reference's own classes with strict=True.

What is pinned (reference symbol -> file):
  HiFTGenerator.inference (generator.py:570-582)            hift_T24.npz, hift_T16_cache.npz
  CausalMaskedDiffWithXvec.inference (flow.py:235-283)       flow_e2e.npz   (full / streaming / streaming non-final)
  CausalConditionalDecoder.forward (decoder.py:405-494)      flow_estimator.npz  (T = 16, 50, 101; full and chunk mask)
  UpsampleConformerEncoder.forward (upsample_encoder.py:243) flow_encoder.npz    (T_tok = 28, 53; full/chunk/context)
  Qwen2LM.inference greedy (llm.py:575-719)                  llm_greedy.npz      (24 layers; zero-shot and cross-lingual)
                                                             llm_greedy_bf16w.npz (same, GEMM weights rounded to bf16 first)
  Qwen2LM.inference_bistream greedy (llm.py:721-834)          llm_bistream.npz    (24 layers; via the cache view of oracle/ref_harness.py)
  flow.inference at T=1010 + HiFTGenerator.inference on 500 frames     fullsize.npz        (BASELINE configs[1] shapes)
  nucleus_sampling candidate set (common.py:120-134)         sampler.npz
  ras_sampling + TransformerLM.sampling_ids decisions (common.py:111-139, llm.py:235-250; injected uniforms)   sampler_ras.npz
Each block also asserts that oracle/ reproduces the reference before writing.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'cosyvoice2-eu_amd'))

from cv2amd import synth            # noqa: E402
from oracle import ref_harness as R  # noqa: E402
from oracle import hift as OH, flow as OF, llm as OL   # noqa: E402


def save(name, **arrs):
    out = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()}
    np.savez_compressed(os.path.join(HERE, name), **out)
    print('wrote', name, {k: v.shape for k, v in out.items()})


def hift_noise(seed, T):
    """Injected noise with a well-defined order (the reference draws on odd strides; see oracle/hift.py)."""
    g = torch.Generator().manual_seed(seed)
    return torch.rand(1, 9, generator=g), torch.randn(1, 480 * T, 9, generator=g)


def gen_hift():
    h = R.build_hift()
    sd = synth.make_hift()
    h.load_state_dict(sd, strict=True)
    real_randn_like, real_rand = torch.randn_like, torch.rand
    for name, T, cache_len, seed in (('hift_T24.npz', 24, 0, 11), ('hift_T16_cache.npz', 16, 3840, 12)):
        g = torch.Generator().manual_seed(seed)
        mel = (torch.randn(1, 80, T, generator=g) * 2 - 4).clamp(-11.5, 2)
        cs = torch.randn(1, 1, cache_len, generator=g) * 0.1
        ri, nz = hift_noise(seed, T)
        draws = [nz, torch.zeros(1, 480 * T, 1)]
        torch.randn_like = lambda x, *a, **k: draws.pop(0)
        torch.rand = lambda *a, **k: ri.clone()
        try:
            with torch.inference_mode():
                f0 = h.f0_predictor(mel)
                wav, src = h.inference(mel, cs)
        finally:
            torch.randn_like, torch.rand = real_randn_like, real_rand
        wo, so = OH.inference(sd, mel, cs, ri, nz)
        assert torch.equal(wo, wav) and torch.equal(so, src), 'oracle != reference (hift)'
        save(name, mel=mel, cache_source=cs, noise_seed=seed, f0=f0, wav=wav, source=src)


def gen_flow():
    f = R.build_flow()
    sd = synth.make_flow()
    f.load_state_dict(sd, strict=True)
    assert torch.equal(f.decoder.rand_noise, OF.rand_noise())
    # --- end to end
    inp = synth.synthetic_inputs(prompt_len=20)
    g = torch.Generator().manual_seed(3)
    tok = torch.randint(0, 6561, (1, 31), generator=g, dtype=torch.int32)
    out = {}
    for tag, streaming, finalize in (('full', False, True), ('stream', True, True), ('stream_nonfinal', True, False)):
        with torch.inference_mode():
            mr, _ = f.inference(token=tok, token_len=torch.tensor([31]), prompt_token=inp['prompt_token'],
                                prompt_token_len=torch.tensor([20]), prompt_feat=inp['prompt_feat'],
                                prompt_feat_len=torch.tensor([40]), embedding=inp['embedding'],
                                streaming=streaming, finalize=finalize)
        mo = OF.inference(sd, tok, inp['prompt_token'], inp['prompt_feat'], inp['embedding'], streaming, finalize)
        assert (mr - mo).abs().max() < 2e-5, 'oracle != reference (flow e2e)'
        out['mel_' + tag] = mr
    save('flow_e2e.npz', token=tok, prompt_len=20, rand_noise_head=f.decoder.rand_noise[0, 0, :16],
         t_span=1 - torch.cos(torch.linspace(0, 1, 11) * 0.5 * torch.pi), **out)
    # --- estimator only
    est = f.decoder.estimator
    out = {}
    for T in (16, 50, 101):
        g = torch.Generator().manual_seed(100 + T)
        x = torch.randn(2, 80, T, generator=g)
        mu = torch.randn(2, 80, T, generator=g)
        mu[1] = 0
        cond = torch.randn(2, 80, T, generator=g)
        cond[1] = 0
        spks = torch.randn(2, 80, generator=g)
        spks[1] = 0
        t = torch.full((2,), 0.3)
        mask = torch.ones(2, 1, T)
        for tag, streaming in (('full', False), ('chunk', True)):
            with torch.inference_mode():
                yr = est(x, mask, mu, t, spks, cond, streaming=streaming)
            yo = OF.estimator(sd, x, mask, mu, t, spks, cond, streaming)
            assert (yr - yo).abs().max() < 2e-5, 'oracle != reference (estimator)'
            out[f'y_T{T}_{tag}'] = yr
    save('flow_estimator.npz', seeds='100+T', t=0.3, **out)
    # --- encoder only
    out = {}
    for T in (28, 53):
        g = torch.Generator().manual_seed(200 + T)
        xs = torch.randn(1, T, 512, generator=g)
        ctx = torch.randn(1, 3, 512, generator=g)
        for tag, streaming, c in (('full', False, None), ('chunk', True, None), ('chunk_ctx', True, ctx)):
            with torch.inference_mode():
                kw = {} if c is None else {'context': c}
                hr, _ = f.encoder(xs, torch.tensor([T]), streaming=streaming, **kw)
            ho = OF.encoder(sd, xs, c, streaming)
            assert (hr - ho).abs().max() < 2e-5, 'oracle != reference (encoder)'
            out[f'h_T{T}_{tag}'] = hr[0, :, ::8]            # every 8th channel: keeps the file small
    save('flow_encoder.npz', seeds='200+T', **out)


def gen_llm(rounded=False):
    """rounded=True: the GEMM matrices are rounded to bf16 first (cv2amd.weights.round_llm_sd), i.e. the reference runs in fp32 on
    exactly the weight values the HIP path multiplies by -> llm_greedy_bf16w.npz, which the GPU parity test consumes in full."""
    from cv2amd import weights as W
    l = R.build_llm(num_layers=24)
    sd = synth.make_llm(layers=24)
    if rounded:
        sd = W.round_llm_sd(sd)
    l.load_state_dict(sd, strict=True)
    l.sampling_ids = types.MethodType(R.greedy_sampling_ids, l)
    inp = synth.synthetic_inputs(text_len=6, prompt_len=12, prompt_text_len=4)
    e0 = torch.zeros(1, 0, dtype=torch.int32)
    out = {}
    for tag, ptxt, ptok in (('zero_shot', inp['prompt_text'], inp['prompt_token']), ('cross_lingual', e0, e0)):
        with torch.inference_mode():
            ref = list(l.inference(text=inp['text'], text_len=torch.tensor([6]), prompt_text=ptxt,
                                   prompt_text_len=torch.tensor([ptxt.shape[1]]), prompt_speech_token=ptok,
                                   prompt_speech_token_len=torch.tensor([ptok.shape[1]]), embedding=inp['embedding']))
        ids, logps = OL.inference(sd, inp['text'], ptxt, ptok, return_logp=True)
        assert ids == ref, 'oracle != reference (llm greedy ids)'
        top2 = torch.stack([lp.topk(2).values for lp in logps])
        out['ids_' + tag] = np.asarray(ref, dtype=np.int32)
        out['margin_' + tag] = (top2[:, 0] - top2[:, 1]).numpy()
        out['logp_head_' + tag] = torch.stack(logps[:3])[:, ::16].numpy()
    save('llm_greedy_bf16w.npz' if rounded else 'llm_greedy.npz', text_len=6, prompt_len=12, prompt_text_len=4, **out)


BISTREAM = dict(fill_bias=6.0, eos_bias=14.0, text_len=23, prompt_len=31, prompt_text_len=6, cuts=(0, 3, 10, 15, 23), seeds=(1, 3))


def bistream_sd(layers=24):
    """Synthetic checkpoint on which bistream decoding terminates: the fill id (6563) and EOS get a decoder bias so that they win some
    argmaxes, 6562 can never win (the reference raises ValueError for it, llm.py:809)."""
    from cv2amd import weights as W
    sd = W.round_llm_sd(synth.make_llm(layers=layers))
    b = sd['llm_decoder.bias'].clone()
    b[6563] += BISTREAM['fill_bias']
    b[6561] += BISTREAM['eos_bias']
    b[6562] = -30.0
    sd['llm_decoder.bias'] = b
    return sd


def gen_bistream():
    """Qwen2LM.inference_bistream (llm.py:721-834) of the reference itself, greedy harness sampler, 24 layers, text delivered in four
    pieces of 3 / 7 / 5 / 8 tokens, with and without prompt speech tokens."""
    l = R.enable_bistream(R.build_llm(num_layers=24))
    sd = bistream_sd()
    l.load_state_dict(sd, strict=True)
    l.sampling_ids = types.MethodType(R.greedy_sampling_ids, l)
    out = {}
    c = BISTREAM['cuts']
    for seed in BISTREAM['seeds']:
        inp = synth.synthetic_inputs(seed=seed, text_len=BISTREAM['text_len'], prompt_len=BISTREAM['prompt_len'], prompt_text_len=BISTREAM['prompt_text_len'])
        e0 = torch.zeros(1, 0, dtype=torch.int32)
        for tag, ptok in (('prompt', inp['prompt_token']), ('noprompt', e0)):
            chunks = [inp['text'][:, a:b] for a, b in zip(c[:-1], c[1:])]
            with torch.inference_mode():
                ref = list(l.inference_bistream(text=(t for t in chunks), prompt_text=inp['prompt_text'], prompt_text_len=torch.tensor([inp['prompt_text'].shape[1]]),
                                                prompt_speech_token=ptok, prompt_speech_token_len=torch.tensor([ptok.shape[1]]), embedding=inp['embedding']))
            ids, outs = OL.inference_bistream(sd, chunks, inp['prompt_text'], ptok)
            assert ids == ref, 'oracle != reference (bistream ids)'
            out[f'ids_{tag}_{seed}'] = np.asarray(ref, dtype=np.int32)
            out[f'out_tokens_{tag}_{seed}'] = np.asarray(outs, dtype=np.int32)
            print(tag, seed, len(ref), 'emitted,', outs.count(6563), 'fills')
    save('llm_bistream.npz', **{k: v for k, v in BISTREAM.items() if k != 'cuts'}, cuts=np.asarray(c), **out)


def gen_llm_bf16w():
    gen_llm(rounded=True)


def gen_fullsize():
    """BASELINE configs[1] shapes through the reference itself: flow at P=255 prompt tokens + 250 generated (T = 1010 mel frames in
    the estimator) and HiFT on the resulting 500 frames with injected noise.  Stored: the full mel, every 8th waveform / source sample."""
    inp = synth.synthetic_inputs(prompt_len=255)
    g = torch.Generator().manual_seed(5)
    tok = torch.randint(0, 6561, (1, 250), generator=g, dtype=torch.int32)
    f = R.build_flow()
    fsd = synth.make_flow()
    f.load_state_dict(fsd, strict=True)
    with torch.inference_mode():
        mel, _ = f.inference(token=tok, token_len=torch.tensor([250]), prompt_token=inp['prompt_token'],
                             prompt_token_len=torch.tensor([255]), prompt_feat=inp['prompt_feat'],
                             prompt_feat_len=torch.tensor([510]), embedding=inp['embedding'], streaming=False, finalize=True)
    mo = OF.inference(fsd, tok, inp['prompt_token'], inp['prompt_feat'], inp['embedding'], False, True)
    assert mel.shape == (1, 80, 500) and (mel - mo).abs().max() < 5e-5, 'oracle != reference (flow, T=1010)'
    del f
    h = R.build_hift()
    hsd = synth.make_hift()
    h.load_state_dict(hsd, strict=True)
    seed, T = 13, 500
    ri, nz = hift_noise(seed, T)
    real_randn_like, real_rand = torch.randn_like, torch.rand
    draws = [nz, torch.zeros(1, 480 * T, 1)]
    torch.randn_like = lambda x, *a, **k: draws.pop(0)
    torch.rand = lambda *a, **k: ri.clone()
    try:
        with torch.inference_mode():
            wav, src = h.inference(mel, torch.zeros(1, 1, 0))
    finally:
        torch.randn_like, torch.rand = real_randn_like, real_rand
    wo, so = OH.inference(hsd, mel, torch.zeros(1, 1, 0), ri, nz)
    assert torch.equal(wo, wav) and torch.equal(so, src), 'oracle != reference (hift, 500 frames)'
    save('fullsize.npz', token=tok, prompt_len=255, mel=mel[0], noise_seed=seed, wav8=wav[0, ::8], source8=src[0, 0, ::8],
         wav_absmax=wav.abs().max(), mel_absmax=mel.abs().max())


def gen_chain_rounded():
    """Scale for the chain-level waveform tolerance (tests/test_fullsize_gpu.py::test_chain_tokens_to_waveform_vs_reference_chain): the mel of
    fullsize.npz's tokens from the ORACLE flow in rounded-operand mode (every matrix-product operand rounded to bf16 where the HIP path
    rounds, everything else fp32) -- what bf16 operand rounding alone does in the reference's arithmetic.  ~100 s of CPU, hence a fixture."""
    gd = np.load(os.path.join(HERE, 'fullsize.npz'))
    inp = synth.synthetic_inputs(prompt_len=int(gd['prompt_len']))
    fsd = synth.make_flow()
    with torch.inference_mode():
        mo = OF.inference(fsd, torch.from_numpy(gd['token']), inp['prompt_token'], inp['prompt_feat'], inp['embedding'], False, True)
        assert (mo[0] - torch.from_numpy(gd['mel'])).abs().max() < 5e-5, 'oracle flow no longer reproduces the reference mel of fullsize.npz'
        with OF.rounded_operands():
            mr = OF.inference(fsd, torch.from_numpy(gd['token']), inp['prompt_token'], inp['prompt_feat'], inp['embedding'], False, True)
    save('chain_rounded.npz', mel_rounded=mr[0])


TEXT_SAMPLES = [
    "Bonjour, je m'appelle Claire et j'habite à Lyon. Aujourd'hui nous allons parler de la synthèse vocale ! Est-ce que vous êtes prêts ? "
    "La première partie concerne les modèles de langage : ils prédisent des jetons de parole; la seconde partie concerne le vocodeur. "
    "Enfin, nous verrons comment mesurer la qualité. Merci.",
    "Guten Tag. Heute sprechen wir über Sprachsynthese und darüber, wie man sie schnell macht! Ist das nicht spannend? Ja: sehr. "
    "Die Modelle sind groß; die Rechner sind schnell. Am Ende hören wir ein Beispiel.",
    'He said "stop." Then she answered "why?" and left. Nobody knew what to do next; everyone waited: ten minutes, twenty minutes, an hour',
    "短句。这是一个测试句子，用来检查分段逻辑是否正确！我们还需要更多的文字来超过长度限制；所以这里继续写一些内容、再写一些内容。最后一句",
    "One two three four five six seven eight nine ten eleven twelve thirteen fourteen fifteen sixteen seventeen eighteen nineteen twenty. " * 12,
    "...", "Ok",
]


def gen_text():
    """split_paragraph / is_only_punctuation / _split_sentences of the reference on sample paragraphs (whitespace tokenizer)."""
    import json
    import re
    from cosyvoice.utils import frontend_utils as RU
    tok = lambda t: t.split()      # noqa: E731
    cases = []
    for text in TEXT_SAMPLES:
        for lang, kw in (('en', dict(token_max_n=80, token_min_n=60, merge_len=20)), ('en', dict(token_max_n=12, token_min_n=8, merge_len=4)),
                         ('zh', dict(token_max_n=30, token_min_n=20, merge_len=8)), ('en', dict(token_max_n=12, token_min_n=8, merge_len=4, comma_split=True))):
            try:
                out = RU.split_paragraph(text, tok, lang, **kw)
            except IndexError:
                out = 'IndexError'
            cases.append(dict(text=text, lang=lang, kw=kw, out=out))
    sents = [[s.strip() for s in re.split(r'(?<=[\.\?\!\u2026\u3002\uff01\uff1f])\s+', t) if s.strip()] for t in TEXT_SAMPLES]   # frontend.py:293-294
    punct = {t: RU.is_only_punctuation(t) for t in ('...', '', 'a.', '。！', '$+', ' ', 'Ok')}
    french = {t: RU.contains_french(t) for t in TEXT_SAMPLES}
    with open(os.path.join(HERE, 'split_paragraph.json'), 'w') as f:
        json.dump(dict(cases=cases, sentences=sents, punct=punct, french=french), f, ensure_ascii=False, indent=0)
    print('wrote split_paragraph.json', len(cases))


NORM_SAMPLES = [
    "Das kostet ca. 1.234,50 € bzw. 12 % mehr, z.B. am 3. Mai bei 120 km/h & 30°C.",
    "Genau genommen seit 2020. Die Nr. 7 steht auf S. 12, d.h. u.a. im Kapitel 3.",
    "Wir haben 1 000 000 Gründe und 2.500 Ideen; insb. die 4. ist gut (siehe § 5) @ home.",
    "M. Dupont a 25 % de 300 € etc. Mme Curie habite bd. Voltaire, c-à-d. près de la pl. Monge.",
    "Bonjour, je suis le Dr. Martin et j'ai 2 chats & 1 chien : c'est 100 % vrai + ou - 5 °.",
    "I have 3 cats (and 12 dogs) in 2020. Call 555-1202 or visit room 1001b at 10:30.",
    "Version 2.0 was released on 2021-03-04 with 007 fixes and 1234567 downloads",
    "No digits here, only words `quoted` and an em dash——like this.",
    "Heute ist es schön.\nMorgen wird es 21 Grad.",
    "这是一个测试句子 with 12 numbers。",
    "Short. Two sentences here! And a third one? Yes… indeed.",
]


def gen_textnorm():
    """The dependency-free branches of the reference's text normalisation (cli/frontend.py:64-140, 293-319, 340-417, 419-480;
    utils/frontend_utils.py:57-73) on sample sentences -> text_normalize.json.  num2words / NeMo / WeTextProcessing / lingua are
    absent here exactly as they are optional there; `inflect` (a hard import of the reference, absent here) is replaced by the
    product's restatement NumberWords, so the digit-run scan of spell_out_number is pinned and number_to_words itself is not."""
    import json
    import importlib.util
    from cosyvoice.cli import frontend as RF
    from cosyvoice.utils import frontend_utils as RU
    spec = importlib.util.spec_from_file_location('amd_frontend_utils', os.path.join(ROOT, 'cosyvoice2-eu_amd', 'cosyvoice', 'utils', 'frontend_utils.py'))
    AU = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(AU)
    fe = RF.CosyVoiceFrontEnd.__new__(RF.CosyVoiceFrontEnd)
    fe.use_ttsfrd = False
    fe.en_tn_model = None
    fe.zh_tn_model = None
    fe.lid = None
    fe.nemo_norm = {}
    fe.inflect_parser = AU.NumberWords()
    fe.allowed_special = 'all'

    class Tok:
        def encode(self, t, allowed_special=None):
            return t.split()
    fe.tokenizer = Tok()
    out = dict(contains_german={}, expand_abbr_de={}, spell_de={}, symbols_de={}, spell_en={}, detect={}, normalize={}, text_normalize={})
    for t in NORM_SAMPLES:
        out['contains_german'][t] = bool(RF._fallback_contains_german(t))
        out['expand_abbr_de'][t] = RF._fallback_expand_abbreviations_german(t)
        out['spell_de'][t] = RF._fallback_spell_out_number_german(t)
        out['symbols_de'][t] = RF._fallback_replace_symbols_german(t)
        out['spell_en'][t] = RU.spell_out_number(t, fe.inflect_parser)
        out['text_normalize'][t] = fe.text_normalize(t, split=True, text_frontend=True)
        for sent in fe._split_sentences(t):
            lang = fe._detect_lang(sent)
            out['detect'][sent] = lang
            out['normalize'][sent] = fe._normalize_sentence(sent, lang)
    out['num2words_present'] = bool(RF._HAS_NUM2WORDS)
    with open(os.path.join(HERE, 'text_normalize.json'), 'w') as f:
        json.dump(out, f, ensure_ascii=False, indent=0)
    print('wrote text_normalize.json', len(NORM_SAMPLES), 'samples,', len(out['normalize']), 'sentences')


def gen_sampler():
    from cosyvoice.utils.common import nucleus_sampling
    g = torch.Generator().manual_seed(77)
    cands, probs = [], []
    logps = []
    for i in range(4):
        logp = (torch.randn(6564, generator=g) * (1.0 + i)).log_softmax(0)
        # candidate set of the reference = support of its draws; recover it by exhausting the multinomial
        seen = set()
        for _ in range(400):
            seen.add(int(nucleus_sampling(logp, 0.8, 25)))
        p, idx = OL.nucleus_candidates(logp, 0.8, 25)
        assert seen.issubset(set(idx)), 'oracle candidate set != reference support'
        c = np.full(25, -1, dtype=np.int64)
        c[:len(idx)] = idx
        cands.append(c)
        logps.append(logp.numpy())
    save('sampler.npz', logp=np.stack(logps), candidates=np.stack(cands))


def gen_sampler_ras():
    """The DECISION logic of the sampler, run through the reference's own functions: `ras_sampling` -> `nucleus_sampling` /
    `random_sampling` (utils/common.py:111-139) under `TransformerLM.sampling_ids` (llm/llm.py:235-250: EOS re-draws while
    ignore_eos, RuntimeError after 100).  Only the RNG is replaced: `Tensor.multinomial` draws by inverse CDF from a committed
    table of uniforms (first index whose float64 cumulative sum exceeds u * total) -- the nucleus draw of trial t takes
    u[t][0], the full-vocabulary re-draw of that trial (when the repetition rule fires) u[t][1] -- exactly like torch.rand /
    randn_like are replaced for HiFT above.  The table of a case is Philox4x32-10 under a stored (seed, step) -- the function k_sample
    draws with (cv2amd/philox.py) -- so the GPU test needs no injection: the device draws the same numbers by itself.  Stored per case: logp,
    the decoded window, ignore_eos, the uniforms, (seed, step), and the id the reference returned (-1: it raised the RuntimeError)."""
    from cosyvoice.utils.common import ras_sampling
    from cosyvoice.llm.llm import TransformerLM
    EOS = 6561
    lm = types.SimpleNamespace(sampling=ras_sampling, speech_token_size=EOS)
    g = torch.Generator().manual_seed(1986)
    real_multinomial = torch.Tensor.multinomial
    state = {}

    def inv_cdf_multinomial(self, num_samples, replacement=False, *, generator=None):
        assert num_samples == 1 and self.dim() == 1
        if self.numel() != 6564:                 # the nucleus draw opens a trial; the full-vocabulary draw belongs to the trial in progress
            state['trial'] += 1
        u = state['u'][state['trial'], 0 if self.numel() != 6564 else 1]
        c = torch.cumsum(self.double(), 0)
        i = int(torch.searchsorted(c, (u.double() * c[-1]).reshape(1), right=True))
        state['draws'] += 1
        return torch.tensor([min(i, self.numel() - 1)], dtype=torch.long)

    def run(logp, window, ignore_eos, u):
        state.update(trial=-1, u=u, draws=0)
        torch.Tensor.multinomial = inv_cdf_multinomial
        try:
            try:
                top = int(TransformerLM.sampling_ids(lm, logp, list(window), 25, ignore_eos))
            except RuntimeError as e:
                assert 'max_trials' in str(e)
                top = -1
        finally:
            torch.Tensor.multinomial = real_multinomial
        return top, state['trial'] + 1, state['draws']

    cases = []
    from cv2amd import philox
    seed_ctr = [1000]

    def table(seed, step):
        """The uniforms the device's sampler draws for (slot 0, step, trial 0 .. 101) under `seed` (cv2amd/philox.py = k_sample's Philox):
        the committed table IS what the GPU test's k_sample will use."""
        return torch.tensor([philox.uniforms(0, step, t, seed) for t in range(102)], dtype=torch.float64)

    def add(kind, logp, window, ignore_eos, want=None):
        """want(u) -> bool: take the first seed whose table has the property the case was built for"""
        while True:
            seed_ctr[0] += 1
            seed, step = seed_ctr[0] * 7919 + 13, 1 + seed_ctr[0] % 37
            u = table(seed, step)
            if want is None or want(u):
                break
        cases.append((kind, logp.float().log_softmax(0), list(window), bool(ignore_eos), u, seed, step))

    def peaked(ids, vals, base=-14.0, noise=0.3):
        x = torch.full((6564,), base) + noise * torch.randint(-8, 9, (6564,), generator=g).float() / 4      # (few distinct values: the fixture compresses)
        for i, v in zip(ids, vals):
            x[i] = v
        return x

    # (a) broad logits, window without the likely candidates: the nucleus draw stands (kind 0)
    for i in range(12):
        add(0, torch.randn(6564, generator=g) * (0.8 + 0.4 * i), torch.randint(0, 6561, (int(torch.randint(0, 14, (1,), generator=g)),), generator=g).tolist(), i % 2)
    # (b) a few dominant ids that fill the window: rep_num >= 1 -> the full-vocabulary re-draw decides (kind 1); windows shorter and longer than 10
    for i in range(16):
        ids = torch.randint(0, 6561, (3,), generator=g).tolist()
        x = peaked(ids, [2.0, 1.5, 1.0])
        far = torch.randint(0, 6561, (6,), generator=g).tolist()
        window = ([ids[0]] * 12 + far if i % 4 == 3 else far + [ids[i % 3]] + ids[:(i % 4)])     # (i % 4 == 3: the hit lies OUTSIDE the last 10 -> no re-draw if the nucleus picks it)
        add(1, x, window, i % 2)
    # (c) EOS among the candidates while ignore_eos: re-draws until another id comes (kind 2); the same logits with ignore_eos False keep EOS
    for i in range(16):
        ids = torch.randint(0, 6561, (2,), generator=g).tolist()
        x = peaked([EOS] + ids, [2.2 + 0.05 * i, 2.0, 1.0])
        add(2, x, [ids[0]] if i % 3 == 0 else [], i % 4 != 3)
    # (d) EOS (almost) certain while ignore_eos: 101 trials, then the RuntimeError (kind 3); one case escapes on a late trial through the full-vocabulary draw
    for i in range(8):
        x = peaked([EOS], [40.0], base=-40.0, noise=0.0)
        window = [EOS] if i >= 4 else []           # EOS in the window: every trial also takes the full-vocabulary draw (EOS again)
        add(3, x, window, True)
    for i in range(4):
        x = peaked([EOS, 17 + i], [6.0, 0.0], base=-30.0, noise=0.0)        # EOS holds 0.9975 of the mass (> top_p: the only nucleus candidate) and sits in the window:
        # every trial goes to the full-vocabulary draw, which returns EOS for u >= 0.0025 and id 17 + i below: a seed whose first such u comes on a late trial
        first = lambda u: next((t for t in range(101) if float(u[t, 1]) < 0.0024), -1)
        add(3, x, [EOS], True, want=lambda u: 50 < first(u) <= 100 and all(not (0.0024 <= float(u[t, 1]) < 0.0026) for t in range(101)))
    # (e) ties: equal top logits (the stable sort keeps the lower id first), exactly 25 candidates, one candidate holding > top_p (kind 4)
    for i in range(8):
        x = torch.full((6564,), -20.0)
        if i < 3:
            x[[100, 50, 3000, 7][: i + 2]] = 1.0
        elif i < 6:
            x[torch.randperm(6561, generator=g)[:40]] = 0.0
        else:
            x[1234] = 5.0
        add(4, x, [50] if i == 1 else [], i % 2)
    out = dict(logp=[], window=[], window_len=[], ignore_eos=[], uniforms=[], kind=[], top=[], trials=[], draws=[], seed=[], step=[])
    hist = {}
    for kind, logp, window, ign, u, seed, step in cases:
        top, trials, draws = run(logp, window, ign, u)
        uni = lambda s_, t_: (float(u[t_, 0]), float(u[t_, 1]))
        try:
            o = OL.sampling_ids(logp, list(window), ign, 'ras', uni, 0)
        except RuntimeError:
            o = -1
        assert o == top, f'oracle != reference (sampler decision, kind {kind}): {o} vs {top}'
        w = np.full(24, -1, dtype=np.int64)
        w[:len(window)] = window
        out['logp'].append(logp.numpy()); out['window'].append(w); out['window_len'].append(len(window)); out['ignore_eos'].append(int(ign))
        out['uniforms'].append(u.numpy()); out['kind'].append(kind); out['top'].append(top); out['trials'].append(trials); out['draws'].append(draws)
        out['seed'].append(seed); out['step'].append(step)
        hist.setdefault(kind, []).append((top, trials, draws))
    # the cases cover what they were built for
    assert all(t == 1 and d == 1 for _, t, d in hist[0])                          # no repetition: one trial, one draw
    assert sum(1 for _, t, d in hist[1] if d == 2 * t) >= 8                       # the repetition rule fired
    assert any(t > 1 for _, t, d in hist[2]) and any(top == EOS for top, _, _ in hist[2])
    assert sum(1 for top, t, _ in hist[3] if top == -1 and t == 101) >= 8 and any(top >= 0 and t > 50 for top, t, _ in hist[3])
    save('sampler_ras.npz', **{k: np.stack(v) if k in ('logp', 'window', 'uniforms') else np.asarray(v) for k, v in out.items()})
    print('sampler_ras: kinds', {k: len(v) for k, v in hist.items()}, 'errors', sum(1 for v in out['top'] if v == -1))


if __name__ == '__main__':
    assert R.available(), 'needs /root/reference'
    R.activate()
    which = sys.argv[1:] or ['hift', 'flow', 'llm', 'llm_bf16w', 'bistream', 'fullsize', 'text', 'textnorm', 'sampler', 'sampler_ras']
    for w in which:
        globals()['gen_' + w]()
