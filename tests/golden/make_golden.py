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
  nucleus_sampling candidate set (common.py:120-134)         sampler.npz
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


def gen_llm():
    l = R.build_llm(num_layers=24)
    sd = synth.make_llm(layers=24)
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
    save('llm_greedy.npz', text_len=6, prompt_len=12, prompt_text_len=4, **out)


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


if __name__ == '__main__':
    assert R.available(), 'needs /root/reference'
    R.activate()
    which = sys.argv[1:] or ['hift', 'flow', 'llm', 'sampler']
    for w in which:
        globals()['gen_' + w]()
