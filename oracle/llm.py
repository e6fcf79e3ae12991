"""CPU fp32 restatement of stage 1 (Qwen2-0.5B speech-token LM).  TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module; the product path (`cosyvoice2-eu_amd/`) never does.

Restates, as plain functions over a state dict in the reference's `llm.pt` key schema:
  * HF `Qwen2ForCausalLM` decoder stack as called from `HFBackbone.forward_one_step`
    (cosyvoice/llm/llm.py:107-117; third-party transformers==4.40.1 `modeling_qwen2.py`:
    RMSNorm eps 1e-6, q/k/v bias, rotate-half RoPE theta 1e6, GQA 14:2, SwiGLU, post-norm hidden)
  * `Qwen2LM.inference` input assembly (llm/llm.py:625-647)
  * `Qwen2LM.inference_wrapper` step loop (llm/llm.py:681-719)
  * `TransformerLM.sampling_ids` (llm/llm.py:235-250) and `ras_sampling`/`nucleus_sampling`/
    `random_sampling` (cosyvoice/utils/common.py:111-139)
Pinned against the real reference modules by tests/golden/make_golden.py (golden fixtures in
tests/golden/llm_*.npz) and tests/test_oracle_vs_reference.py.
"""
import math

import torch
import torch.nn.functional as F

SPEECH_TOKEN_SIZE = 6561          # EOS id; fill token = +2 (llm/llm.py:390-398)


class LLMDims:
    def __init__(self, sd):
        self.hidden = sd['llm.model.model.norm.weight'].numel()
        n = 0
        while f'llm.model.model.layers.{n}.input_layernorm.weight' in sd:
            n += 1
        self.layers = n
        self.head_dim = 64
        self.n_q = sd['llm.model.model.layers.0.self_attn.q_proj.weight'].shape[0] // 64
        self.n_kv = sd['llm.model.model.layers.0.self_attn.k_proj.weight'].shape[0] // 64
        self.inter = sd['llm.model.model.layers.0.mlp.gate_proj.weight'].shape[0]
        self.rope_theta = 1e6
        self.eps = 1e-6


def rmsnorm(x, w, eps):
    # modeling_qwen2.Qwen2RMSNorm.forward
    v = x.pow(2).mean(-1, keepdim=True)
    return w * (x * torch.rsqrt(v + eps))


def rope_cos_sin(pos, theta=1e6, dim=64):
    # Qwen2RotaryEmbedding: inv_freq = 1/theta^(2i/dim); emb = cat(freqs, freqs)
    inv = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float32) / dim))
    fr = pos.to(torch.float32)[:, None] * inv[None, :]
    emb = torch.cat([fr, fr], dim=-1)
    return emb.cos(), emb.sin()


def rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat([-x[..., h:], x[..., :h]], dim=-1)


def qwen2_step(sd, d, x, cache):
    """x: [L, hidden] new positions; cache: list of (k [n_kv, P, 64], v) or None. Returns post-norm hidden [L, hidden]."""
    L = x.shape[0]
    past = 0 if cache[0] is None else cache[0][0].shape[1]
    pos = torch.arange(past, past + L)
    cos, sin = rope_cos_sin(pos, d.rope_theta, d.head_dim)
    for i in range(d.layers):
        p = f'llm.model.model.layers.{i}.'
        h = rmsnorm(x, sd[p + 'input_layernorm.weight'], d.eps)
        q = F.linear(h, sd[p + 'self_attn.q_proj.weight'], sd[p + 'self_attn.q_proj.bias']).view(L, d.n_q, 64).transpose(0, 1)
        k = F.linear(h, sd[p + 'self_attn.k_proj.weight'], sd[p + 'self_attn.k_proj.bias']).view(L, d.n_kv, 64).transpose(0, 1)
        v = F.linear(h, sd[p + 'self_attn.v_proj.weight'], sd[p + 'self_attn.v_proj.bias']).view(L, d.n_kv, 64).transpose(0, 1)
        q = q * cos + rotate_half(q) * sin
        k = k * cos + rotate_half(k) * sin
        if cache[i] is not None:
            k = torch.cat([cache[i][0], k], dim=1)
            v = torch.cat([cache[i][1], v], dim=1)
        cache[i] = (k, v)
        rep = d.n_q // d.n_kv
        kk = k.repeat_interleave(rep, dim=0)
        vv = v.repeat_interleave(rep, dim=0)
        s = torch.matmul(q, kk.transpose(1, 2)) / math.sqrt(64)          # [n_q, L, P+L]
        T = past + L
        causal = torch.arange(T)[None, :] <= (past + torch.arange(L))[:, None]
        s = s.masked_fill(~causal[None], float('-inf'))
        a = torch.softmax(s, dim=-1, dtype=torch.float32)
        o = torch.matmul(a, vv).transpose(0, 1).reshape(L, d.n_q * 64)
        x = x + F.linear(o, sd[p + 'self_attn.o_proj.weight'])
        h = rmsnorm(x, sd[p + 'post_attention_layernorm.weight'], d.eps)
        g = F.linear(h, sd[p + 'mlp.gate_proj.weight'])
        u = F.linear(h, sd[p + 'mlp.up_proj.weight'])
        x = x + F.linear(F.silu(g) * u, sd[p + 'mlp.down_proj.weight'])
    return rmsnorm(x, sd['llm.model.model.norm.weight'], d.eps)


def build_lm_input(sd, text, prompt_text, prompt_speech_token):
    """llm/llm.py:625-641 — [sos, embed_tokens(prompt_text ++ text), task_id, speech_embedding(prompt_speech)]."""
    ids = torch.cat([prompt_text.reshape(-1), text.reshape(-1)]).long()
    temb = sd['llm.model.model.embed_tokens.weight'][ids]
    sos = sd['llm_embedding.weight'][0:1]
    task = sd['llm_embedding.weight'][1:2]
    pemb = sd['speech_embedding.weight'][prompt_speech_token.reshape(-1).long()]
    return torch.cat([sos, temb, task, pemb], dim=0)


def min_max_len(text_len, min_ratio=2, max_ratio=20):
    # llm/llm.py:643-644 (text_len = target text tokens only)
    return int(text_len * min_ratio), int(text_len * max_ratio)


# ----------------------------- samplers -----------------------------------
def greedy_ids(logp, ignore_eos):
    """Harness-defined greedy: the fixed point of sampling_ids (llm/llm.py:235-250) under a deterministic
    sampler — argmax with EOS excluded while ignore_eos (instead of 100 identical re-draws + RuntimeError)."""
    s = logp.clone()
    if ignore_eos:
        s[SPEECH_TOKEN_SIZE] = float('-inf')
    return int(s.argmax())


def multinomial_inv_cdf(p, u):
    """Draw one index from (unnormalised) p with the injected uniform u in [0,1): first i with cdf_i > u*sum."""
    c = torch.cumsum(p.double(), 0)
    i = int(torch.searchsorted(c, torch.tensor(u, dtype=torch.float64) * c[-1], right=True))
    return min(i, p.numel() - 1)


def nucleus_candidates(logp, top_p=0.8, top_k=25):
    """utils/common.py:120-134: stable descending sort of softmax; keep while cum < top_p and count < top_k."""
    sv, si = logp.softmax(dim=0).sort(descending=True, stable=True)
    prob, idx = [], []
    cum = torch.tensor(0.0)          # the reference's `cum_prob += sorted_value[i]` accumulates a 0-dim fp32 tensor
    for i in range(len(si)):
        if bool(cum < top_p) and len(prob) < top_k:
            cum = cum + sv[i]
            prob.append(float(sv[i]))
            idx.append(int(si[i]))
        else:
            break
    return torch.tensor(prob), idx


def ras_ids(logp, decoded, u_pair, top_p=0.8, top_k=25, win_size=10, tau_r=0.1):
    """utils/common.py:111-117 with injected uniforms (u_nucleus, u_random) replacing torch.multinomial's RNG."""
    prob, idx = nucleus_candidates(logp, top_p, top_k)
    top = idx[multinomial_inv_cdf(prob, u_pair[0])]
    rep = sum(1 for t in decoded[-win_size:] if t == top)
    if rep >= win_size * tau_r:
        top = multinomial_inv_cdf(logp.softmax(dim=0), u_pair[1])
    return top


def sampling_ids(logp, decoded, ignore_eos, mode, uniforms=None, step=0, max_trials=100):
    """llm/llm.py:235-250.  mode 'greedy' or 'ras'; uniforms: callable (step, trial) -> (u_nucleus, u_random),
    the injected noise that replaces torch.multinomial's RNG in 'ras' mode."""
    if mode == 'greedy':
        return greedy_ids(logp, ignore_eos)
    trials = 0
    while True:
        top = ras_ids(logp, decoded, uniforms(step, trials))
        if (not ignore_eos) or top != SPEECH_TOKEN_SIZE:
            return top
        trials += 1
        if trials > max_trials:
            raise RuntimeError('sampling reaches max_trials {} and still get eos when ignore_eos is True, '
                               'check your input!'.format(max_trials))


def inference(sd, text, prompt_text, prompt_speech_token, mode='greedy', uniforms=None,
              max_ratio=20, min_ratio=2, force_len=None, return_logp=False):
    """Qwen2LM.inference + inference_wrapper (llm/llm.py:575-719), unistream.  Returns emitted ids (list).

    force_len: synthetic-weights mode — ignore EOS entirely and stop after exactly force_len emitted tokens
    (SURVEY.md §8(d)); EOS/fill ids are masked so every step emits.
    """
    d = LLMDims(sd)
    lm_input = build_lm_input(sd, text, prompt_text, prompt_speech_token)
    min_len, max_len = min_max_len(text.numel(), min_ratio, max_ratio)
    if force_len is not None:
        min_len, max_len = force_len, force_len
    cache = [None] * d.layers
    out, logps = [], []
    x = lm_input
    for i in range(max_len):
        y = qwen2_step(sd, d, x, cache)
        logp = F.linear(y[-1], sd['llm_decoder.weight'], sd['llm_decoder.bias']).log_softmax(dim=-1)
        if i == 0:
            logp[SPEECH_TOKEN_SIZE] = float('-inf')
        if force_len is not None:
            logp[SPEECH_TOKEN_SIZE:] = float('-inf')
        if return_logp:
            logps.append(logp.clone())
        top = sampling_ids(logp, out, i < min_len, mode, uniforms, i)
        if top == SPEECH_TOKEN_SIZE:
            break
        x = sd['speech_embedding.weight'][top:top + 1]
        if top > SPEECH_TOKEN_SIZE:
            continue
        out.append(top)
    return (out, logps) if return_logp else out


FILL_TOKEN = SPEECH_TOKEN_SIZE + 2     # "need more text" (llm/llm.py:412)


def inference_bistream(sd, text_chunks, prompt_text, prompt_speech_token, mode='greedy', uniforms=None, mix_ratio=(5, 15),
                       max_steps=100000):
    """Qwen2LM.inference_bistream (llm/llm.py:721-834) restated: text arrives in pieces; the LM input interleaves mix_ratio[0]
    text tokens with mix_ratio[1] speech tokens; id 6563 (fill) asks for the next text block and is forced every
    mix_ratio[1] + 1 emitted entries once it has appeared.  Yields nothing: returns (emitted ids, out_tokens incl. fill / EOS).

    Kept exactly, including what looks accidental in the reference: `out_tokens` (the sampler's repetition window) contains the
    fill tokens; after a fill the stale `lm_input` (the last fed embedding) is REPLACED by the next text block, but at the end of
    the text it is fed AGAIN in front of the remaining text and the task id (llm.py:817); EOS is only re-drawn (never masked)
    while text is still expected; an id >= 6561 other than fill (mid) / EOS (final) raises ValueError (llm.py:809, 829).
    `max_steps` bounds the loop for synthetic weights (the reference has no bound here)."""
    d = LLMDims(sd)
    emb = sd['llm.model.model.embed_tokens.weight']
    sp = sd['speech_embedding.weight']
    n_text, n_speech = mix_ratio
    prompt_speech = sp[prompt_speech_token.reshape(-1).long()]          # remaining prompt speech embeddings [P, H]
    lm_input = sd['llm_embedding.weight'][0:1]                          # sos
    task = sd['llm_embedding.weight'][1:2]
    text_cache = emb[prompt_text.reshape(-1).long()]
    cache = [None] * d.layers
    out_tokens, emitted = [], []
    next_fill = -1
    step = 0

    def draw(logp, ignore_eos):
        nonlocal step
        top = sampling_ids(logp, out_tokens, ignore_eos, mode, uniforms, step)
        step += 1
        return top

    for this_text in text_chunks:
        text_cache = torch.cat([text_cache, emb[this_text.reshape(-1).long()]], dim=0)
        while prompt_speech.shape[0] != 0:                              # llm.py:766-774
            if text_cache.shape[0] >= n_text:
                lm_input = torch.cat([lm_input, text_cache[:n_text], prompt_speech[:n_speech]], dim=0)
                text_cache, prompt_speech = text_cache[n_text:], prompt_speech[n_speech:]
            else:
                break
        if prompt_speech.shape[0] == 0:                                 # llm.py:776-811
            last_fill = len(out_tokens) != 0 and out_tokens[-1] == FILL_TOKEN
            if last_fill or (len(out_tokens) == 0 and lm_input.shape[0] == 1):
                if text_cache.shape[0] >= n_text:
                    lm_input = text_cache[:n_text] if last_fill else torch.cat([lm_input, text_cache[:n_text]], dim=0)
                    text_cache = text_cache[n_text:]
                else:
                    continue
            while True:
                y = qwen2_step(sd, d, lm_input, cache)
                logp = F.linear(y[-1], sd['llm_decoder.weight'], sd['llm_decoder.bias']).log_softmax(dim=-1)
                if next_fill != -1 and len(out_tokens) == next_fill:
                    top = FILL_TOKEN
                    next_fill += n_speech + 1
                    step += 1                                           # the forced entry consumes a step index (no draw)
                else:
                    top = draw(logp, True)
                if top == FILL_TOKEN:
                    next_fill = len(out_tokens) + n_speech + 1
                out_tokens.append(top)
                if top >= SPEECH_TOKEN_SIZE:
                    if top == FILL_TOKEN:
                        break
                    raise ValueError('should not get token {}'.format(top))
                emitted.append(top)
                lm_input = sp[top:top + 1]
                if len(out_tokens) > max_steps:
                    raise RuntimeError('bistream: max_steps exceeded')
    lm_input = torch.cat([lm_input, text_cache, task], dim=0)           # llm.py:817
    while True:
        y = qwen2_step(sd, d, lm_input, cache)
        logp = F.linear(y[-1], sd['llm_decoder.weight'], sd['llm_decoder.bias']).log_softmax(dim=-1)
        top = draw(logp, False)
        out_tokens.append(top)
        if top >= SPEECH_TOKEN_SIZE:
            if top == SPEECH_TOKEN_SIZE:
                break
            raise ValueError('should not get token {}'.format(top))
        emitted.append(top)
        lm_input = sp[top:top + 1]
        if len(out_tokens) > max_steps:
            raise RuntimeError('bistream: max_steps exceeded')
    return emitted, out_tokens
