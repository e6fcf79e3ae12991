"""CPU fp32 restatement of stage 2 (token -> mel: conformer encoder + CFM Euler solver).  TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module.

Plain functions over a state dict in the reference's `flow.pt` key schema:
  * CausalMaskedDiffWithXvec.inference          cosyvoice/flow/flow.py:235-283
  * UpsampleConformerEncoder.forward            cosyvoice/transformer/upsample_encoder.py:243-306
    (LinearNoSubsampling subsampling.py:69-113, EspnetRelPositionalEncoding embedding.py:201-302,
     PreLookaheadLayer upsample_encoder.py:66-102, Upsample1D :37-63, ConformerEncoderLayer encoder_layer.py:160-236,
     RelPositionMultiHeadedAttention attention.py:200-330, PositionwiseFeedForward positionwise_feed_forward.py:20-57)
  * CausalConditionalCFM.forward / solve_euler  cosyvoice/flow/flow_matching.py:200-225, 71-123
  * CausalConditionalDecoder.forward            cosyvoice/flow/decoder.py:405-494 (+ :36-85 causal conv/block/resnet,
    matcha decoder.py:14-29,46-61,73-117; matcha transformer.py:83-134,243-316 over diffusers==0.29.0
    Attention/GELU, which are NOT under /root/reference: their published semantics are restated here —
    q/k/v Linear without bias, out Linear with bias, SDPA scale 1/sqrt(64), additive mask; exact-erf GELU)
  * masks: cosyvoice/utils/mask.py subsequent_chunk_mask / add_optional_chunk_mask; utils/common.py:160-168
Pinned against the real reference modules (with the diffusers shim of oracle/ref_harness.py) by
tests/golden/make_golden.py -> tests/golden/flow_*.npz.  The diffusers part itself is "parity unpinned"
by any reference-held vector (the reference has no tests).
"""
import math

import torch
import torch.nn.functional as F

_NOISE = {}

# Rounded-operand mode: every matrix-product operand (activations AND weights of Linear / Conv1d, q / k / v and the softmax
# probabilities of attention) is rounded to bf16 before the fp32 product, which is where the HIP path rounds (bf16 MFMA operands,
# fp32 accumulation, fp32 LayerNorm / softmax / residual stream).  With it the HIP-vs-oracle comparison separates operand
# rounding (modelled) from everything else (accumulation order, exp / erf approximations), so the GPU tests can hold a ~10x
# tighter bar than against the fp32 oracle.  Off by default: the golden vectors of the reference are fp32.
ROUND = {'on': False}


class rounded_operands:
    def __enter__(self):
        self.prev, ROUND['on'] = ROUND['on'], True

    def __exit__(self, *a):
        ROUND['on'] = self.prev


def _r(x):
    return x.to(torch.bfloat16).to(torch.float32) if ROUND['on'] else x


def _linear(x, w, b=None):
    return F.linear(_r(x), _r(w), b)


def _conv1d(x, w, b=None):
    return F.conv1d(_r(x), _r(w), b)


def _matmul(a, b):
    return torch.matmul(_r(a), _r(b))


def rand_noise():
    """flow_matching.py:197-198 — set_all_random_seed(0); randn([1, 80, 15000]) on the CPU generator."""
    if 'z' not in _NOISE:
        g = torch.Generator(device='cpu')
        g.manual_seed(0)
        _NOISE['z'] = torch.randn([1, 80, 50 * 300], generator=g)
    return _NOISE['z']


def lin(sd, name, x):
    return _linear(x, sd[name + '.weight'], sd.get(name + '.bias'))


def ln(sd, name, x, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[name + '.weight'], sd[name + '.bias'], eps)


# ------------------------------ encoder -----------------------------------
def rel_pos_emb(T, d_model=512):
    """embedding.py:226-302: positions T-1 .. -(T-1), interleaved sin/cos."""
    pos = torch.arange(T - 1, -T, -1, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * -(math.log(10000.0) / d_model))
    pe = torch.zeros(2 * T - 1, d_model)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe.unsqueeze(0)


def chunk_mask(T, chunk):
    """mask.py subsequent_chunk_mask with num_left_chunks=-1: key j visible to query i iff j < (i//chunk+1)*chunk."""
    i = torch.arange(T)
    return i[None, :] < ((i // chunk + 1) * chunk)[:, None]


def embed(sd, p, x):
    # LinearNoSubsampling.out = Linear, LayerNorm(eps 1e-5), Dropout ; then x * sqrt(d_model)
    x = ln(sd, p + '.out.1', lin(sd, p + '.out.0', x), 1e-5)
    return x * math.sqrt(x.shape[-1])


def rel_shift(x):
    # attention.py:225-247
    b, h, t1, n = x.shape
    zp = torch.zeros((b, h, t1, 1), dtype=x.dtype)
    xp = torch.cat([zp, x], dim=-1).view(b, h, n + 1, t1)
    return xp[:, :, 1:].view_as(x)[:, :, :, : n // 2 + 1]


def rel_attention(sd, p, x, mask, pos_emb, heads=8):
    B, T, D = x.shape
    dk = D // heads
    q = lin(sd, p + '.linear_q', x).view(B, T, heads, dk)
    k = lin(sd, p + '.linear_k', x).view(B, T, heads, dk).transpose(1, 2)
    v = lin(sd, p + '.linear_v', x).view(B, T, heads, dk).transpose(1, 2)
    pp = _linear(pos_emb, sd[p + '.linear_pos.weight']).view(1, -1, heads, dk).transpose(1, 2)
    qu = (q + sd[p + '.pos_bias_u']).transpose(1, 2)
    qv = (q + sd[p + '.pos_bias_v']).transpose(1, 2)
    ac = _matmul(qu, k.transpose(-2, -1))
    bd = rel_shift(_matmul(qv, pp.transpose(-2, -1)))
    s = (ac + bd) / math.sqrt(dk)
    m = mask.unsqueeze(1).eq(0)
    s = s.masked_fill(m, float('-inf'))
    a = torch.softmax(s, dim=-1).masked_fill(m, 0.0)
    o = _matmul(a, v).transpose(1, 2).contiguous().view(B, T, D)
    return lin(sd, p + '.linear_out', o)


def conformer_layer(sd, p, x, mask, pos_emb):
    # encoder_layer.py:160-236 with macaron_style=False, use_cnn_module=False, normalize_before=True
    x = x + rel_attention(sd, p + '.self_attn', ln(sd, p + '.norm_mha', x, 1e-12), mask, pos_emb)
    h = ln(sd, p + '.norm_ff', x, 1e-12)
    return x + lin(sd, p + '.feed_forward.w_2', F.silu(lin(sd, p + '.feed_forward.w_1', h)))


def pre_lookahead(sd, x, context):
    # upsample_encoder.py:82-102
    o = x.transpose(1, 2)
    if context is None:
        o = F.pad(o, (0, 3))
    else:
        assert context.shape[1] == 3
        o = torch.cat([o, context.transpose(1, 2)], dim=2)
    o = F.leaky_relu(_conv1d(o, sd['encoder.pre_lookahead_layer.conv1.weight'], sd['encoder.pre_lookahead_layer.conv1.bias']))
    o = F.pad(o, (2, 0))
    o = _conv1d(o, sd['encoder.pre_lookahead_layer.conv2.weight'], sd['encoder.pre_lookahead_layer.conv2.bias'])
    return o.transpose(1, 2) + x


def encoder(sd, xs, context=None, streaming=False, n_enc=None, n_up=None):
    """xs [1, T, 512] (already embedded tokens); context [1, 3, 512] or None.  Returns [1, 2T, 512]."""
    B, T, _ = xs.shape
    masks = torch.ones(B, 1, T, dtype=torch.bool)
    xs = embed(sd, 'encoder.embed', xs)
    pos = rel_pos_emb(T)
    if context is not None:
        context = embed(sd, 'encoder.embed', context)
    cm = (masks & chunk_mask(T, 25).unsqueeze(0)) if streaming else masks
    xs = pre_lookahead(sd, xs, context)
    i = 0
    while f'encoder.encoders.{i}.norm_ff.weight' in sd and (n_enc is None or i < n_enc):
        xs = conformer_layer(sd, f'encoder.encoders.{i}', xs, cm, pos)
        i += 1
    # Upsample1D: nearest x2, left pad 4, conv k=5
    o = F.interpolate(xs.transpose(1, 2), scale_factor=2.0, mode='nearest')
    o = _conv1d(F.pad(o, (4, 0)), sd['encoder.up_layer.conv.weight'], sd['encoder.up_layer.conv.bias'])
    xs = o.transpose(1, 2)
    T2 = xs.shape[1]
    masks = torch.ones(B, 1, T2, dtype=torch.bool)
    xs = embed(sd, 'encoder.up_embed', xs)
    pos = rel_pos_emb(T2)
    cm = (masks & chunk_mask(T2, 50).unsqueeze(0)) if streaming else masks
    i = 0
    while f'encoder.up_encoders.{i}.norm_ff.weight' in sd and (n_up is None or i < n_up):
        xs = conformer_layer(sd, f'encoder.up_encoders.{i}', xs, cm, pos)
        i += 1
    return ln(sd, 'encoder.after_norm', xs, 1e-5)


# ------------------------------ estimator ---------------------------------
def sinusoidal_pos_emb(t, dim=320, scale=1000):
    # matcha decoder.py:14-29
    half = dim // 2
    e = math.log(10000) / (half - 1)
    e = torch.exp(torch.arange(half).float() * -e)
    e = scale * t.unsqueeze(1) * e.unsqueeze(0)
    return torch.cat((e.sin(), e.cos()), dim=-1)


def causal_conv(sd, name, x):
    k = sd[name + '.weight'].shape[-1]
    return _conv1d(F.pad(x, (k - 1, 0)), sd[name + '.weight'], sd[name + '.bias'])


def causal_block(sd, p, x, mask):
    # decoder.py:65-78: conv(x*mask) -> LN over C -> Mish -> *mask
    h = causal_conv(sd, p + '.block.0', x * mask)
    h = ln(sd, p + '.block.2', h.transpose(1, 2), 1e-5).transpose(1, 2)
    return F.mish(h) * mask


def resnet(sd, p, x, mask, temb):
    # matcha decoder.py:56-61
    h = causal_block(sd, p + '.block1', x, mask)
    h = h + lin(sd, p + '.mlp.1', F.mish(temb)).unsqueeze(-1)
    h = causal_block(sd, p + '.block2', h, mask)
    return h + _conv1d(x * mask, sd[p + '.res_conv.weight'], sd[p + '.res_conv.bias'])


def transformer_block(sd, p, x, bias, heads=8):
    # matcha transformer.py:243-316 (self-attn + FF only; dropout 0)
    B, T, C = x.shape
    h = ln(sd, p + '.norm1', x, 1e-5)
    q = _linear(h, sd[p + '.attn1.to_q.weight']).view(B, T, heads, -1).transpose(1, 2)
    k = _linear(h, sd[p + '.attn1.to_k.weight']).view(B, T, heads, -1).transpose(1, 2)
    v = _linear(h, sd[p + '.attn1.to_v.weight']).view(B, T, heads, -1).transpose(1, 2)
    s = _matmul(q, k.transpose(-2, -1)) / math.sqrt(q.shape[-1]) + bias.unsqueeze(1)
    o = _matmul(torch.softmax(s, dim=-1), v).transpose(1, 2).reshape(B, T, -1)
    x = x + lin(sd, p + '.attn1.to_out.0', o)
    h = ln(sd, p + '.norm3', x, 1e-5)
    return x + lin(sd, p + '.ff.net.2', F.gelu(lin(sd, p + '.ff.net.0.proj', h)))


def attn_bias(mask, streaming, chunk=50):
    # decoder.py:436-442 + utils/common.py:160-168 ; mask [B,1,T] float
    B, _, T = mask.shape
    m = mask.bool()
    m = (m & chunk_mask(T, chunk).unsqueeze(0)) if streaming else m.repeat(1, T, 1)
    return (1.0 - m.float()) * -1.0e10


def _blocks(sd, prefix):
    i = 0
    while f'{prefix}.{i}.0.mlp.1.weight' in sd:
        i += 1
    return i


def estimator(sd, x, mask, mu, t, spks, cond, streaming=False):
    """CausalConditionalDecoder.forward.  x,mu,cond [B,80,T]; mask [B,1,T]; t [B]; spks [B,80] -> [B,80,T]."""
    P = 'decoder.estimator'
    temb = sinusoidal_pos_emb(t)
    temb = lin(sd, P + '.time_mlp.linear_2', F.silu(lin(sd, P + '.time_mlp.linear_1', temb)))
    T = x.shape[-1]
    h = torch.cat([x, mu, spks.unsqueeze(-1).expand(-1, -1, T), cond], dim=1)
    bias = attn_bias(mask, streaming)

    def tblocks(p, h):
        h = h.transpose(1, 2)
        j = 0
        while f'{p}.1.{j}.norm1.weight' in sd:
            h = transformer_block(sd, f'{p}.1.{j}', h, bias)
            j += 1
        return h.transpose(1, 2)

    h = resnet(sd, P + '.down_blocks.0.0', h, mask, temb)
    h = tblocks(P + '.down_blocks.0', h)
    skip = h
    h = causal_conv(sd, P + '.down_blocks.0.2', h * mask)
    for i in range(_blocks(sd, P + '.mid_blocks')):
        h = resnet(sd, f'{P}.mid_blocks.{i}.0', h, mask, temb)
        h = tblocks(f'{P}.mid_blocks.{i}', h)
    h = torch.cat([h[:, :, :skip.shape[-1]], skip], dim=1)
    h = resnet(sd, P + '.up_blocks.0.0', h, mask, temb)
    h = tblocks(P + '.up_blocks.0', h)
    h = causal_conv(sd, P + '.up_blocks.0.2', h * mask)
    h = causal_block(sd, P + '.final_block', h, mask)
    o = _conv1d(h * mask, sd[P + '.final_proj.weight'], sd[P + '.final_proj.bias'])
    return o * mask


def t_span(n=10):
    t = torch.linspace(0, 1, n + 1)
    return 1 - torch.cos(t * 0.5 * torch.pi)


def solve_euler(sd, z, mu, mask, spks, cond, n_timesteps=10, streaming=False, cfg=0.7, return_steps=False):
    """flow_matching.py:71-123 (B=1 utterance, CFG batch of 2)."""
    ts = t_span(n_timesteps)
    x = z
    t, dt = ts[0:1], ts[1] - ts[0]
    T = x.shape[2]
    steps = []
    for step in range(1, len(ts)):
        x_in = x.expand(2, -1, -1).contiguous()
        mask_in = mask.expand(2, -1, -1).contiguous()
        mu_in = torch.zeros(2, 80, T)
        mu_in[0] = mu[0]
        t_in = t.expand(2).contiguous()
        spks_in = torch.zeros(2, 80)
        spks_in[0] = spks[0]
        cond_in = torch.zeros(2, 80, T)
        cond_in[0] = cond[0]
        d = estimator(sd, x_in, mask_in, mu_in, t_in, spks_in, cond_in, streaming)
        d = (1.0 + cfg) * d[0:1] - cfg * d[1:2]
        x = x + dt * d
        t = t + dt
        if return_steps:
            steps.append(x.clone())
        if step < len(ts) - 1:
            dt = ts[step + 1] - t
    return (x, steps) if return_steps else x


def inference(sd, token, prompt_token, prompt_feat, embedding, streaming=False, finalize=True, n_timesteps=10):
    """CausalMaskedDiffWithXvec.inference.  token [1,n] int, prompt_token [1,P], prompt_feat [1,2P,80], embedding [1,192]
    -> mel [1, 80, 2n(-6 if not finalize)]."""
    spk = lin(sd, 'spk_embed_affine_layer', F.normalize(embedding, dim=1))
    tok = torch.cat([prompt_token, token], dim=1).clamp(min=0).long()
    x = sd['input_embedding.weight'][tok]
    if finalize:
        h = encoder(sd, x, None, streaming)
    else:
        h = encoder(sd, x[:, :-3], x[:, -3:], streaming)
    mel_len1 = prompt_feat.shape[1]
    mel_len2 = h.shape[1] - mel_len1
    h = lin(sd, 'encoder_proj', h)
    conds = torch.zeros(1, mel_len1 + mel_len2, 80)
    conds[:, :mel_len1] = prompt_feat
    conds = conds.transpose(1, 2)
    mask = torch.ones(1, 1, mel_len1 + mel_len2)
    mu = h.transpose(1, 2).contiguous()
    z = rand_noise()[:, :, :mu.shape[2]]
    feat = solve_euler(sd, z, mu, mask, spk, conds, n_timesteps, streaming)
    return feat[:, :, mel_len1:].float()
