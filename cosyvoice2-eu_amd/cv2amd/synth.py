"""Synthetic checkpoints in the reference's key schema.

There are no model weights offline, so benchmarks and parity tests run on seeded random weights at the exact
shapes of `llm.pt` / `flow.pt` / `hift.pt` (cosyvoice/cli/model.py:67-90 loads these with load_state_dict).
The generator is deterministic on the torch CPU generator, so the build container, the GPU box and the oracle
all see identical tensors without shipping them.

Init rule: matrices / conv kernels ~ N(0, gain^2 / fan_in) (variance preserving), norm gains 1, Snake alpha 1,
biases small N(0, 0.02^2); weight-norm g = ||v|| so the effective kernel equals v.  A few tensors get a
different gain so the synthetic model exercises every branch (voiced/unvoiced F0, un-clipped magnitudes).
"""
import math

import torch

LLM_DIMS = dict(hidden=896, inter=4864, layers=24, n_q=14, n_kv=2, vocab=151936, speech_vocab=6564)


class _Gen:
    def __init__(self, seed, dtype=torch.float32, meta=False):
        self.g = torch.Generator(device='cpu')
        self.g.manual_seed(seed)
        self.sd = {}
        self.dtype = dtype
        self.meta = meta             # shapes only (cv2amd/checkpoint.py builds the key schema from the same code path)

    def normal(self, name, shape, std):
        if self.meta:
            self.sd[name] = torch.empty(shape, dtype=self.dtype, device='meta')
            return
        self.sd[name] = (torch.randn(shape, generator=self.g) * std).to(self.dtype)

    def mat(self, name, shape, gain=1.0, fan_in=None):
        fi = fan_in if fan_in is not None else int(torch.tensor(shape[1:]).prod())
        self.normal(name, shape, gain / math.sqrt(fi))

    def bias(self, name, n, std=0.02):
        self.normal(name, (n,), std)

    def ones(self, name, n):
        self.sd[name] = torch.ones(n, dtype=self.dtype)

    def lin(self, name, out, inp, bias=True, gain=1.0):
        self.mat(name + '.weight', (out, inp), gain)
        if bias:
            self.bias(name + '.bias', out)

    def norm(self, name, n):
        # gains near 1, small bias: keeps LayerNorm affine part non-trivial for parity tests
        self.sd[name + '.weight'] = (1.0 + 0.1 * torch.randn(n, generator=self.g)).to(self.dtype)
        self.bias(name + '.bias', n)

    def wn_conv(self, name, shape, gain=1.0, fan_in=None, bias=True, n_bias=None):
        fi = fan_in if fan_in is not None else int(torch.tensor(shape[1:]).prod())
        if self.meta:
            self.sd[name + '.parametrizations.weight.original1'] = torch.empty(shape, dtype=self.dtype, device='meta')
            self.sd[name + '.parametrizations.weight.original0'] = torch.empty((shape[0],) + (1,) * (len(shape) - 1), dtype=self.dtype, device='meta')
            if bias:
                self.bias(name + '.bias', n_bias if n_bias is not None else shape[0])
            return
        v = torch.randn(shape, generator=self.g) * (gain / math.sqrt(fi))
        self.sd[name + '.parametrizations.weight.original1'] = v.to(self.dtype)
        self.sd[name + '.parametrizations.weight.original0'] = v.flatten(1).norm(dim=1).view(-1, *([1] * (len(shape) - 1))).to(self.dtype)
        if bias:
            self.bias(name + '.bias', n_bias if n_bias is not None else shape[0])


def make_llm(seed=1986, layers=24, hidden=896, inter=4864, n_q=14, n_kv=2, vocab=151936, tie_lm_head=True, meta=False):
    """llm.pt schema: Qwen2LM over HFBackbone(Qwen2ForCausalLM) (cosyvoice/llm/llm.py:350-413)."""
    g = _Gen(seed, meta=meta)
    g.normal('llm.model.model.embed_tokens.weight', (vocab, hidden), 1.0)
    for i in range(layers):
        p = f'llm.model.model.layers.{i}.'
        g.lin(p + 'self_attn.q_proj', n_q * 64, hidden)
        g.lin(p + 'self_attn.k_proj', n_kv * 64, hidden)
        g.lin(p + 'self_attn.v_proj', n_kv * 64, hidden)
        g.lin(p + 'self_attn.o_proj', hidden, n_q * 64, bias=False)
        g.lin(p + 'mlp.gate_proj', inter, hidden, bias=False)
        g.lin(p + 'mlp.up_proj', inter, hidden, bias=False)
        g.lin(p + 'mlp.down_proj', hidden, inter, bias=False)
        g.sd[p + 'input_layernorm.weight'] = 1.0 + 0.1 * torch.randn(hidden, generator=g.g)
        g.sd[p + 'post_attention_layernorm.weight'] = 1.0 + 0.1 * torch.randn(hidden, generator=g.g)
    g.sd['llm.model.model.norm.weight'] = 1.0 + 0.1 * torch.randn(hidden, generator=g.g)
    if tie_lm_head:
        g.sd['llm.model.lm_head.weight'] = g.sd['llm.model.model.embed_tokens.weight']
    g.normal('llm_embedding.weight', (2, hidden), 1.0)
    g.normal('speech_embedding.weight', (6564, hidden), 1.0)
    # gain 3: logits with a usable top-1 margin on random weights
    g.lin('llm_decoder', 6564, hidden, gain=3.0)
    return g.sd


def _conformer_layer(g, p):
    g.normal(p + '.self_attn.pos_bias_u', (8, 64), 0.1)
    g.normal(p + '.self_attn.pos_bias_v', (8, 64), 0.1)
    for n in ('linear_q', 'linear_k', 'linear_v', 'linear_out'):
        g.lin(p + '.self_attn.' + n, 512, 512)
    g.lin(p + '.self_attn.linear_pos', 512, 512, bias=False)
    g.lin(p + '.feed_forward.w_1', 2048, 512)
    g.lin(p + '.feed_forward.w_2', 512, 2048)
    g.norm(p + '.norm_ff', 512)
    g.norm(p + '.norm_mha', 512)


def _tblock(g, p):
    g.norm(p + '.norm1', 256)
    for n in ('to_q', 'to_k', 'to_v'):
        g.lin(p + '.attn1.' + n, 512, 256, bias=False)
    g.lin(p + '.attn1.to_out.0', 256, 512)
    g.norm(p + '.norm3', 256)
    g.lin(p + '.ff.net.0.proj', 1024, 256)
    g.lin(p + '.ff.net.2', 256, 1024)


def _resnet(g, p, cin):
    g.lin(p + '.mlp.1', 256, 1024)
    g.mat(p + '.block1.block.0.weight', (256, cin, 3))
    g.bias(p + '.block1.block.0.bias', 256)
    g.norm(p + '.block1.block.2', 256)
    g.mat(p + '.block2.block.0.weight', (256, 256, 3))
    g.bias(p + '.block2.block.0.bias', 256)
    g.norm(p + '.block2.block.2', 256)
    g.mat(p + '.res_conv.weight', (256, cin, 1))
    g.bias(p + '.res_conv.bias', 256)


def make_flow(seed=1987, num_mid_blocks=12, n_blocks=4, enc_blocks=6, up_blocks=4, meta=False):
    """flow.pt schema: CausalMaskedDiffWithXvec (cosyvoice/flow/flow.py:150-196 and sub-modules)."""
    g = _Gen(seed, meta=meta)
    g.normal('input_embedding.weight', (6561, 512), 1.0)
    g.lin('spk_embed_affine_layer', 80, 192, gain=math.sqrt(192))      # unit-norm input -> O(1) output
    for e in ('embed', 'up_embed'):
        g.lin(f'encoder.{e}.out.0', 512, 512)
        g.norm(f'encoder.{e}.out.1', 512)
    g.mat('encoder.pre_lookahead_layer.conv1.weight', (512, 512, 4))
    g.bias('encoder.pre_lookahead_layer.conv1.bias', 512)
    g.mat('encoder.pre_lookahead_layer.conv2.weight', (512, 512, 3))
    g.bias('encoder.pre_lookahead_layer.conv2.bias', 512)
    for i in range(enc_blocks):
        _conformer_layer(g, f'encoder.encoders.{i}')
    g.mat('encoder.up_layer.conv.weight', (512, 512, 5))
    g.bias('encoder.up_layer.conv.bias', 512)
    for i in range(up_blocks):
        _conformer_layer(g, f'encoder.up_encoders.{i}')
    g.norm('encoder.after_norm', 512)
    g.lin('encoder_proj', 80, 512)
    P = 'decoder.estimator'
    g.lin(P + '.time_mlp.linear_1', 1024, 320)
    g.lin(P + '.time_mlp.linear_2', 1024, 1024)

    def block(p, cin, tail):
        _resnet(g, p + '.0', cin)
        for j in range(n_blocks):
            _tblock(g, f'{p}.1.{j}')
        if tail:
            g.mat(p + '.2.weight', (256, 256, 3))
            g.bias(p + '.2.bias', 256)
    block(P + '.down_blocks.0', 320, True)
    for i in range(num_mid_blocks):
        block(f'{P}.mid_blocks.{i}', 256, False)
    block(P + '.up_blocks.0', 512, True)
    g.mat(P + '.final_block.block.0.weight', (256, 256, 3))
    g.bias(P + '.final_block.block.0.bias', 256)
    g.norm(P + '.final_block.block.2', 256)
    g.mat(P + '.final_proj.weight', (80, 256, 1))
    g.bias(P + '.final_proj.bias', 80)
    return g.sd


def make_hift(seed=1988, meta=False):
    """hift.pt schema: HiFTGenerator (cosyvoice/hifigan/generator.py:392-503)."""
    g = _Gen(seed, meta=meta)
    g.lin('m_source.l_linear', 1, 9, gain=3.0)
    g.wn_conv('conv_pre', (512, 80, 7))
    chans = [512, 256, 128, 64]
    for i, k in enumerate((16, 11, 7)):
        # ConvTranspose weight is [C_in, C_out, k]; weight-norm g over dim 0; fan-in per output ~ C_in * k / stride
        g.wn_conv(f'ups.{i}', (chans[i], chans[i + 1], k), fan_in=chans[i] * k // (8, 5, 3)[i], n_bias=chans[i + 1])
    for i, (k, st) in enumerate(((30, 15), (6, 3), (1, 1))):
        g.mat(f'source_downs.{i}.weight', (chans[i + 1], 18, k), gain=0.5)
        g.bias(f'source_downs.{i}.bias', chans[i + 1])

    def rb(p, c, k):
        for j in range(3):
            g.wn_conv(f'{p}.convs1.{j}', (c, c, k), gain=0.7)
            g.wn_conv(f'{p}.convs2.{j}', (c, c, k), gain=0.5)
            g.sd[f'{p}.activations1.{j}.alpha'] = 1.0 + 0.2 * torch.randn(c, generator=g.g)
            g.sd[f'{p}.activations2.{j}.alpha'] = 1.0 + 0.2 * torch.randn(c, generator=g.g)
    for i in range(3):
        rb(f'source_resblocks.{i}', chans[i + 1], (7, 7, 11)[i])
        for j, k in enumerate((3, 7, 11)):
            rb(f'resblocks.{i * 3 + j}', chans[i + 1], k)
    g.wn_conv('conv_post', (18, 64, 7), gain=0.15)
    for n, i in enumerate((0, 2, 4, 6, 8)):
        g.wn_conv(f'f0_predictor.condnet.{i}', (512, 80 if n == 0 else 512, 3), gain=1.4)
    # f0 = |w.x + 80| with w.x ~ N(0, 100^2): 0-350 Hz, a mix of voiced (>10 Hz) and unvoiced frames
    g.mat('f0_predictor.classifier.weight', (1, 512), gain=25.0)
    g.sd['f0_predictor.classifier.bias'] = torch.tensor([80.0])
    return g.sd


def synthetic_inputs(seed=1986, text_len=50, prompt_len=255, prompt_text_len=0):
    """text ids, prompt speech ids, prompt mel N(-4, 2^2) clipped to [-11.5, 2], spk emb N(0,1)."""
    g = torch.Generator(device='cpu')
    g.manual_seed(seed)
    return dict(
        text=torch.randint(0, 151000, (1, text_len), generator=g, dtype=torch.int32),
        prompt_text=torch.randint(0, 151000, (1, prompt_text_len), generator=g, dtype=torch.int32),
        prompt_token=torch.randint(0, 6561, (1, prompt_len), generator=g, dtype=torch.int32),
        prompt_feat=(torch.randn(1, 2 * prompt_len, 80, generator=g) * 2 - 4).clamp(-11.5, 2.0),
        embedding=torch.randn(1, 192, generator=g),
    )


# ---- heavy-tailed variants (parity off the Gaussian) --------------------------------------------------------------------------
def heavy_tail_llm(sd, seed=7, massive=(7, 300, 555), massive_scale=400.0, bias_outlier=8.0, boost=2.5):
    """A copy of an LLM checkpoint with the statistics real Qwen2 checkpoints show and N(0, s) weights do not, built so that the
    network stays WELL-CONDITIONED (a first version that simply multiplied gains made the 24-layer net chaotic: its own fp32 and
    fp64 evaluations differed by 17 %, which no implementation can be compared against):
      * RMSNorm gains spread over [0.05, 30]: every gain vector is multiplied by a log-normal factor s (sigma 1.0) and the columns
        of the matrices that consume the normalised vector (q / k / v, gate / up, llm_decoder) are divided by s -- the function of
        the real-valued network is unchanged, its operands are not: products of huge and tiny factors, as in trained checkpoints;
      * 'massive activation' residual channels: the first layer's down projection writes values ~400x larger into three channels;
        from then on those channels dominate every RMS, their gains are small and the other channels' gains `boost` x larger, the
        compensation trained models show;
      * outlier q / k biases and embedding rows of very different norms.
    Conditioning of the 24-layer result (its fp32 against its fp64 evaluation, prefill logits): 1.5e-6, as on the Gaussian weights."""
    g = torch.Generator(device='cpu')
    g.manual_seed(seed)
    out = {k: v.clone() for k, v in sd.items()}
    hidden = out['llm.model.model.norm.weight'].numel()
    layers = 0
    while f'llm.model.model.layers.{layers}.input_layernorm.weight' in out:
        layers += 1
    mc = torch.tensor([c for c in massive if c < hidden])

    def regain(norm_key, consumers, after_massive):
        w = out[norm_key].clone()
        if after_massive:
            w = w * boost
            w[mc] = 0.05
        s = torch.exp(torch.randn(hidden, generator=g) * 1.0)
        s = torch.minimum(torch.maximum(s, 0.05 / w.abs().clamp_min(1e-6)), 30.0 / w.abs().clamp_min(1e-6))      # keep w * s inside [0.05, 30]
        out[norm_key] = w * s
        for k in consumers:
            out[k] = out[k] / s.view(1, -1)
    for i in range(layers):
        p = f'llm.model.model.layers.{i}.'
        regain(p + 'input_layernorm.weight', [p + f'self_attn.{n}_proj.weight' for n in 'qkv'], i >= 1)
        regain(p + 'post_attention_layernorm.weight', [p + 'mlp.gate_proj.weight', p + 'mlp.up_proj.weight'], i >= 1)
        for n in ('q_proj', 'k_proj'):
            b = out[p + f'self_attn.{n}.bias']
            idx = torch.randint(0, b.numel(), (6,), generator=g)
            b[idx] = (torch.rand(6, generator=g) * 2 - 1) * bias_outlier
    out['llm.model.model.layers.0.mlp.down_proj.weight'][mc] *= massive_scale
    regain('llm.model.model.norm.weight', ['llm_decoder.weight'], True)
    out['llm_decoder.weight'] = out['llm_decoder.weight'] * 3.0           # the massive channels shrink every normalised vector: usable top-1 margins
    for name in ('speech_embedding.weight', 'llm.model.model.embed_tokens.weight'):
        e = out[name]
        rows = torch.randint(0, e.shape[0], (max(8, e.shape[0] // 50),), generator=g)
        e[rows] *= 4.0
    if 'llm.model.lm_head.weight' in out:
        out['llm.model.lm_head.weight'] = out['llm.model.model.embed_tokens.weight']
    return out


def heavy_tail_flow(sd, seed=8):
    """A copy of a flow checkpoint with LayerNorm gains over [0.1, 10] (log-normal) and every convolution / linear weight scaled
    per output channel by a factor spread over 10x (log-uniform in [0.3, 3]): the products keep their shapes, their operands lose
    the uniform scale that makes bf16 rounding benign."""
    g = torch.Generator(device='cpu')
    g.manual_seed(seed)
    out = {k: v.clone() for k, v in sd.items()}
    for k, v in out.items():
        if not k.startswith('decoder.estimator') and not k.startswith('encoder.'):
            continue
        if k.endswith('.weight') and v.dim() == 1:                                        # LayerNorm gains
            out[k] = torch.exp(torch.randn(v.numel(), generator=g) * 0.8).clamp(0.1, 10.0)
        elif k.endswith('.weight') and v.dim() >= 2 and 'pos_bias' not in k:
            s = torch.exp(torch.empty(v.shape[0]).uniform_(math.log(0.3), math.log(3.0), generator=g))
            out[k] = v * s.view(-1, *([1] * (v.dim() - 1)))
    return out
