"""Import harness for the REAL reference modules (build container only).

TEST INFRASTRUCTURE.  This file is used only by `tests/golden/make_golden.py`
and by the `-m "not gpu"` tests that are skipped when `/root/reference` is
absent (it does not exist on the GPU box).  Nothing in the product path
imports it.

It makes `cosyvoice.*` / `matcha.*` under `/root/reference/cosy_repo`
importable in this container by stubbing the third-party modules that are
imported but not used by the inference arithmetic, and
by restating the four `diffusers==0.29.0` symbols the CFM estimator uses
(`matcha/models/components/transformer.py:5-14`, `.../decoder.py:8`).

The reference package is loaded under its own module names (`cosyvoice`,
`matcha`), which collide with this repo's drop-in package `cosyvoice`.  Call
`activate()` before importing and `deactivate()` after to swap `sys.modules`
entries, or run it in a separate process (what make_golden.py does).
"""
import importlib.machinery
import logging
import math
import os
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

REF_ROOT = '/root/reference/cosy_repo'
REF_PATHS = [REF_ROOT, REF_ROOT + '/third_party/Matcha-TTS']


def available():
    return os.path.isdir(REF_ROOT + '/cosyvoice')


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class _AttrDict(dict):
    """omegaconf.DictConfig stand-in: attribute access over `content`."""

    def __init__(self, content=None, **kw):
        super().__init__(content or {}, **kw)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


# ---- diffusers 0.29.0 restatement (only what matcha's transformer block uses) ----
class _LoRACompatibleLinear(nn.Linear):
    def forward(self, x, scale=1.0):
        return super().forward(x)


class _GELU(nn.Module):
    # diffusers.models.activations.GELU: proj Linear then F.gelu(approximate)
    def __init__(self, dim_in, dim_out, approximate='none', bias=True):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out, bias=bias)
        self.approximate = approximate

    def forward(self, x):
        return F.gelu(self.proj(x), approximate=self.approximate)


class _Attention(nn.Module):
    # diffusers.models.attention_processor.Attention with AttnProcessor2_0,
    # self-attention only: q/k/v Linear(query_dim, heads*dim_head, bias=bias),
    # to_out = [Linear(inner, query_dim, bias=True), Dropout]; SDPA with additive mask.
    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, dropout=0.0,
                 bias=False, upcast_attention=False, **kw):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        self.to_q = nn.Linear(query_dim, inner, bias=bias)
        self.to_k = nn.Linear(query_dim, inner, bias=bias)
        self.to_v = nn.Linear(query_dim, inner, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim, bias=True), nn.Dropout(dropout)])

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
        b, t, _ = hidden_states.shape
        h = self.heads
        q = self.to_q(hidden_states).view(b, t, h, -1).transpose(1, 2)
        k = self.to_k(hidden_states).view(b, t, h, -1).transpose(1, 2)
        v = self.to_v(hidden_states).view(b, t, h, -1).transpose(1, 2)
        if attention_mask is not None:
            # prepare_attention_mask: repeat_interleave over heads, view (b, h, -1, t)
            attention_mask = attention_mask.repeat_interleave(h, dim=0).view(b, h, -1, attention_mask.shape[-1])
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=attention_mask, dropout_p=0.0, is_causal=False)
        o = o.transpose(1, 2).reshape(b, t, -1).to(q.dtype)
        return self.to_out[1](self.to_out[0](o))


def _get_activation(name):
    return {'silu': nn.SiLU(), 'swish': nn.SiLU(), 'mish': nn.Mish(), 'gelu': nn.GELU(), 'relu': nn.ReLU()}[name.lower()]


_SAVED = {}
_OURS = ('cosyvoice', 'matcha')


def activate():
    """Put the reference on sys.path, install stubs, hide this repo's `cosyvoice` package."""
    assert available(), 'reference not mounted'
    for k in list(sys.modules):
        if k.split('.')[0] in _OURS:
            _SAVED[k] = sys.modules.pop(k)
    for p in reversed(REF_PATHS):
        if p not in sys.path:
            sys.path.insert(0, p)
    ta = _stub('torchaudio')
    _stub('torchaudio.compliance', kaldi=types.SimpleNamespace())
    _stub('torchaudio.compliance.kaldi')
    ta.transforms = types.SimpleNamespace(Resample=None)
    _stub('omegaconf', DictConfig=_AttrDict)
    _stub('onnxruntime')
    _stub('whisper')
    _stub('hyperpyyaml', load_hyperpyyaml=None)
    # imported by cli/frontend.py at module level, used only inside CosyVoiceFrontEnd.__init__ (never called here)
    _stub('inflect', engine=None)
    _stub('tn'); _stub('tn.chinese'); _stub('tn.english')
    _stub('tn.chinese.normalizer', Normalizer=None)
    _stub('tn.english.normalizer', Normalizer=None)
    _stub('modelscope', snapshot_download=None)
    _stub('conformer', ConformerBlock=type('ConformerBlock', (nn.Module,), {}))
    _stub('matcha.utils.pylogger', get_pylogger=lambda n=None: logging.getLogger(n))
    _stub('diffusers')
    _stub('diffusers.models')
    _stub('diffusers.models.activations', get_activation=_get_activation)
    _stub('diffusers.models.attention', GEGLU=None, GELU=_GELU, AdaLayerNorm=None, AdaLayerNormZero=None,
          ApproximateGELU=None)
    _stub('diffusers.models.attention_processor', Attention=_Attention)
    _stub('diffusers.models.lora', LoRACompatibleLinear=_LoRACompatibleLinear)
    _stub('diffusers.utils')
    _stub('diffusers.utils.torch_utils', maybe_allow_in_graph=lambda c: c)
    logging.getLogger().setLevel(logging.WARNING)


def deactivate():
    for k in list(sys.modules):
        if k.split('.')[0] in _OURS + ('diffusers', 'conformer', 'torchaudio', 'omegaconf', 'onnxruntime', 'whisper',
                                       'hyperpyyaml', 'modelscope', 'inflect', 'tn'):
            sys.modules.pop(k)
    for p in REF_PATHS:
        if p in sys.path:
            sys.path.remove(p)
    sys.modules.update(_SAVED)
    _SAVED.clear()


# ---------------------------------------------------------------------------
# builders: the reference's own classes at cosyvoice2.yaml dims
# (examples/libritts/cosyvoice2/conf/cosyvoice2.yaml:23-112)
# ---------------------------------------------------------------------------
def build_hift():
    from cosyvoice.hifigan.generator import HiFTGenerator
    from cosyvoice.hifigan.f0_predictor import ConvRNNF0Predictor
    return HiFTGenerator(in_channels=80, base_channels=512, nb_harmonics=8, sampling_rate=24000, nsf_alpha=0.1,
                         nsf_sigma=0.003, nsf_voiced_threshold=10, upsample_rates=[8, 5, 3],
                         upsample_kernel_sizes=[16, 11, 7], istft_params={'n_fft': 16, 'hop_len': 4},
                         resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5]] * 3,
                         source_resblock_kernel_sizes=[7, 7, 11], source_resblock_dilation_sizes=[[1, 3, 5]] * 3,
                         lrelu_slope=0.1, audio_limit=0.99,
                         f0_predictor=ConvRNNF0Predictor(num_class=1, in_channels=80, cond_channels=512)).eval()


def build_flow(num_mid_blocks=12, n_blocks=4, enc_blocks=6, up_blocks=4):
    from cosyvoice.flow.flow import CausalMaskedDiffWithXvec
    from cosyvoice.flow.flow_matching import CausalConditionalCFM
    from cosyvoice.flow.decoder import CausalConditionalDecoder
    from cosyvoice.transformer.upsample_encoder import UpsampleConformerEncoder
    enc = UpsampleConformerEncoder(output_size=512, attention_heads=8, linear_units=2048, num_blocks=enc_blocks,
                                   dropout_rate=0.1, positional_dropout_rate=0.1, attention_dropout_rate=0.1,
                                   normalize_before=True, input_layer='linear', pos_enc_layer_type='rel_pos_espnet',
                                   selfattention_layer_type='rel_selfattn', input_size=512, use_cnn_module=False,
                                   macaron_style=False, static_chunk_size=25)
    if up_blocks != 4:
        enc.up_encoders = enc.up_encoders[:up_blocks]
    est = CausalConditionalDecoder(in_channels=320, out_channels=80, channels=[256], dropout=0.0,
                                   attention_head_dim=64, n_blocks=n_blocks, num_mid_blocks=num_mid_blocks,
                                   num_heads=8, act_fn='gelu', static_chunk_size=50, num_decoding_left_chunks=-1)
    cfm = CausalConditionalCFM(in_channels=240, n_spks=1, spk_emb_dim=80,
                               cfm_params=_AttrDict(dict(sigma_min=1e-06, solver='euler', t_scheduler='cosine',
                                                         training_cfg_rate=0.2, inference_cfg_rate=0.7,
                                                         reg_loss_type='l1')),
                               estimator=est)
    return CausalMaskedDiffWithXvec(input_size=512, output_size=80, spk_embed_dim=192, output_type='mel',
                                    vocab_size=6561, input_frame_rate=25, only_mask_loss=True, token_mel_ratio=2,
                                    pre_lookahead_len=3, encoder=enc, decoder=cfm).eval()


def build_llm(num_layers=24, sampling=None):
    """Qwen2LM over an HFBackbone holding a random-init HF Qwen2ForCausalLM (no weights on disk)."""
    from cosyvoice.llm.llm import Qwen2LM, HFBackbone
    from cosyvoice.utils.common import ras_sampling
    from transformers import Qwen2Config, Qwen2ForCausalLM
    cfg = Qwen2Config(vocab_size=151936, hidden_size=896, intermediate_size=4864, num_hidden_layers=num_layers,
                      num_attention_heads=14, num_key_value_heads=2, rope_theta=1e6, rms_norm_eps=1e-6,
                      max_position_embeddings=32768, tie_word_embeddings=True)
    bb = HFBackbone.__new__(HFBackbone)
    nn.Module.__init__(bb)
    bb.pretrain_path = '/nonexistent/CosyVoice-BlankEN'     # keeps the unistream path (llm/llm.py:597-601)
    bb.model = Qwen2ForCausalLM(cfg)
    # transformers here is 5.15, the reference pins 4.40.1 (cosy_repo/requirements.txt:37).  The reference passes
    # `masks[:, -1, :]` = an all-ones (1, q_len) mask even when the KV cache is longer (llm/llm.py:107-117, 686-688).
    # 4.40.1 drops an all-ones 2-D mask (modeling_attn_mask_utils._ignore_causal_mask_sdpa: query_length == 1 or
    # kv_len == q_len -> mask None, plain causal attention); 5.15 applies it as a 1-key mask instead.  Restore the
    # pinned behaviour: an all-ones mask is passed as None.
    _fwd = bb.model.forward

    def _forward_pinned(*a, attention_mask=None, **k):
        if attention_mask is not None and bool(torch.all(attention_mask == 1)):
            attention_mask = None
        return _fwd(*a, attention_mask=attention_mask, **k)
    bb.model.forward = _forward_pinned
    import functools
    samp = sampling or functools.partial(ras_sampling, top_p=0.8, top_k=25, win_size=10, tau_r=0.1)
    return Qwen2LM(llm_input_size=0, llm_output_size=0, speech_token_size=6561, llm=bb, sampling=samp,
                   length_normalized_loss=True, lsm_weight=0, mix_ratio=[5, 15]).eval()


class _CacheView:
    """What `inference_bistream` needs from the KV cache of transformers 4.40.1 (a tuple of (k, v) per layer): `cache[0][0].size(2)`
    = cached length (llm/llm.py:794, 820).  transformers 5 returns a DynamicCache without that indexing; this view answers the one
    question and hands the real object back to the model."""

    def __init__(self, inner):
        self.inner = inner

    class _Len:
        def __init__(self, n):
            self.n = n

        def size(self, dim):
            assert dim == 2
            return self.n

    def __getitem__(self, i):
        return (self._Len(self.inner.get_seq_length()),)


def enable_bistream(llm):
    """Make Qwen2LM.inference_bistream of the reference runnable under the installed transformers: forward_one_step keeps its
    signature and arithmetic, only the cache object is wrapped in _CacheView on the way out and unwrapped on the way in."""
    bb = llm.llm
    orig = bb.forward_one_step

    def forward_one_step(xs, masks, cache=None):
        y, new = orig(xs, masks, cache.inner if isinstance(cache, _CacheView) else cache)
        return y, _CacheView(new)
    bb.forward_one_step = forward_one_step
    return llm


def greedy_sampling_ids(self, weighted_scores, decoded_tokens, sampling, ignore_eos=True):
    """Harness-defined greedy: what `sampling_ids` (llm/llm.py:235-250) converges to with a deterministic
    sampler -- argmax with EOS (speech_token_size) excluded while ignore_eos, instead of 100 identical re-draws."""
    s = weighted_scores.clone()
    if ignore_eos:
        s[self.speech_token_size] = -float('inf')
    return s.argmax(dim=-1, keepdim=True)
