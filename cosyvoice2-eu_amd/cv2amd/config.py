"""`cosyvoice2.yaml` reader (replaces `load_hyperpyyaml` at cosyvoice/cli/cosyvoice.py:176-226 for the keys the hot path uses).

The reference instantiates its whole object graph from the HyperPyYAML file (`!new:` builds modules, `!name:` partials,
`!ref <key>` references, `!apply:` calls).  The MI355X build has a fixed architecture in HIP, so the file is read for
  * the scalar hyper-parameters that are RUN-TIME parameters of the engines (sampler constants conf/cosyvoice2.yaml:33-37,
    `inference_cfg_rate` :74, `pre_lookahead_len` :48, chunk sizes :17,:64,:87, NSF constants :94-96, `audio_limit` :108,
    `sample_rate` :8), and
  * the architectural dimensions, which are CHECKED against what the kernels implement (a different value raises ValueError
    instead of silently synthesising with the wrong network).
Tags are parsed structurally (no class is imported, nothing is executed).
"""
import re

import yaml


class _Loader(yaml.SafeLoader):
    pass


def _tagged(loader, suffix, node):
    if isinstance(node, yaml.MappingNode):
        v = loader.construct_mapping(node, deep=True)
    elif isinstance(node, yaml.SequenceNode):
        v = loader.construct_sequence(node, deep=True)
    else:
        v = loader.construct_scalar(node)
    return v


def _ref(loader, node):
    return _Ref(loader.construct_scalar(node))


class _Ref(str):
    """`!ref <a>` or an arithmetic expression over references (`<chunk_size> * <token_mel_ratio>`)."""


_Loader.add_constructor('!ref', _ref)
for _p in ('!new:', '!name:', '!apply:', '!module:'):
    _Loader.add_multi_constructor(_p, _tagged)


def _resolve(v, root, depth=0):
    if depth > 16:
        raise ValueError('cosyvoice2.yaml: reference cycle')
    if isinstance(v, _Ref):
        names = re.findall(r'<([A-Za-z0-9_.]+)>', v)
        if len(names) == 1 and v.strip() == '<{}>'.format(names[0]):
            return _resolve(_lookup(root, names[0]), root, depth + 1)
        expr = v
        for n in names:
            expr = expr.replace('<{}>'.format(n), repr(_resolve(_lookup(root, n), root, depth + 1)))
        if not re.fullmatch(r'[0-9eE+\-*/(). ]+', expr):
            raise ValueError('cosyvoice2.yaml: unsupported !ref expression {!r}'.format(str(v)))
        return eval(expr, {'__builtins__': {}}, {})          # digits and arithmetic operators only (checked above)
    if isinstance(v, dict):
        return {k: _resolve(x, root, depth) for k, x in v.items()}
    if isinstance(v, list):
        return [_resolve(x, root, depth) for x in v]
    return v


def _lookup(root, dotted):
    cur = root
    for part in dotted.split('.'):
        cur = cur[part]
    return cur


def parse(text, overrides=None):
    """YAML text -> plain nested dict with references resolved.  `overrides` replaces top-level keys first (the reference passes
    {'qwen_pretrain_path': ...}, cli/cosyvoice.py:222-226)."""
    raw = yaml.load(text, Loader=_Loader) or {}
    raw.update(overrides or {})
    return _resolve(raw, raw)


# what the HIP kernels implement (conf/cosyvoice2.yaml:23-112)
_FIXED = {
    'llm.speech_token_size': 6561,
    'flow.input_size': 512, 'flow.output_size': 80, 'flow.spk_embed_dim': 192, 'flow.vocab_size': 6561, 'flow.token_mel_ratio': 2,
    'flow.encoder.output_size': 512, 'flow.encoder.attention_heads': 8, 'flow.encoder.linear_units': 2048, 'flow.encoder.num_blocks': 6,
    'flow.encoder.input_layer': 'linear', 'flow.encoder.pos_enc_layer_type': 'rel_pos_espnet',
    'flow.encoder.selfattention_layer_type': 'rel_selfattn', 'flow.encoder.normalize_before': True,
    'flow.encoder.use_cnn_module': False, 'flow.encoder.macaron_style': False,
    'flow.decoder.in_channels': 240, 'flow.decoder.spk_emb_dim': 80,
    'flow.decoder.cfm_params.content.t_scheduler': 'cosine', 'flow.decoder.cfm_params.content.solver': 'euler',
    'flow.decoder.estimator.in_channels': 320, 'flow.decoder.estimator.out_channels': 80, 'flow.decoder.estimator.channels': [256],
    'flow.decoder.estimator.attention_head_dim': 64, 'flow.decoder.estimator.n_blocks': 4,
    'flow.decoder.estimator.num_mid_blocks': 12, 'flow.decoder.estimator.num_heads': 8, 'flow.decoder.estimator.act_fn': 'gelu',
    'flow.decoder.estimator.num_decoding_left_chunks': -1,
    'hift.in_channels': 80, 'hift.base_channels': 512, 'hift.nb_harmonics': 8, 'hift.upsample_rates': [8, 5, 3],
    'hift.upsample_kernel_sizes': [16, 11, 7], 'hift.istft_params.n_fft': 16, 'hift.istft_params.hop_len': 4,
    'hift.resblock_kernel_sizes': [3, 7, 11], 'hift.resblock_dilation_sizes': [[1, 3, 5]] * 3,
    'hift.source_resblock_kernel_sizes': [7, 7, 11], 'hift.source_resblock_dilation_sizes': [[1, 3, 5]] * 3,
    'hift.lrelu_slope': 0.1, 'hift.nsf_alpha': 0.1, 'hift.nsf_sigma': 0.003, 'hift.nsf_voiced_threshold': 10, 'hift.audio_limit': 0.99,
    'hift.f0_predictor.num_class': 1, 'hift.f0_predictor.in_channels': 80, 'hift.f0_predictor.cond_channels': 512,
    'flow.encoder.static_chunk_size': 25, 'flow.decoder.estimator.static_chunk_size': 50, 'flow.pre_lookahead_len': 3,
}


class Config:
    """Hyper-parameters of one model directory.  Defaults = the values of examples/libritts/cosyvoice2/conf/cosyvoice2.yaml."""

    def __init__(self):
        self.sample_rate = 24000
        self.sampling = dict(top_p=0.8, top_k=25, win_size=10, tau_r=0.1)
        self.inference_cfg_rate = 0.7
        self.n_timesteps = 10                 # flow.inference passes n_timesteps=10 (flow/flow.py:279); not in the yaml
        self.token_mel_ratio = 2
        self.pre_lookahead_len = 3
        self.input_frame_rate = 25
        self.chunk_size = 25
        self.qwen_pretrain_path = ''
        self.allowed_special = 'all'
        self.raw = None

    @classmethod
    def from_file(cls, path, overrides=None):
        with open(path, 'r') as f:
            return cls.from_dict(parse(f.read(), overrides))

    @classmethod
    def from_dict(cls, d):
        c = cls()
        c.raw = d
        for dotted, want in _FIXED.items():
            try:
                got = _lookup(d, dotted)
            except (KeyError, TypeError):
                continue                                        # key absent: the class default of the reference applies
            if isinstance(want, float):
                ok = abs(float(got) - want) <= 1e-9 * max(1.0, abs(want))
            else:
                ok = got == want
            if not ok:
                raise ValueError('cosyvoice2.yaml: {} = {!r} but the MI355X kernels implement {!r} (fixed architecture)'.format(dotted, got, want))
        c.sample_rate = int(d.get('sample_rate', c.sample_rate))
        if c.sample_rate != 24000:
            raise ValueError('cosyvoice2.yaml: sample_rate {} (the HiFT kernels generate 24000 Hz)'.format(c.sample_rate))
        samp = (d.get('llm') or {}).get('sampling') or {}
        for k in c.sampling:
            if k in samp:
                c.sampling[k] = type(c.sampling[k])(samp[k])
        if not (1 <= c.sampling['top_k'] <= 25 and 0 <= c.sampling['win_size'] <= 64 and c.sampling['top_p'] > 0):
            raise ValueError('cosyvoice2.yaml: sampler constants {} outside what k_sample supports (top_k 1..25, win_size 0..64)'.format(c.sampling))
        flow = d.get('flow') or {}
        cfm = (((flow.get('decoder') or {}).get('cfm_params') or {}).get('content')) or {}
        c.inference_cfg_rate = float(cfm.get('inference_cfg_rate', c.inference_cfg_rate))
        c.input_frame_rate = int(flow.get('input_frame_rate', c.input_frame_rate))
        c.token_mel_ratio = int(flow.get('token_mel_ratio', c.token_mel_ratio))
        c.pre_lookahead_len = int(flow.get('pre_lookahead_len', c.pre_lookahead_len))
        c.chunk_size = int(d.get('chunk_size', c.chunk_size))
        c.qwen_pretrain_path = d.get('qwen_pretrain_path', '') or ''
        c.allowed_special = d.get('allowed_special', 'all')
        return c
