"""Checkpoint validation: the `load_state_dict(strict=True)` of cosyvoice/cli/model.py:67-90 for engines that are not nn.Modules.

The reference's classes define which keys (and shapes) `llm.pt`, `flow.pt`, `hift.pt` must hold; a surplus key such as
`decoder.estimator...attn1.to_q.bias` (diffusers Attention built with `bias=True`) means the checkpoint was trained with
arithmetic the HIP kernels do not implement and must be refused, not ignored.  The schema is the key set of cv2amd/synth.py,
which tests/golden/make_golden.py loads into the reference's own classes with strict=True (so it equals their state_dict()).

Semantics kept from the reference (model.py:67-90):
  * flow / hift: strict — missing or unexpected keys, or a shape mismatch, raise RuntimeError with torch's message layout;
  * llm: strict first; on missing keys / size mismatch a warning and a non-strict second pass (unexpected keys ignored,
    missing keys tolerated only where the MI355X engine does not read them — a missing weight it needs raises, there being
    no "random initialisation" worth synthesising with);
  * hift keys may carry a `generator.` prefix (stripped, model.py:88); `epoch` / `step` entries of training checkpoints
    (utils/train_utils.py:214) are dropped.
"""
import logging

import torch

from . import synth

# tensors of llm.pt that the decode path never reads (the tied lm_head of the HF backbone, llm.py:322 is training-only)
LLM_UNUSED = ('llm.model.lm_head.weight',)


def schema(kind, llm_layers=24):
    """{key: shape} of `kind` in ('llm', 'flow', 'hift') at the cosyvoice2.yaml dimensions."""
    if kind == 'llm':
        sd = synth.make_llm(layers=llm_layers, meta=True)
    elif kind == 'flow':
        sd = synth.make_flow(meta=True)
    elif kind == 'hift':
        sd = synth.make_hift(meta=True)
    else:
        raise ValueError(kind)
    return {k: tuple(v.shape) for k, v in sd.items()}


def strip(sd, kind):
    """Checkpoint dict -> state dict: drop training metadata, strip the hifigan `generator.` prefix (model.py:88)."""
    out = {}
    for k, v in sd.items():
        if k in ('epoch', 'step') and not torch.is_tensor(v):
            continue
        if kind == 'hift':
            k = k.replace('generator.', '')
        out[k] = v
    return out


def _llm_layers(sd):
    n = 0
    while 'llm.model.model.layers.{}.input_layernorm.weight'.format(n) in sd:
        n += 1
    return n


def check(sd, kind, strict=True):
    """Returns (missing, unexpected, mismatched) key lists; raises RuntimeError like load_state_dict(strict=True) when strict."""
    want = schema(kind, _llm_layers(sd) or 24) if kind == 'llm' else schema(kind)
    missing = [k for k in want if k not in sd]
    unexpected = [k for k in sd if k not in want]
    mism = ['size mismatch for {}: copying a param with shape {} from checkpoint, the shape in current model is {}.'.format(
        k, tuple(sd[k].shape), want[k]) for k in want if k in sd and torch.is_tensor(sd[k]) and tuple(sd[k].shape) != want[k]]
    if strict and (missing or unexpected or mism):
        msgs = []
        if unexpected:
            msgs.append('Unexpected key(s) in state_dict: {}. '.format(', '.join('"{}"'.format(k) for k in unexpected)))
        if missing:
            msgs.append('Missing key(s) in state_dict: {}. '.format(', '.join('"{}"'.format(k) for k in missing)))
        msgs += mism
        raise RuntimeError('Error(s) in loading state_dict for {}:\n\t{}'.format(kind, '\n\t'.join(msgs)))
    return missing, unexpected, mism


def validate_llm(sd):
    """model.py:67-82: strict, then the strict=False fallback for backbone mismatches."""
    try:
        check(sd, 'llm', strict=True)
        logging.info('Successfully validated LLM checkpoint with strict=True')
        return sd
    except RuntimeError as e:
        if 'Missing key(s) in state_dict' not in str(e) and 'size mismatch' not in str(e):
            raise
        logging.warning('Strict loading failed, trying with strict=False: %s', e)
    missing, unexpected, mism = check(sd, 'llm', strict=False)
    if unexpected:
        logging.warning('Unexpected keys (will be ignored): %s', unexpected)
    needed = [k for k in missing if k not in LLM_UNUSED]
    if needed or mism:
        raise RuntimeError('LLM checkpoint cannot drive the MI355X engine: missing {} ; {}'.format(needed, ' '.join(mism)))
    if missing:
        logging.warning('Missing keys (not read by the decode path): %s', missing)
    return {k: v for k, v in sd.items() if k not in unexpected}


def validate(llm_sd, flow_sd, hift_sd):
    llm_sd, flow_sd, hift_sd = strip(llm_sd, 'llm'), strip(flow_sd, 'flow'), strip(hift_sd, 'hift')
    llm_sd = validate_llm(llm_sd)
    check(flow_sd, 'flow', strict=True)
    check(hift_sd, 'hift', strict=True)
    return llm_sd, flow_sd, hift_sd


def verify_dir(model_dir, final=True, out=print):
    """`python -m cv2amd.checkpoint --verify <model_dir>`: the strict key / shape diff of llm.pt, flow.pt, hift.pt (`final=False`:
    the `-original` files of cli/cosyvoice.py:240-265) against the architecture the HIP engines implement, without touching a GPU.
    The first person with real weights learns from ONE command whether `load_state_dict(strict=True)` of the reference's classes would
    hold — in particular whether the estimator's attention projections carry biases (diffusers `Attention(bias=...)`, matcha
    transformer.py:168,201: the q / k / v-without-bias reading of SURVEY.md §8(c) is confirmed or refuted here).  Returns the number of
    problems found."""
    import os
    bad = 0
    suffix = '' if final else '-original'
    for kind in ('llm', 'flow', 'hift'):
        path = os.path.join(model_dir, '{}{}.pt'.format(kind, suffix))
        if not os.path.exists(path):
            out('{}: MISSING FILE {}'.format(kind, path))
            bad += 1
            continue
        sd = strip(torch.load(path, map_location='cpu', weights_only=True), kind)
        missing, unexpected, mism = check(sd, kind, strict=False)
        n_par = sum(v.numel() for v in sd.values() if torch.is_tensor(v))
        out('{}: {} tensors, {:.1f} M parameters in {}'.format(kind, sum(torch.is_tensor(v) for v in sd.values()), n_par / 1e6, path))
        unused = [k for k in missing if kind == 'llm' and k in LLM_UNUSED]
        missing = [k for k in missing if k not in unused]
        qkv_bias = [k for k in unexpected if any(k.endswith('attn1.to_{}.bias'.format(x)) for x in 'qkv')]
        for k in unused:
            out('  note: {} absent (not read by the decode path)'.format(k))
        for k in missing:
            out('  MISSING    {}'.format(k))
        for k in unexpected:
            out('  UNEXPECTED {} {}'.format(k, tuple(sd[k].shape) if torch.is_tensor(sd[k]) else type(sd[k]).__name__))
        for m in mism:
            out('  ' + m)
        if qkv_bias:
            out('  => the estimator was built with attention_bias=True ({} q/k/v bias tensors): the HIP kernels implement q / k / v WITHOUT '
                'bias (DESIGN.md section 2) and refuse this checkpoint'.format(len(qkv_bias)))
        elif kind == 'flow' and not (missing or unexpected or mism):
            out('  => strict load holds: q / k / v projections carry no bias, to_out carries one (the reading of SURVEY.md 8(c) is confirmed)')
        bad += len(missing) + len(unexpected) + len(mism)
    out('OK: the three checkpoints match the architecture key for key, shape for shape' if bad == 0 else '{} problem(s)'.format(bad))
    return bad


if __name__ == '__main__':
    import argparse
    import sys
    ap = argparse.ArgumentParser(description='strict key / shape check of a CosyVoice2 model directory against the MI355X engines (CPU only)')
    ap.add_argument('--verify', metavar='MODEL_DIR', required=True)
    ap.add_argument('--original', action='store_true', help='check llm-original.pt / flow-original.pt / hift-original.pt (setting="original")')
    a = ap.parse_args()
    sys.exit(1 if verify_dir(a.verify, final=not a.original) else 0)
