"""Which statistic of the heavy-tailed checkpoint breaks LLM parity?  Prefill logits HIP vs oracle, one feature at a time."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cosyvoice2-eu_amd'))
import torch
import torch.nn.functional as F
from cv2amd import synth, weights as W
from cv2amd.llm import LLMEngine
from oracle import llm as OL

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 4
base = synth.make_llm(layers=layers)
full = synth.heavy_tail_llm(base)


def variant(keys):
    sd = {k: v.clone() for k, v in base.items()}
    for k in full:
        if any(t in k for t in keys):
            sd[k] = full[k].clone()
    return sd


cases = {
    'gaussian': [],
    'norm gains': ['layernorm.weight', 'model.norm.weight'],
    'massive down_proj rows': ['layers.0.mlp.down_proj'],
    'gains + massive': ['layernorm.weight', 'model.norm.weight', 'layers.0.mlp.down_proj'],
    'q/k bias outliers': ['q_proj.bias', 'k_proj.bias'],
    'embedding rows': ['speech_embedding', 'embed_tokens', 'lm_head'],
    'all': [''],
}
inp = synth.synthetic_inputs(seed=77, text_len=25, prompt_len=70, prompt_text_len=5)
for name, keys in cases.items():
    sd = variant(keys) if keys != [''] else full
    sdr = W.round_llm_sd(sd)
    eng = LLMEngine(sd, 'cuda:0', max_seqs=2, max_pos=512, max_out=64)
    x = eng.build_lm_input(inp['text'], inp['prompt_text'], inp['prompt_token'])
    eng.add_request(0, x, 10, 10)
    torch.cuda.synchronize()
    got = eng.logits[(x.shape[0] - 1) % 32, :eng.vocab].cpu()
    d = OL.LLMDims(sdr)
    y = OL.qwen2_step(sdr, d, OL.build_lm_input(sdr, inp['text'], inp['prompt_text'], inp['prompt_token']), [None] * d.layers)
    want = F.linear(y[-1], sdr['llm_decoder.weight'], sdr['llm_decoder.bias'])
    err = float((got - want).abs().max() / want.abs().max())
    # the same through the batched GEMM prefill
    eng.add_requests([0], [x], [(10, 10)])
    torch.cuda.synchronize()
    got2 = eng.logits[0, :eng.vocab].cpu()
    err2 = float((got2 - want).abs().max() / want.abs().max())
    print(f'{name:26s} chunked prefill rel err {err:.3e}   batched prefill rel err {err2:.3e}   |y| max {float(y.abs().max()):.2e}', flush=True)
    del eng
