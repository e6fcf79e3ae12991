"""Checkpoint -> HBM layout.  Runs once at load time (torch is used as the allocator / copy engine).

LLM GEMM matrices are rounded to bf16 and re-ordered into 1 KiB MFMA-operand blocks (layout: include/cv2_amd.h).
`round_llm_sd` applies the same rounding to a state dict so the CPU oracle can be run on identical weights.
"""
import torch

LLM_GEMM_KEYS = ('q_proj.weight', 'k_proj.weight', 'v_proj.weight', 'o_proj.weight', 'gate_proj.weight', 'up_proj.weight',
                 'down_proj.weight')


def bf16_round(w):
    return w.to(torch.bfloat16).to(torch.float32)


def round_llm_sd(sd):
    """State dict with exactly the values the HIP path multiplies by (bf16-rounded GEMM matrices, fp32 elsewhere)."""
    out = {}
    for k, v in sd.items():
        if k.endswith(LLM_GEMM_KEYS) or k == 'llm_decoder.weight':
            out[k] = bf16_round(v)
        else:
            out[k] = v
    return out


def pack_bf16(w):
    """[N, K] fp32/bf16 -> packed bf16 (uint16 view as int16 tensor) in [N/16][K/32][64 lanes][8] order; N padded to 16."""
    n, k = w.shape
    assert k % 32 == 0, 'K must be a multiple of 32'
    npad = (n + 15) // 16 * 16
    wb = torch.zeros(npad, k, dtype=torch.bfloat16, device=w.device)
    wb[:n] = w.to(torch.bfloat16)
    # [nt, r(16), ks, h(4), j(8)] -> [nt, ks, h, r, j]
    p = wb.view(npad // 16, 16, k // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous()
    return p.view(torch.int16).reshape(-1)


def interleave_tiles(a, b):
    """Row tiles of 16 interleaved: a-tile 0, b-tile 0, a-tile 1, ...  (gate/up for the fused SwiGLU epilogue)."""
    n, k = a.shape
    assert n % 16 == 0 and a.shape == b.shape
    return torch.stack([a.view(n // 16, 16, k), b.view(n // 16, 16, k)], dim=1).reshape(2 * n, k)


def rope_tables(max_pos, theta=1e6, dim=64):
    """cos/sin [max_pos, dim/2] computed on the host exactly as HF's Qwen2RotaryEmbedding does (fp32)."""
    inv = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float32) / dim))
    fr = torch.arange(max_pos, dtype=torch.float32)[:, None] * inv[None, :]
    return fr.cos().contiguous(), fr.sin().contiguous()
