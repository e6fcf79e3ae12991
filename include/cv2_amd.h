/* cv2_amd.h — C ABI of libcv2amd.so: the MI355X (gfx950) hot path of CosyVoice2-0.5B-EU synthesis.
 *
 * Drop-in boundary.  The reference is pure Python; the seams these entry points replace are
 *   LLM   : Qwen2LM.inference / inference_wrapper            cosy_repo/cosyvoice/llm/llm.py:575-719
 *           (precedent for an external engine: the vLLM branch, llm.py:651-680 — add_request(prompt_embeds),
 *            step(), queue of ids)
 *   Flow  : CausalMaskedDiffWithXvec.inference               cosy_repo/cosyvoice/flow/flow.py:235-283
 *           ConditionalCFM.forward_estimator (TensorRT seam) cosy_repo/cosyvoice/flow/flow_matching.py:125-150
 *   HiFT  : HiFTGenerator.inference                          cosy_repo/cosyvoice/hifigan/generator.py:570-582
 *   fade  : fade_in_out                                      cosy_repo/cosyvoice/utils/common.py:142-150
 *
 * Conventions: plain pointers and sizes only; every tensor argument is a DEVICE pointer owned by the caller
 * (the Python host allocates with torch); the library never frees caller memory and never returns owned
 * memory except opaque handles; all launches go to the hipStream_t passed in (as void*); return 0 on success,
 * negative on error with a message available from cv2_last_error().
 */
#ifndef CV2_AMD_H
#define CV2_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char* cv2_last_error(void);
int cv2_version(void);

/* ------------------------------------------------------------------------------------------------
 * Stage 1 — Qwen2 speech-token LM (replaces llm.py:575-719; HF Qwen2ForCausalLM call sites llm.py:107-117)
 *
 * Weight layout in HBM ("packed"): a [N,K] bf16 matrix is stored as 1 KiB blocks in MFMA 16x16x32 operand
 * order: block (nt, ks) holds rows 16nt..16nt+15, columns 32ks..32ks+31; within a block lane l (0..63) owns
 * 8 consecutive bf16 of row 16nt + (l & 15), columns 32ks + 8(l >> 4) .. +7.  Blocks are ordered [nt][ks].
 * A wave streams one row tile with perfectly sequential 1 KiB loads.  cv2amd/weights.py builds this layout.
 * ---------------------------------------------------------------------------------------------- */
typedef struct cv2_llm cv2_llm;

typedef struct {
    int32_t hidden;     /* 896  */
    int32_t inter;      /* 4864 */
    int32_t layers;     /* 24   */
    int32_t n_q;        /* 14 query heads */
    int32_t n_kv;       /* 2 kv heads     */
    int32_t vocab;      /* 6564 = speech_token_size + 3 (llm.py:392-397) */
    int32_t vocab_pad;  /* vocab rounded up to 16 */
    int32_t eos;        /* 6561 */
    int32_t max_seqs;   /* sequence slots (<= 32) */
    int32_t max_pos;    /* KV capacity per sequence */
    int32_t max_out;    /* capacity of out_tokens per sequence */
    float rms_eps;      /* 1e-6 */
} cv2_llm_dims;

typedef struct {
    const uint16_t* wqkv;  /* packed [(n_q+2n_kv)*64, hidden]: q rows, then k rows, then v rows */
    const float* bqkv;     /* [(n_q+2n_kv)*64] */
    const uint16_t* wo;    /* packed [hidden, n_q*64] */
    const uint16_t* wgu;   /* packed [2*inter, hidden]; row tiles interleaved: gate tile i, up tile i, ... */
    const uint16_t* wdown; /* packed [hidden, inter] */
    const float* ln1;      /* input_layernorm.weight [hidden] */
    const float* ln2;      /* post_attention_layernorm.weight [hidden] */
} cv2_llm_layer;

typedef struct {
    const cv2_llm_layer* layers; /* HOST array of `layers` entries holding device pointers */
    const float* final_norm;     /* [hidden] */
    const uint16_t* wdec;        /* packed [vocab_pad, hidden] (llm_decoder, zero rows beyond vocab) */
    const float* bdec;           /* [vocab_pad] */
    const float* speech_emb;     /* [vocab, hidden] fp32 (speech_embedding.weight) */
    const float* rope_cos;       /* [max_pos, 32] cos(pos * theta^(-2i/64)) — built on the host as HF does */
    const float* rope_sin;       /* [max_pos, 32] */
} cv2_llm_weights;

/* per-sequence state, int32[CV2_LLM_STATE_STRIDE] per slot, device memory owned by the caller */
enum {
    CV2_ST_POS = 0,      /* next KV position */
    CV2_ST_STEP = 1,     /* loop index i of inference_wrapper (llm.py:684) */
    CV2_ST_NOUT = 2,     /* emitted tokens so far */
    CV2_ST_DONE = 3,     /* 1 once EOS was drawn or max_len reached */
    CV2_ST_MINLEN = 4,   /* int(text_len * min_token_text_ratio) */
    CV2_ST_MAXLEN = 5,   /* int(text_len * max_token_text_ratio) */
    CV2_ST_MODE = 6,     /* 0 greedy (harness-defined), 1 RAS top-p 0.8 / top-k 25 / win 10 / tau 0.1 */
    CV2_ST_FORCE = 7,    /* 1: synthetic-weights mode, ids >= eos never drawn (fixed decode length) */
    CV2_ST_SEED_LO = 8,
    CV2_ST_SEED_HI = 9,
    CV2_ST_ERR = 10,     /* 1: sampler exhausted 100 EOS re-draws (RuntimeError in llm.py:249) */
    CV2_ST_LAST = 11,    /* last drawn id */
    CV2_LLM_STATE_STRIDE = 16
};

typedef struct {
    int32_t* state;      /* [max_seqs][CV2_LLM_STATE_STRIDE] */
    int32_t* out_tokens; /* [max_seqs][max_out] emitted speech tokens */
    float* logits;       /* [32][vocab_pad] last step's llm_decoder output (for parity tests) */
} cv2_llm_io;

size_t cv2_llm_workspace_bytes(const cv2_llm_dims* dims);
int cv2_llm_create(const cv2_llm_dims* dims, const cv2_llm_weights* w, const cv2_llm_io* io, void* workspace,
                   size_t workspace_bytes, cv2_llm** out);
int cv2_llm_destroy(cv2_llm* h);
/* Step 0 of inference_wrapper for slot `seq`: run the backbone over prompt_embeds [len, hidden] fp32
 * (= lm_input of llm.py:641, the vLLM seam's prompt_embeds), fill the KV cache, draw the first token.
 * state[seq] must have been initialised by the caller (pos = step = nout = done = 0, lens, mode, seed). */
int cv2_llm_prefill(cv2_llm* h, int32_t seq, const float* prompt_embeds, int32_t len, void* stream);
/* n_steps iterations of the decode loop for slots 0..n_seqs-1 in lock step (one hipGraph replay per step);
 * finished slots idle.  No host synchronisation inside. */
int cv2_llm_decode(cv2_llm* h, int32_t n_seqs, int32_t n_steps, void* stream);

/* Stand-alone skinny GEMM used by the LLM (exported for unit tests and for the flow time-MLP):
 * out[r][n] = sum_k W[n][k] x[r][k] (+bias[n]); rows <= 32; W packed; K % 32 == 0; N % 16 == 0. */
int cv2_skinny_gemm(const uint16_t* w_packed, const float* bias, const float* x, float* out, int32_t rows,
                    int32_t n, int32_t k, void* stream);

#ifdef __cplusplus
}
#endif
#endif
