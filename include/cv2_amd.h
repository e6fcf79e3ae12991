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
/* ABI revision of the structs and entry points below; bumped whenever a struct's layout or a signature changes (2: cv2_hift_weights gained
 * sd_conv[3] in front of src_rb).  cv2_version() returns the library's; a binding built against another revision must refuse to run
 * (cv2amd/lib.py does) -- a stale libcv2amd.so would otherwise misread every pointer behind the changed field without any error. */
#define CV2_ABI_VERSION 2
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
    int32_t max_prefill_rows; /* capacity of cv2_llm_prefill_batch in prompt rows (0: only the 32-row cv2_llm_prefill) */
    /* ras_sampling constants (utils/common.py:111-139, values from conf/cosyvoice2.yaml:33-37); top_k == 0 selects 0.8 / 25 / 10 / 0.1 */
    float top_p;        /* nucleus mass */
    int32_t top_k;      /* nucleus size, 1..25 */
    int32_t win_size;   /* repetition window, <= 64 */
    float tau_r;        /* full-vocabulary re-draw when the drawn id occurs >= win_size * tau_r times in the window */
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
    CV2_ST_MODE = 6,     /* 0 greedy (harness-defined), 1 RAS with the constants of cv2_llm_dims */
    CV2_ST_FORCE = 7,    /* 1: synthetic-weights mode, ids >= eos never drawn (fixed decode length) */
    CV2_ST_SEED_LO = 8,
    CV2_ST_SEED_HI = 9,
    CV2_ST_ERR = 10,     /* 1: sampler exhausted 100 EOS re-draws (RuntimeError in llm.py:249); 2: bistream drew a special id it must
                          * not ("should not get token", ValueError in llm.py:809, 829) */
    CV2_ST_LAST = 11,    /* last drawn id */
    CV2_ST_BIMODE = 12,  /* inference_bistream (llm.py:721-834): 0 unistream, 1 text still expected, 2 final decode */
    CV2_ST_NEXTFILL = 13,/* next_fill_index of llm.py:783 (-1: none yet); counts entries of out_tokens incl. fill ids */
    CV2_ST_WAIT = 14,    /* 1: the slot stopped on the fill id (eos + 2) and waits for cv2_llm_extend with the next text block */
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
/* Step 0 for several slots at once through the MFMA GEMM path (operand split hi + lo): embeds = the prompts' rows
 * concatenated, fp32 [sum(lens)][hidden] on the device; slots / lens are HOST arrays of n entries.  state[slot] must be
 * initialised like for cv2_llm_prefill.  Weights are streamed once for all rows. */
int cv2_llm_prefill_batch(cv2_llm* h, int32_t n, const int32_t* slots, const int32_t* lens, const float* embeds, void* stream);
/* inference_bistream (llm.py:787-811, 817-832): `len` further input rows for slot `seq` at KV positions pos0 .. pos0 + len - 1
 * (embeds fp32 [len][hidden], device), then one draw from the last row under state[seq] (CV2_ST_BIMODE, CV2_ST_NEXTFILL).  In
 * bistream modes out_tokens receives EVERY drawn id (fill and EOS included), as the reference's out_tokens list does. */
int cv2_llm_extend(cv2_llm* h, int32_t seq, const float* embeds, int32_t len, int32_t pos0, void* stream);
/* The same for n slots in one pass over the weights (the matrix-core path of cv2_llm_prefill_batch): the text blocks that concurrent
 * inference_bistream calls (cli/model.py:120-128 x the streams of BASELINE configs[4]) are waiting to feed.  slots / lens / pos0 are HOST
 * arrays of n entries (distinct slots), embeds = the slots' rows concatenated, fp32 [sum(lens)][hidden] on the device; one draw per slot
 * from its last row under state[slot].  sum(lens) rounded up to 128 must fit max_prefill_rows. */
int cv2_llm_extend_batch(cv2_llm* h, int32_t n, const int32_t* slots, const int32_t* lens, const int32_t* pos0, const float* embeds, void* stream);
/* n_steps iterations of the decode loop for slots 0..n_seqs-1 in lock step (one hipGraph replay per step);
 * finished slots idle.  No host synchronisation inside.
 * Up to 24 sequences: a step is ONE launch followed by the sampler -- k_step (csrc/chain.h: every layer's Q / attention / O / gate-up /
 * down roles and the head as blocks of one grid in dependency order, activations handed over as epoch-tagged granules) at one row;
 * k_step2 / k_step4 at 2 .. 8 / 9 .. 24 rows (the rows in pairs / fours as columns of the blocks' MFMA operands, one chain of blocks per
 * group; 3 rows: one chain per row).  From 25 rows on, and whenever the environment variable CV2_LLM_CHAIN=0 is set at create, a step is
 * five to seven launches per layer.  CV2_ST_ERR = 3 reports a hand-off that timed out: that step (and every later step of the call)
 * commits nothing -- state, tokens and the KV positions are the ones before it -- so the caller clears the flag and repeats the steps
 * with CV2_DECODE_SHARED (the launches). */
int cv2_llm_decode(cv2_llm* h, int32_t n_seqs, int32_t n_steps, void* stream);
/* The same with flags.  CV2_DECODE_SHARED: other streams' kernels run beside these steps (the streaming scheduler overlaps a decode
 * burst with the previous chunk's flow + HiFT): use the launches even at one row -- k_step's resident polling waves cost the
 * neighbours more than they save here (one stream: 34 -> 40 ms between chunks with k_step beside the chunk work). */
#define CV2_DECODE_SHARED 1
int cv2_llm_decode_ex(cv2_llm* h, int32_t n_seqs, int32_t n_steps, int32_t flags, void* stream);
/* n_steps of the decode loop over the LIVE slots only: row r of every step serves slot slots[r] (host array, n_rows distinct slot
 * ids, copied before the call returns).  A batch of llm.py:649-719 generations ends request by request; the scheduler drops the
 * finished slots from the list at every poll, so a step costs what its live rows cost (the slots' state, caches and results are
 * the ones cv2_llm_decode uses: the two calls may alternate on the same slots). */
int cv2_llm_decode_rows(cv2_llm* h, const int32_t* slots, int32_t n_rows, int32_t n_steps, int32_t flags, void* stream);
/* 1 when decode steps of up to 24 rows of this engine run as one launch (k_step / k_step2 / k_step4), 0 when they run as launches (dims
 * outside k_step's limits, or CV2_LLM_CHAIN=0). */
int cv2_llm_one_launch_step(const cv2_llm* h);
/* test hook: from now on Q-role block `q_block` of layer `layer` of every one-launch step does not publish its results (layer < 0: off; the
 * same (layer, block) key in every one-launch form), so the blocks behind it end in their bounded waits and EVERY row of the block's chain
 * reports CV2_ST_ERR = 3 without committing anything. */
int cv2_llm_debug_skip_publish(cv2_llm* h, int32_t layer, int32_t q_block);
/* test / diagnostic hook: device addresses of the decode workspaces and the granule layout (16 values; tools/dbg_chain_vals.py) */
int cv2_llm_debug_ptrs(cv2_llm* h, uint64_t* out);
/* test hook: ONE sampler launch (k_sample) over slots 0 .. n_rows - 1 from whatever the caller has put into logits[row], state[slot] and
 * out_tokens[slot] -- TransformerLM.sampling_ids + ras_sampling (llm/llm.py:235-250, utils/common.py:111-139) on their own, so that
 * tests/golden/sampler_ras.npz (decisions of the reference's functions under a committed table of uniforms = this sampler's Philox draws)
 * can be replayed on the device.  Updates the slots exactly as a decode step's draw does. */
int cv2_llm_debug_sample(cv2_llm* h, int32_t n_rows, void* stream);

/* Stand-alone skinny GEMM used by the LLM (exported for unit tests and for the flow time-MLP):
 * out[r][n] = sum_k W[n][k] x[r][k] (+bias[n]); rows <= 32; W packed; K % 32 == 0; N % 16 == 0. */
int cv2_skinny_gemm(const uint16_t* w_packed, const float* bias, const float* x, float* out, int32_t rows,
                    int32_t n, int32_t k, void* stream);


/* ------------------------------------------------------------------------------------------------
 * Generic bf16 MFMA GEMM (exported for unit tests): out[m][n] = sum_k A[m][k] W[n][k] (+bias[n]).
 * A row-major bf16 [M][lda] (M % 128 == 0, K % 64 == 0), W packed (above), out fp32 [M][ldo], N % 128 == 0.
 * ---------------------------------------------------------------------------------------------- */
int cv2_gemm_bf16(const uint16_t* a, int64_t lda, const uint16_t* w_packed, const float* bias, float* out, int64_t ldo,
                  int32_t m, int32_t n, int32_t k, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Stage 2 — token -> mel (replaces CausalMaskedDiffWithXvec.inference, cosyvoice/flow/flow.py:235-283, and
 * everything below it: UpsampleConformerEncoder.forward transformer/upsample_encoder.py:243-306,
 * CausalConditionalCFM.forward / solve_euler flow/flow_matching.py:200-225,71-123,
 * CausalConditionalDecoder.forward flow/decoder.py:405-494).
 *
 * Linear / Conv1d weights are bf16 in the packed layout above; a Conv1d weight [C_out][C_in][k] is packed as the
 * [C_out][k*C_in] matrix with column index tap*C_in + c_in (time-major activations make the tap window contiguous).
 * Biases, LayerNorm parameters, embeddings stay fp32.
 * ---------------------------------------------------------------------------------------------- */
typedef struct cv2_flow cv2_flow;

typedef struct { const uint16_t* w; const float* b; } cv2_lin;      /* packed bf16 weight, fp32 bias (may be NULL) */
typedef struct { const float* g; const float* b; } cv2_ln;

typedef struct {            /* ConformerEncoderLayer (transformer/encoder_layer.py:160-236) */
    cv2_ln norm_mha, norm_ff;
    cv2_lin qkv;            /* rows [Wq; Wq; Wk; Wv] (2048 x 512), bias [bq + pos_bias_u; bq + pos_bias_v; bk; bv] */
    cv2_lin pos;            /* linear_pos 512 -> 512, no bias */
    cv2_lin out;            /* linear_out */
    cv2_lin w1, w2;         /* feed_forward 512 -> 2048 -> 512 */
} cv2_conformer;

typedef struct {            /* matcha BasicTransformerBlock (matcha/models/components/transformer.py:243-316) */
    cv2_ln norm1, norm3;
    cv2_lin qkv;            /* rows [to_q; to_k; to_v] (1536 x 256), no bias */
    cv2_lin out;            /* to_out.0 512 -> 256 */
    cv2_lin ff1, ff2;       /* 256 -> 1024 (GELU) -> 256 */
} cv2_tblock;

typedef struct {            /* CausalResnetBlock1D (flow/decoder.py:65-85, matcha decoder.py:46-61) */
    cv2_lin conv1;          /* [256][3*C_in] */
    cv2_ln ln1;
    cv2_lin mlp;            /* Linear 1024 -> 256 applied to Mish(time embedding) */
    cv2_lin conv2;          /* [256][3*256] */
    cv2_ln ln2;
    cv2_lin res;            /* res_conv 1x1: [256][C_in] */
} cv2_resnet;

typedef struct {
    cv2_resnet rn;
    cv2_tblock tb[4];
    cv2_lin tail;           /* down/up CausalConv1d(256,256,3); w == NULL for mid blocks */
} cv2_unet_block;

typedef struct {
    const float* input_embedding;   /* [6561][512] */
    const float* spk_w;             /* spk_embed_affine_layer.weight [80][192] fp32 */
    const float* spk_b;             /* [80] */
    cv2_lin embed; cv2_ln embed_ln;             /* encoder.embed.out.{0,1} */
    cv2_lin pre1, pre2;                         /* pre_lookahead conv1 [512][4*512], conv2 [512][3*512] */
    cv2_conformer enc[6];
    cv2_lin up_conv;                            /* encoder.up_layer.conv [512][5*512] */
    cv2_lin up_embed; cv2_ln up_embed_ln;
    cv2_conformer up[4];
    cv2_ln after_norm;
    cv2_lin enc_proj;                           /* encoder_proj 512 -> 80, rows padded to 128 */
    cv2_lin time1, time2;                       /* time_mlp linear_1 (320 -> 1024), linear_2 (1024 -> 1024) */
    cv2_unet_block down, mid[12], up_blk;
    cv2_lin final_conv; cv2_ln final_ln;        /* final_block */
    cv2_lin final_proj;                         /* 256 -> 80, rows padded to 128 */
    const float* rand_noise;                    /* [15000][80] fp32: flow_matching.py:197-198 noise, time-major */
} cv2_flow_weights;

typedef struct {
    int32_t max_rows;       /* capacity: packed mel-rate rows (sum over sequences of the padded lengths) */
    int32_t max_seqs;       /* sequences per call incl. the CFG twins (2 x utterances) */
    int32_t max_len;        /* longest single sequence in mel frames */
    int32_t n_timesteps;    /* Euler steps (10) */
    float cfg_rate;         /* inference_cfg_rate (0.7) */
} cv2_flow_dims;

size_t cv2_flow_workspace_bytes(const cv2_flow_dims* d);
int cv2_flow_create(const cv2_flow_dims* d, const cv2_flow_weights* w, void* workspace, size_t workspace_bytes,
                    void* stream, cv2_flow** out);
int cv2_flow_destroy(cv2_flow* h);

/* One batch of utterances through flow.inference (flow.py:235-283).  For utterance u (all pointers DEVICE):
 *   tokens[u]       int32 [n_tok[u]]     prompt tokens followed by generated tokens (flow.py:252)
 *   prompt_feat[u]  fp32  [n_prompt_feat[u]][80]   prompt mel, time-major (the reference's prompt_feat[0])
 *   embedding[u]    fp32  [192]
 *   mel_out[u]      fp32  [80][mel_len2[u]] channel-major like the reference's return value,
 *                   mel_len2 = 2 * (n_tok - 3 * !finalize) - n_prompt_feat
 * streaming: chunk-causal attention masks (static chunk 25 tokens / 50 frames); finalize == 0: the last 3 tokens are
 * look-ahead context (flow.py:260-263). */
typedef struct {
    const int32_t* tokens; int32_t n_tok;
    const float* prompt_feat; int32_t n_prompt_feat;
    const float* embedding;
    float* mel_out;
} cv2_flow_utt;
int cv2_flow_inference(cv2_flow* h, const cv2_flow_utt* utts, int32_t n_utts, int32_t streaming, int32_t finalize, void* stream);

/* Streaming calls with a per-stream cache.  The reference's token2wav re-runs flow.inference over the WHOLE prefix for every chunk
 * (cli/model.py:351-381, 300-311) and keeps mel[token_offset * 2:].  With streaming masks the estimator is chunk-causal (flow/decoder.py:
 * 439-441: static chunk of 50 frames; every convolution is causal), so the frames of finished chunks never change: this entry point
 * computes only the frames the cache does not hold yet and returns exactly those the recompute would have produced for them.  The encoder
 * (transformer/upsample_encoder.py:243-306) is chunk-causal in the same way; its per-layer keys / values and convolution tails live in the same
 * cache, so a call's encoder pass covers the new tokens only (CV2_ENC_CACHE=0: recompute the prefix, diagnostics).
 *   cache         device memory of cv2_flow_cache_bytes(h, cache_frames) bytes, ZERO-initialised by the caller before the first call of a
 *                 stream, owned by the caller (one per stream); cache_frames = a multiple of 64 >= the longest prefix in frames
 *   n_cached      frames the cache holds = 2 * (n_tok - 3) of the stream's previous (non-final) call, 0 on the first call; a multiple of 50
 *   gen           0 on the first call, +1 after every SUCCESSFUL call (the convolution tails are double-buffered on it, so a failed
 *                 call can be repeated)
 * utts[u] as for cv2_flow_inference (tokens = the whole prefix), except
 *   mel_out[u]    fp32 [80][n] with n = 2 * (n_tok - 3 * !finalize) - max(n_cached, n_prompt_feat): the frames after the cached ones
 * A non-final call must end on a chunk boundary (2 * (n_tok - 3) % 50 == 0), as the reference's hop alignment guarantees
 * (cli/model.py:357-360). */
typedef struct { void* cache; int32_t cache_frames; int32_t n_cached; int32_t gen; } cv2_flow_cache_ref;
size_t cv2_flow_cache_bytes(const cv2_flow* h, int32_t cache_frames);
int cv2_flow_inference_chunk(cv2_flow* h, const cv2_flow_utt* utts, const cv2_flow_cache_ref* refs, int32_t n_utts, int32_t finalize,
                             void* stream);
/* The first n_frames (whole chunks of 50) of one cache and its convolution tails into another cache (other capacity allowed): a stream
 * whose prompt (tokens, mel, speaker embedding) another call has already run starts from that call's cache — n_cached = n_frames and the
 * same `gen` as the source.  The source must hold exactly n_frames (its tails belong to that position). */
int cv2_flow_cache_copy(const cv2_flow* h, const void* src, int32_t src_frames, void* dst, int32_t dst_frames, int32_t n_frames, void* stream);

/* Test hook: 1 / 0 = the estimator attention of large batches (>= 4096 rows) with / without LDS DMA staging of its key / value tiles
 * (the DMA form permutes a tile's keys among the score rows: the same products summed in another order inside a matrix-core k group --
 * the two forms agree to fp32 round-off, 3-4e-3 of the mel range after the 10 Euler steps, NOT bit for bit; batches of >= 4096 rows
 * therefore differ from the same utterances run alone by that margin), -1 = the default. */
int cv2_flow_debug_attn_dma(int32_t on);
/* Test hook: 1 = cv2_flow_inference captures the launches of a call into a hipGraph at the second use of a shape (token / prompt lengths, flags)
 * and replays it from then on (bit-identical results; CV2_FLOW_GRAPH=1), 0 = always the launches, -1 = the environment (default off: measured
 * 0.4 ms slower per configs[1] utterance, profiles/r6_flow_graph_ab.txt). */
int cv2_flow_debug_graph(int32_t on);

/* The estimator alone behind the reference's TensorRT seam (flow_matching.py:125-150): six contiguous device
 * tensors x(2,80,T) mask(2,1,T) mu(2,80,T) t(2,) spks(2,80) cond(2,80,T), result written in place into x. */
int cv2_flow_estimator(cv2_flow* h, float* x, const float* mask, const float* mu, const float* t, const float* spks,
                       const float* cond, int32_t T, int32_t streaming, void* stream);
/* The encoder alone (UpsampleConformerEncoder.forward): xs fp32 [T][512] embedded tokens (+ optional context [3][512]),
 * out fp32 [2T][512]. */
int cv2_flow_encoder(cv2_flow* h, const float* xs, int32_t T, const float* context, int32_t streaming, float* out, void* stream);


/* ------------------------------------------------------------------------------------------------
 * Stage 3 — HiFT vocoder (replaces HiFTGenerator.inference, cosyvoice/hifigan/generator.py:570-582, and decode :520-552,
 * _stft/_istft :504-518, ResBlock :94-101, SourceModuleHnNSF2 :375-389, SineGen2 :256-339, ConvRNNF0Predictor
 * f0_predictor.py:55-58).  fp32 storage and accumulation throughout.
 *
 * Conv weights (weight-norm already folded, w = g v / ||v||) are fp32 in the MFMA 32x32x2 B-operand order
 * [tap][c_in/2][c_out/32][64 lanes], lane = (c_in & 1) * 32 + (c_out & 31), both channel counts zero-padded to 64.
 * A ConvTranspose1d(stride u, kernel k, padding p) is stored as its polyphase Conv1d with u*C_out output channels
 * (index r*C_out + c_out) over 3 taps: W'[r*C_out + c_out][c_in][j] = W[c_in][c_out][r + p + u (1 - j)] where that
 * index lies in [0, k), else 0.
 * ---------------------------------------------------------------------------------------------- */
typedef struct cv2_hift cv2_hift;
typedef struct {
    const float* w; const float* b;     /* packed weight, bias [cout_pad] */
    int32_t cin, cout, cin_pad, cout_pad, taps, dil, pad_left;
    const uint16_t* w3;                 /* optional: the same weight as three bf16 planes (w = w0 + w1 + w2, each the bf16 rounding of the
                                           remainder) in the MFMA 32x32x16 B-operand order [tap][c_in/16][c_out/32][plane][64 lanes][8],
                                           lane = ((c_in % 16) / 8) * 32 + (c_out & 31), element = c_in % 8.  When given, the convolution runs
                                           on the bf16 matrix cores with six products per term (fp32-equivalent accuracy, csrc/hift.hip k_conv6);
                                           NULL: fp32 matrix cores (exact fp32 FMA chains; kept for the f0 predictor, whose output is integrated
                                           into a 1e5-rad phase) */
} cv2_conv;
typedef struct {                          /* ResBlock: convs1 (dilated), convs2, Snake alphas */
    cv2_conv c1[3], c2[3];
    const float* a1[3]; const float* a2[3];
} cv2_resblock;
typedef struct {
    cv2_conv f0_conv[5];                  /* f0_predictor.condnet.{0,2,4,6,8} */
    const float* f0_w; const float* f0_b; /* f0_predictor.classifier [512], [1] */
    const float* src_w; const float* src_b; /* m_source.l_linear [9], [1] */
    cv2_conv conv_pre;
    cv2_conv ups[3];
    const float* sd_w[3]; const float* sd_b[3];  /* source_downs weights re-ordered [k][18][C] (the scalar kernel; used when sd_conv[i].w3 is NULL) */
    cv2_conv sd_conv[3];                  /* source_downs.{0,1,2} (generator.py:468-479) as 1-tap convolutions over flat windows of the [F][18] STFT rows:
                                             cin = 18 k with input index j * 18 + c for tap j, channel c; taps = 1; w3 given -> they run on the matrix cores */
    cv2_resblock src_rb[3];
    cv2_resblock rb[9];
    cv2_conv conv_post;
} cv2_hift_weights;
typedef struct { int32_t max_frames; int32_t lanes; } cv2_hift_dims;   /* longest mel in frames; lanes > 1: workspace copies for cv2_hift_inference_batch */

size_t cv2_hift_workspace_bytes(const cv2_hift_dims* d);
int cv2_hift_create(const cv2_hift_dims* d, const cv2_hift_weights* w, void* workspace, size_t workspace_bytes, cv2_hift** out);
int cv2_hift_destroy(cv2_hift* h);
/* Test hook: pair / xcd_split = 1 / 0 runs the vocoder with / without the fused ResBlock pairs of the 64- and 128-channel stages (one launch
 * per (dilated conv, conv) pair, the intermediate kept in LDS) and with / without the per-layer split of the convolution grids over the XCDs;
 * -1 = the default.  All four combinations produce the same waveform bit for bit.  Process-wide; calls of <= 160 frames replay graphs
 * captured under the mode of their first call. */
int cv2_hift_debug_modes(int32_t pair, int32_t xcd_split);
/* Test hook: how the convolutions with plane weights multiply.  2 = two bf16 planes per operand, three products (~2^-15 of a term; the
 * default since round 6), 0 = three planes, six products (fp32-equivalent; CV2_HIFT_PLANES=3), 1 = the fp32 matrix-core kernel everywhere
 * (CV2_HIFT_FP32=1), -1 = what the environment says.  Process-wide; graphs as for cv2_hift_debug_modes. */
int cv2_hift_debug_precision(int32_t mode);
/* Test hook: the f0 track [T] (Hz; ConvRNNF0Predictor.forward, f0_predictor.py:55-58) of the engine's last cv2_hift_inference call. */
int cv2_hift_debug_f0(cv2_hift* h, float* out, int32_t T, void* stream);
/* mel fp32 [80][T] (channel-major, the reference's speech_feat[0]); cache_source fp32 [n_cache] or NULL;
 * noise: fp32 [480 T][9] standard normals injected in place of the reference's randn_like (generator.py:334), or NULL
 * to draw them on the device from Philox(seed); wav fp32 [480 T]; source fp32 [480 T]. */
int cv2_hift_inference(cv2_hift* h, const float* mel, int32_t T, const float* cache_source, int32_t n_cache,
                       const float* noise, uint64_t seed, float* wav, float* source, void* stream);
/* n <= lanes chunks of ONE shape (T <= 160 frames, n_cache cached source samples each) as one set of launches (gridDim.z = n): the
 * chunks of a streaming round (cli/model.py:351-381 x the concurrent streams).  Arrays of n device pointers / seeds on the host; the
 * results equal n cv2_hift_inference calls with the same seeds. */
int cv2_hift_inference_batch(cv2_hift* h, int32_t n, const float* const* mel, int32_t T, const float* const* cache_source, int32_t n_cache,
                             const uint64_t* seeds, float* const* wav, float* const* source, void* stream);

/* fade_in_out (cosyvoice/utils/common.py:142-150): new[:w] = new[:w] * win[:w] + old[-w:] * win[w:], in place on `fade_in`;
 * win fp32 [2w] (np.hamming(2w)), old_tail points at the last w samples of the previous chunk. */
int cv2_fade_in_out(float* fade_in, const float* old_tail, const float* window, int32_t w, void* stream);
/* cli/model.py:328-330 (speed != 1, non-streaming): out[r][j] = linear interpolation of in[r][.] at (j + 0.5) n_in / n_out - 0.5
 * (torch.nn.functional.interpolate(mode='linear'), align_corners False); in [rows][n_in], out [rows][n_out], fp32 device. */
int cv2_interp_linear(const float* in, float* out, int32_t rows, int32_t n_in, int32_t n_out, void* stream);

/* =====================================================================================================================
 * Prompt features (the extractor that feeds this path): replaces the reference's
 *   feat_extractor = matcha.utils.audio.mel_spectrogram(n_fft 1920, num_mels 80, sampling_rate 24000, hop_size 480, win_size 1920,
 *                    fmin 0, fmax 8000, center False)          conf/cosyvoice2.yaml:152-160, third_party/Matcha-TTS/matcha/utils/audio.py:45-82
 *   torchaudio.transforms.Resample(16000, 24000)               cli/frontend.py:497 (torchaudio 2.3.1, functional.py _apply_sinc_resample_kernel)
 * Table-driven: the host computes window, twiddles, mel filterbank and the polyphase kernel (cv2amd/prompt.py), all DEVICE pointers.
 * cv2_melspec: wav fp32 [n]; out fp32 [n_frames][n_mels] time-major (= prompt_speech_feat[0]); reflect padding (n_fft - hop) / 2 on both
 *   sides, frames of n_fft at stride hop, |DFT| = sqrt(re^2 + im^2 + 1e-9), out = log(max(mel_fb @ |DFT|, clamp_min));
 *   n_frames = 1 + (n + (n_fft - hop) - n_fft) / hop.
 * cv2_resample: out[i * up + p] = sum_j kernel[p][j] * xpad[i * down + j], xpad = zeros(pad_left) ++ in ++ zeros; n_out <= ceil(n_in * up / down). */
typedef struct {
    int32_t n_fft, hop, n_mels, n_bins;          /* n_bins = n_fft / 2 + 1 */
    const double* window;                        /* [n_fft] fp64 */
    const double* twiddle;                       /* [n_fft][2] fp64: cos, sin of 2 pi j / n_fft */
    const float* mel_fb;                         /* [n_mels][n_bins] */
    const int32_t* fb_lo; const int32_t* fb_hi;  /* [n_mels]: filter m is zero outside bins [fb_lo[m], fb_hi[m]) */
    float clamp_min;
} cv2_melspec_cfg;
int cv2_melspec(const cv2_melspec_cfg* cfg, const float* wav, int64_t n, float* out, int32_t n_frames, void* stream);
int cv2_resample(const float* in, int64_t n_in, const float* kernel, int32_t up, int32_t down, int32_t klen, int32_t pad_left,
                 float* out, int64_t n_out, void* stream);

/* The two feature extractors in front of the ONNX prompt models (cosyvoice/cli/frontend.py:262-283; third-party algorithms):
 *   whisper.log_mel_spectrogram(speech, n_mels=128) -> speech_tokenizer_v2.onnx   (frontend.py:264): center = 1, win = n_fft = 400, hop 160,
 *       periodic Hann window, |X|^2 of every frame but the last, Slaney filters [128][201], log10(max(., 1e-10));
 *       cv2_whisper_post: max(x, max(x) - 8), (x + 4) / 4, transposed to [n_mels][frames] (the layout the ONNX graph takes)
 *   torchaudio.compliance.kaldi.fbank(speech, num_mel_bins=80, dither=0, sample_frequency=16000) -> campplus.onnx (frontend.py:277):
 *       center = 0 (snip_edges), win 400, n_fft 512 (zero padded), hop 160, remove_dc = 1, preemph 0.97, povey window, |X|^2, kaldi mel
 *       banks [80][257] (20 Hz .. Nyquist), ln(max(., FLT_EPSILON)); cv2_sub_col_mean: feat - feat.mean(dim=0) (frontend.py:278)
 * cv2_framefeat: wav fp32 [n]; out fp32 [n_frames][n_mels]; n_frames = n / hop (center) or 1 + (n - win) / hop (snip_edges).
 * Tables are DEVICE pointers computed by the host (cv2amd/prompt.py). */
typedef struct {
    int32_t n_fft, win, hop, n_bins, n_mels;     /* win <= n_fft: the windowed frame is zero-padded to the DFT length */
    const double* window;                        /* [win] fp64 */
    const double* twiddle;                       /* [n_fft][2] fp64: cos, sin of 2 pi j / n_fft */
    const float* mel_fb;                         /* [n_mels][n_bins] */
    const int32_t* fb_lo; const int32_t* fb_hi;  /* [n_mels]: filter m is zero outside bins [fb_lo[m], fb_hi[m]) */
    int32_t center, remove_dc;
    float preemph;
    int32_t log10;                               /* 1: log10, 0: natural log */
    float floor;                                 /* clamp before the log */
} cv2_framefeat_cfg;
int cv2_framefeat(const cv2_framefeat_cfg* cfg, const float* wav, int64_t n, float* out, int32_t n_frames, void* stream);
int cv2_whisper_post(const float* in, float* out, int32_t n_frames, int32_t n_mels, void* stream);
int cv2_sub_col_mean(float* x, int32_t n_rows, int32_t n_cols, void* stream);

/* Test hooks: the element-wise functions exactly as the kernels evaluate them (fast-math exp / sin / reciprocal), so that their
 * error against libm can be swept over the input ranges of the real checkpoint (tests/test_math_gpu.py).
 *   act: 1 exact-erf GELU (diffusers GELU, matcha transformer.py FeedForward), 2 SiLU, 3 Mish, 4 leaky ReLU (slope)
 *   pre: 1 Snake x + sin^2(alpha x) / (alpha + 1e-9) (hifigan/generator.py ResBlock activations), 2 leaky ReLU (slope) */
int cv2_dbg_act(const float* x, float* out, int64_t n, int32_t act, float slope, void* stream);
int cv2_dbg_pre(const float* x, float* out, int64_t n, int32_t pre, float alpha, float slope, void* stream);

#ifdef __cplusplus
}
#endif
#endif
