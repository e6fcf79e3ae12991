// Stage 2 on MI355X: speech tokens -> mel (replaces cosyvoice/flow/flow.py:235-283 and everything under it).
//
// Data layout in HBM ("packed ragged rows"): every activation is TIME-MAJOR, one row per frame, channels contiguous.
// The sequences of a call (utterances; for the estimator also their classifier-free-guidance twins) are packed
// one after another; sequence s owns rows [start_s, start_s + len_s), start_s is a multiple of 128 and at least 8
// zero rows follow every sequence.  Buffers that feed convolutions carry 128 zero guard rows in front.  With this
// layout a causal Conv1d(k) is a GEMM whose A operand is the same buffer read at row offset -(k-1) with
// K = k*C (the tap window of a row is contiguous memory), and the zero rows are the causal padding.  Every
// epilogue re-zeroes the rows beyond a sequence (the reference's `* mask`), so the invariant holds end to end.
//
// Kernels (roofline that bounds each):
//   k_gemm (gemm.h)        MFMA bf16   every Linear / Conv1d of the encoder and the estimator, fused epilogues
//   k_attn_est             MFMA bf16   estimator self-attention, flash style (no T x T matrix in HBM); below 4 096 rows four key groups per block
//   k_attn_est_dma         MFMA bf16   the same from 4 096 rows on, key / value tiles by LDS DMA (three stages, three waves per SIMD)
//   k_relsoftmax           HBM         conformer rel-pos softmax over explicit score matrices (encoder only, ~2 % of FLOPs)
//   k_layernorm, k_pack, k_euler, ...  HBM   small row-wise kernels
#include "gemm_launch.h"
#include "skinny_launch.h"
#include "../../include/cv2_amd.h"
#include <algorithm>
#include <map>
#include <math.h>
#include <atomic>
#include <vector>

#define GUARD 128

extern "C" int cv2_gemm_bf16(const uint16_t* a, int64_t lda, const uint16_t* w, const float* bias, float* out, int64_t ldo,
                             int32_t m, int32_t n, int32_t k, void* stream) {
    CV2_CHECK(a && w && out, "cv2_gemm_bf16: null argument");
    GemmArgs g = gemm_args(a, lda, 0, w, m, n, k);
    g.bias = bias; g.out_f32 = out; g.ldo = ldo;
    return gemm_launch_cfg(g, n % 256 == 0 && n <= 256 ? 1 : 0, 1, true, (hipStream_t)stream);
}

// =========================================================================== small kernels
struct LnArgs {
    const float* x; long ldx; int C;
    const float* g; const float* b; float eps; float scale;
    SeqTable seq; int rows;
    float* out_f32; long ldo; uint16_t* out_bf16; long ldo16;
};
// one wave per row; C in {256, 512}
template <int PER>
__global__ __launch_bounds__(256) void k_layernorm(LnArgs a) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= a.rows) return;
    const int s = a.seq.tile_seq[row >> 6];
    const bool valid = s >= 0 && row - a.seq.seq_start[s] < a.seq.seq_len[s];
    constexpr int per = PER;                        // 4 or 8 consecutive channels per lane
    float v[PER];
    const float* xr = a.x + (size_t)row * a.ldx + lane * per;
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < per; i += 4) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(xr + i);
        v[i] = t[0]; v[i + 1] = t[1]; v[i + 2] = t[2]; v[i + 3] = t[3];
        sum += (t[0] + t[1]) + (t[2] + t[3]);
    }
    const float mean = wave_sum(sum) / a.C;
    float sq = 0.f;
    for (int i = 0; i < per; i++) { v[i] -= mean; sq += v[i] * v[i]; }
    const float rstd = rsqrtf(wave_sum(sq) / a.C + a.eps);
    for (int i = 0; i < per; i++) {
        const int c = lane * per + i;
        v[i] = valid ? (v[i] * rstd * a.g[c] + a.b[c]) * a.scale : 0.f;
    }
    if (a.out_f32) for (int i = 0; i < per; i += 4)
        *reinterpret_cast<f32x4*>(a.out_f32 + (size_t)row * a.ldo + lane * per + i) = (f32x4){v[i], v[i + 1], v[i + 2], v[i + 3]};
    if (a.out_bf16) for (int i = 0; i < per; i += 4)
        *reinterpret_cast<uint2*>(a.out_bf16 + (size_t)row * a.ldo16 + lane * per + i) = make_uint2(pack_bf16x2(v[i], v[i + 1]), pack_bf16x2(v[i + 2], v[i + 3]));
}

// token ids -> embedding rows (flow.py:252-256): bf16 [rows][512]; ids clamped at 0; rows beyond the sequence -> 0
struct EmbedPtrArgs { const int* const* tokens; const float* table; SeqTable seq; int rows; uint16_t* out; };
__global__ __launch_bounds__(256) void k_embed_tokens_ptr(EmbedPtrArgs a) {
    const int row = blockIdx.x * 2 + (threadIdx.x >> 7), c = (threadIdx.x & 127) * 4;
    if (row >= a.rows) return;
    const int s = a.seq.tile_seq[row >> 6];
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (s >= 0) {
        const int t = row - a.seq.seq_start[s];
        if (t < a.seq.seq_len[s]) {
            int id = a.tokens[s][t];
            id = id < 0 ? 0 : id;
            v = *reinterpret_cast<const f32x4*>(a.table + (size_t)id * 512 + c);
        }
    }
    *reinterpret_cast<uint2*>(a.out + (size_t)row * 512 + c) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
}
// 3 look-ahead context rows (fp32 [3][512]) -> bf16 rows
__global__ __launch_bounds__(256) void k_ctx_rows(const float* ctx, uint16_t* out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < 3 * 512) out[i] = f2bf(ctx[i]);
}

// fp32 [rows][512] -> bf16 (plain cast with mask), used for externally supplied encoder inputs
struct CastArgs { const float* x; uint16_t* out; SeqTable seq; int rows; int C; const int* src_row_off; };
__global__ __launch_bounds__(256) void k_cast_rows(CastArgs a) {
    const int per_row = a.C / 4;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int row = idx / per_row, c = (idx % per_row) * 4;
    if (row >= a.rows) return;
    const int s = a.seq.tile_seq[row >> 6];
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (s >= 0) {
        const int t = row - a.seq.seq_start[s];
        if (t < a.seq.seq_len[s]) v = *reinterpret_cast<const f32x4*>(a.x + ((size_t)a.src_row_off[s] + t) * a.C + c);
    }
    *reinterpret_cast<uint2*>(a.out + (size_t)row * a.C + c) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
}

// Upsample1D nearest x2 (upsample_encoder.py:55-57): token-rate fp32 rows -> mel-rate bf16 rows of the new layout
struct RepeatArgs { const float* x; SeqTable seq_in; SeqTable seq_out; int rows_out; uint16_t* out; };
__global__ __launch_bounds__(256) void k_repeat2(RepeatArgs a) {
    const int row = blockIdx.x * 2 + (threadIdx.x >> 7), c = (threadIdx.x & 127) * 4;
    if (row >= a.rows_out) return;
    const int s = a.seq_out.tile_seq[row >> 6];
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (s >= 0) {
        const int t = row - a.seq_out.seq_start[s];
        if (t < a.seq_out.seq_len[s]) v = *reinterpret_cast<const f32x4*>(a.x + ((size_t)a.seq_in.seq_start[s] + (t >> 1)) * 512 + c);
    }
    *reinterpret_cast<uint2*>(a.out + (size_t)row * 512 + c) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
}

// EspnetRelPositionalEncoding (embedding.py:226-302): rows r = 0 .. 2T-2 hold position T-1-r; even cols sin, odd cos.
__global__ __launch_bounds__(256) void k_pos_emb(uint16_t* out, int T, int rows_pad) {
    const int r = blockIdx.x, i2 = threadIdx.x;              // 256 (sin, cos) pairs
    float sv = 0.f, cv = 0.f;
    if (r < 2 * T - 1) {
        const float pos = (float)(T - 1 - r);
        const float div = expf((float)(2 * i2) * (-(logf(10000.0f) / 512.f)));
        sv = sinf(pos * div); cv = cosf(pos * div);
    }
    if (r < rows_pad) *reinterpret_cast<uint32_t*>(out + (size_t)r * 512 + 2 * i2) = pack_bf16x2(sv, cv);
}

// softmax over (ac[i][j] + bd[i][T-1-ig+j]) / 8 with the chunk mask (attention.py:225-247 rel_shift, :297-330); the launch's query rows
// i < n_new are positions ig = n0 + i of a sequence of T keys (n0 = 0, n_new = T: the whole sequence; cached streaming: the new rows)
struct RelSmArgs { const float* ac; const float* bd; uint16_t* probs; int T, Mp, Tk, P, chunk, n0, n_new; };
__global__ __launch_bounds__(256) void k_relsoftmax(RelSmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sc = reinterpret_cast<float*>(smem);
    __shared__ float red[8];
    const int i = blockIdx.x, h = blockIdx.y, tid = threadIdx.x;
    uint16_t* pr = a.probs + ((size_t)h * a.Mp + i) * a.Tk;
    if (i >= a.n_new) { for (int j = tid; j < a.Tk; j += 256) pr[j] = 0; return; }
    const int ig = a.n0 + i;
    const int kmax = a.chunk > 0 ? min(a.T, (ig / a.chunk + 1) * a.chunk) : a.T;
    const float* acr = a.ac + ((size_t)h * a.Mp + i) * a.Tk;
    const float* bdr = a.bd + ((size_t)h * a.Mp + i) * a.P + (a.T - 1 - ig);
    float mx = -INFINITY;
    for (int j = tid; j < kmax; j += 256) { const float v = (acr[j] + bdr[j]) * 0.125f; sc[j] = v; mx = fmaxf(mx, v); }
    mx = wave_max(mx);
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int j = tid; j < kmax; j += 256) { const float p = __expf(sc[j] - mx); sc[j] = p; sum += p; }
    sum = wave_sum(sum);
    if ((tid & 63) == 0) red[4 + (tid >> 6)] = sum;
    __syncthreads();
    const float inv = 1.f / (red[4] + red[5] + red[6] + red[7]);
    for (int j = tid; j < a.Tk; j += 256) pr[j] = j < kmax ? f2bf(sc[j] * inv) : (uint16_t)0;
}

// F.normalize(embedding) -> Linear 192 -> 80 (flow.py:248-249), fp32; one block per utterance
struct SpkArgs { const float* const* emb; const float* w; const float* b; float* out; };
__global__ __launch_bounds__(128) void k_spk(SpkArgs a) {
    __shared__ float e[192];
    __shared__ float red[2];
    const float* x = a.emb[blockIdx.x];
    float sq = 0.f;
    for (int i = threadIdx.x; i < 192; i += 128) { e[i] = x[i]; sq += x[i] * x[i]; }
    sq = wave_sum(sq);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sq;
    __syncthreads();
    const float inv = 1.f / fmaxf(sqrtf(red[0] + red[1]), 1e-12f);
    if (threadIdx.x < 80) {
        float acc = 0.f;
        for (int i = 0; i < 192; i++) acc += a.w[threadIdx.x * 192 + i] * (e[i] * inv);
        a.out[blockIdx.x * 80 + threadIdx.x] = acc + a.b[threadIdx.x];
    }
}

// ---- estimator input pack / Euler update (flow_matching.py:94-121) --------------------------------------------
// Estimator rows: cond twins occupy rows [0, RU) in the encoder's mel-rate layout, uncond twins rows [RU, 2RU).
struct PackArgs {
    float* x;                    // [RU][80] state
    const float* mu;             // [RU][80]
    const float* spk;            // [U][80]
    const float* const* prompt;  // [U] -> fp32 [n_prompt][80]
    const int* n_prompt;         // [U]
    const float* noise;          // [15000][80]; != null on the first step: x = noise[t]
    const float* v;              // [2RU][80] estimator output of the previous step (null on the first)
    float dt, cfg;
    SeqTable seq;                // estimator table (2U sequences)
    int RU, U;
    uint16_t* a0;                // [2RU][320]
    // cached streaming: the rows are the NEW frames of every stream; frame t of sequence s sits at position pos0[s] + t of its
    // utterance (noise, prompt mel) and its mu row is mu_start[s] + pos0[s] + t.  twin = row distance to the uncond twin; rows
    // below `lead` belong to no sequence (their twin rows do not exist).
    const int* pos0; const int* mu_start; int twin, lead;
};
__global__ __launch_bounds__(256) void k_euler_pack(PackArgs a) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int row = idx / 20, c = (idx % 20) * 4;            // 80 channels = 20 float4
    if (row >= a.RU) return;
    const int s = a.seq.tile_seq[row >> 6];
    f32x4 x = {0.f, 0.f, 0.f, 0.f}, mu = x, sp = x, cd = x;
    bool valid = false;
    if (s >= 0) {
        const int t = row - a.seq.seq_start[s];
        valid = t < a.seq.seq_len[s];
        if (valid) {
            const int pt = (a.pos0 ? a.pos0[s] : 0) + t;               // frame index within the utterance
            if (a.noise) x = *reinterpret_cast<const f32x4*>(a.noise + (size_t)pt * 80 + c);
            else {
                x = *reinterpret_cast<const f32x4*>(a.x + (size_t)row * 80 + c);
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(a.v + (size_t)row * 80 + c);
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(a.v + (size_t)(row + a.twin) * 80 + c);
                x = x + a.dt * ((1.0f + a.cfg) * v0 - a.cfg * v1);
            }
            const size_t mrow = a.mu_start ? (size_t)((long)a.mu_start[s] + pt) : (size_t)row;      // (mu_start may be negative: first mu row minus the cached frames)
            mu = *reinterpret_cast<const f32x4*>(a.mu + mrow * 80 + c);
            sp = *reinterpret_cast<const f32x4*>(a.spk + (size_t)s * 80 + c);
            if (pt < a.n_prompt[s]) cd = *reinterpret_cast<const f32x4*>(a.prompt[s] + (size_t)pt * 80 + c);
        }
    }
    *reinterpret_cast<f32x4*>(a.x + (size_t)row * 80 + c) = x;
    const uint2 xb = make_uint2(pack_bf16x2(x[0], x[1]), pack_bf16x2(x[2], x[3]));
    uint16_t* r0 = a.a0 + (size_t)row * 320 + c;
    uint16_t* r1 = a.a0 + (size_t)(row + a.twin) * 320 + c;
    if (row < a.lead) {                                              // lead rows: no twin
        *reinterpret_cast<uint2*>(r0) = make_uint2(0u, 0u); *reinterpret_cast<uint2*>(r0 + 80) = make_uint2(0u, 0u);
        *reinterpret_cast<uint2*>(r0 + 160) = make_uint2(0u, 0u); *reinterpret_cast<uint2*>(r0 + 240) = make_uint2(0u, 0u);
        return;
    }
    *reinterpret_cast<uint2*>(r0) = xb;
    *reinterpret_cast<uint2*>(r0 + 80) = make_uint2(pack_bf16x2(mu[0], mu[1]), pack_bf16x2(mu[2], mu[3]));
    *reinterpret_cast<uint2*>(r0 + 160) = make_uint2(pack_bf16x2(sp[0], sp[1]), pack_bf16x2(sp[2], sp[3]));
    *reinterpret_cast<uint2*>(r0 + 240) = make_uint2(pack_bf16x2(cd[0], cd[1]), pack_bf16x2(cd[2], cd[3]));
    *reinterpret_cast<uint2*>(r1) = xb;
    *reinterpret_cast<uint2*>(r1 + 80) = make_uint2(0u, 0u);
    *reinterpret_cast<uint2*>(r1 + 160) = make_uint2(0u, 0u);
    *reinterpret_cast<uint2*>(r1 + 240) = make_uint2(0u, 0u);
}

// final state -> mel_out[u][80][mel_len2] (flow.py:281): drop the prompt frames, channel-major
struct MelOutArgs { const float* x; float* const* out; const int* n_prompt; SeqTable seq; };
__global__ __launch_bounds__(256) void k_mel_out(MelOutArgs a) {
    const int s = blockIdx.y;
    const int len = a.seq.seq_len[s], p = a.n_prompt[s], start = a.seq.seq_start[s];
    const int n2 = len - p;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)n2 * 80) return;
    const int c = idx / n2, t = idx % n2;
    a.out[s][(size_t)c * n2 + t] = a.x[(size_t)(start + p + t) * 80 + c];
}

// ---- cached streaming (cv2_flow_inference_chunk) ------------------------------------------------------------------------------
// The estimator of a streaming call is chunk-causal (decoder.py:439-441 masks, causal convolutions), so the frames of finished
// chunks never change and the reference's recompute of the whole prefix per chunk (cli/model.py:351-381) can be replaced by a
// per-stream cache: for every (CFG branch, Euler step, transformer block) the keys / values of all frames so far, and for every
// causal convolution the last two input rows.
#define INC_LEAD 128            // rows in front of the first sequence: its conv tails are written there (never into the zero guard)
#define INC_TBLOCKS 56
#define INC_CONVS 31
struct IncTabs {
    uint16_t* const* kv;        // [2U] cache base of (stream, branch): [n_steps][56] slots of K [frames][512] + V^T [512][frames]
    const int* kv_frames;       // [2U]
    const int* pos0;            // [2U] frames cached before this call
    uint16_t* const* tails;     // [2U] conv tails of (stream, branch): [2 generations][n_steps][31][2 rows][512]
    const int* gen;             // [2U] generation to READ (the call writes the other one: a failed call can be repeated)
};
// (the call's new keys / values reach the cache slot through the QKV projection's epilogue: GemmArgs.kvc, gemm.h)
// before a causal k=3 convolution reads `buf`: rows [start-2, start) <- the cached last two input rows of the previous call,
// and the last two input rows of this call -> cache (other generation)
struct ConvTailArgs { uint16_t* buf; int C; SeqTable seq; IncTabs inc; int rd_slot, wr_slot; };     // slot = (gen * n_steps + step) * 31 + conv
__global__ __launch_bounds__(256) void k_conv_tail(ConvTailArgs a) {
    const int s = blockIdx.x;
    const int start = a.seq.seq_start[s], len = a.seq.seq_len[s], p0 = a.inc.pos0[s], g = a.inc.gen[s] & 1;
    const uint16_t* tr = a.inc.tails[s] + (size_t)(g ? a.wr_slot : a.rd_slot) * 1024;       // slots are given for gen = 0; swapped for gen = 1
    uint16_t* tw = a.inc.tails[s] + (size_t)(g ? a.rd_slot : a.wr_slot) * 1024;
    for (int e = threadIdx.x; e < 2 * a.C; e += 256) {
        const int j = e / a.C, c = e - j * a.C;
        const uint16_t o0 = p0 > 0 ? tr[c] : (uint16_t)0, o1 = p0 > 0 ? tr[512 + c] : (uint16_t)0;
        a.buf[(long)(start - 2 + j) * a.C + c] = j ? o1 : o0;
        const int pos = len - 2 + j;
        tw[j * 512 + c] = pos >= 0 ? a.buf[(long)(start + pos) * a.C + c] : (pos == -1 ? o1 : o0);
    }
}

// TensorRT-seam pack / unpack (flow_matching.py:125-150): channel-major (2,80,T) tensors <-> packed rows
struct SeamPackArgs { const float* x; const float* mu; const float* spks; const float* cond; int T; int row1; uint16_t* a0; int rows; };
__global__ __launch_bounds__(256) void k_seam_pack(SeamPackArgs a) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int row = idx / 320, c = idx % 320;
    if (row >= a.rows) return;
    const int b = row >= a.row1 ? 1 : 0, t = row - b * a.row1;
    float v = 0.f;
    if (t < a.T) {
        const int g = c / 80, ch = c % 80;
        const size_t o = ((size_t)b * 80 + ch) * a.T + t;
        v = g == 0 ? a.x[o] : g == 1 ? a.mu[o] : g == 2 ? a.spks[b * 80 + ch] : a.cond[o];
    }
    a.a0[(size_t)row * 320 + c] = f2bf(v);
}
__global__ __launch_bounds__(256) void k_seam_unpack(const float* v, float* x, int T, int row1) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= 2L * 80 * T) return;
    const int t = idx % T, ch = (idx / T) % 80, b = idx / (80L * T);
    x[idx] = v[(size_t)(b * row1 + t) * 80 + ch];
}

// sinusoidal time embedding (matcha decoder.py:14-29): [n][320] fp32, scale 1000
__global__ void k_sinus(const float* t, float* out, int n) {
    const int i = blockIdx.x, j = threadIdx.x;          // 160 threads
    if (i >= n || j >= 160) return;
    const float e = expf((float)j * -(logf(10000.0f) / 159.f));
    const float arg = 1000.f * t[i] * e;
    out[i * 320 + j] = sinf(arg);
    out[i * 320 + 160 + j] = cosf(arg);
}
__global__ void k_act_inplace(float* x, int n, int act) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) x[i] = act_apply(x[i], act, 0.f);
}

// =========================================================================== estimator attention (flash style)
// grid (q tiles of 64 rows, 8 heads); 4 waves x 16 query rows.  S^T = K Q^T so a lane owns one query column and
// 4 keys per 16-key tile; P feeds the PV MFMA from the same registers (the k-slot order of the second product is
// chosen to match), V^T comes pre-transposed from the QKV GEMM epilogue.
struct AttnEstArgs {
    const uint16_t* qk;   // [R][1024]: q cols [0,512), k cols [512,1024)
    const uint16_t* vt;   // [512][R]
    uint16_t* out;        // [R][512]
    SeqTable seq; int chunk; long R;
    // cached streaming (cv2_flow_inference_chunk): keys / values of sequence s come from its cache slot, which already holds the
    // rows of this call; the queries are the call's new frames at positions pos0[s] + t
    uint16_t* const* kv;  // [S] -> cache base of the sequence's CFG branch (null: keys from qk / vt)
    const int* kv_frames; // [S] capacity in frames of that cache
    const int* pos0;      // [S] frames already cached before this call
    long slot;            // (Euler step * 56 + transformer block): slot s of a cache = K [frames][512] then V^T [512][frames]
};
#define AK_LD 72    // K tile row stride (bf16 elements): 64 + 8
#define AV_LD 68    // V^T tile row stride: 64 + 4
// one 64-key tile of the flash loop in three phases: scores, running softmax, O^T update (Kt / Vt = the staged tile in LDS)
// SWZ: the tile was written by LDS DMA (k_attn_est_dma): rows of 128 B without padding; chunk c (16 B) of V^T row d sits in slot c ^ (d & 7),
// chunk c of K row r in slot c ^ att_kswz-of-its-score-row.  The score rows are a permutation of the tile's keys, chosen so that the eight
// P entries a lane holds per key-pair tile are eight consecutive keys: the V^T operand of the second product is then ONE 16-byte read
// (the register-staged form reads two 8-byte halves 16 keys apart and moves them together).  Same products, another order inside a
// matrix-core k group: the two forms agree to fp32 round-off, not bit for bit.
__device__ __forceinline__ int att_kswz(int i) { return (i >> 2) * 2 + ((i >> 1) & 1); }      // i = score row (0 .. 15): two rows per value
// S^T: 4 key tiles x (d = 64 in two k-steps), K fragments shared by the QS query sub-tiles
template <int QS, bool SWZ>
__device__ __forceinline__ void att_qk(const uint16_t* Kt, const bf16x8 (&qf)[QS][2], f32x4 (&sacc)[QS][4], int q16, int g) {
#pragma unroll
    for (int k4 = 0; k4 < 4; k4++) {
#pragma unroll
        for (int u = 0; u < QS; u++) sacc[u][k4] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            // SWZ: score row i = q16 of key tile k4 is key 32 (k4 / 2) + 8 (i / 4) + 4 (k4 % 2) + i % 4 (see the V^T read in att_pv)
            const bf16x8 kf = SWZ ? *reinterpret_cast<const bf16x8*>(reinterpret_cast<const char*>(Kt) + (k4 >> 1) * 4096 + (k4 & 1) * 512 +
                                                                     (((8 * (q16 >> 2) + (q16 & 3)) * 128 + ((g ^ att_kswz(q16)) << 4)) ^ (ks << 6)))
                                  : *reinterpret_cast<const bf16x8*>(&Kt[(16 * k4 + q16) * AK_LD + ks * 32 + g * 8]);
#pragma unroll
            for (int u = 0; u < QS; u++) {
                sacc[u][k4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[u][ks], sacc[u][k4], 0, 0, 0);
            }
        }
    }
}
// softmax in the exp2 domain on the RAW scores: p = 2^(s C - m C), C = log2(e) / sqrt(64), one FMA + one v_exp_f32 per score.
// Masking and the rescale of O are wave-uniform branches: only the last tile(s) of a sequence mask, and the running maximum
// stops moving after the first few tiles.
template <int QS, bool SWZ>
__device__ __forceinline__ void att_softmax(f32x4 (&sacc)[QS][4], bf16x8 (&pf)[QS][2], f32x4 (&o)[QS][4], float (&mrun)[QS], float (&lrun)[QS],
                                            const int (&kmax_q)[QS], int kt, int g) {
    constexpr float SC = 0.125f * 1.4426950408889634f;
#pragma unroll
    for (int u = 0; u < QS; u++) {
        if (!__all(kt * 64 + 64 <= kmax_q[u])) {
#pragma unroll
            for (int k4 = 0; k4 < 4; k4++)
#pragma unroll
                for (int r = 0; r < 4; r++)
                    if (kt * 64 + (SWZ ? 32 * (k4 >> 1) + 8 * g + 4 * (k4 & 1) : 16 * k4 + 4 * g) + r >= kmax_q[u]) sacc[u][k4][r] = -INFINITY;
        }
        // 16 scores -> one maximum as a chain of three-operand maxima (8 v_max3_f32; a balanced tree of pairs compiled to 23 instructions)
        float mloc = fmaxf(fmaxf(sacc[u][0][0], sacc[u][0][1]), sacc[u][0][2]);
        mloc = fmaxf(fmaxf(mloc, sacc[u][0][3]), sacc[u][1][0]);
        mloc = fmaxf(fmaxf(mloc, sacc[u][1][1]), sacc[u][1][2]);
        mloc = fmaxf(fmaxf(mloc, sacc[u][1][3]), sacc[u][2][0]);
        mloc = fmaxf(fmaxf(mloc, sacc[u][2][1]), sacc[u][2][2]);
        mloc = fmaxf(fmaxf(mloc, sacc[u][2][3]), sacc[u][3][0]);
        mloc = fmaxf(fmaxf(mloc, sacc[u][3][1]), sacc[u][3][2]);
        mloc = fmaxf(mloc, sacc[u][3][3]);
        // over the query's four lanes (q16 + 16 g): two v_permlane*_swap instead of two LDS-routed shuffles with a wait behind each
        const float mnew = fmaxf(mrun[u], rows4_max(mloc));      // raw-score domain
        const float msafe = mnew == -INFINITY ? 0.f : mnew;
        const float mc = msafe * SC;
        const float alpha = __builtin_amdgcn_exp2f((mrun[u] - msafe) * SC);     // mrun = -inf -> 0
        float psum = 0.f;
#pragma unroll
        for (int k4 = 0; k4 < 4; k4++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float p = __builtin_amdgcn_exp2f(fmaf(sacc[u][k4][r], SC, -mc));
                sacc[u][k4][r] = p; psum += p;
            }
        lrun[u] = lrun[u] * alpha + psum;
        mrun[u] = mnew;
        if (__any(alpha != 1.f)) {
#pragma unroll
            for (int dt = 0; dt < 4; dt++) o[u][dt] *= alpha;
        }
#pragma unroll
        for (int kp = 0; kp < 2; kp++) {
            typedef __attribute__((ext_vector_type(8))) float f32x8;
            const f32x8 pv = {sacc[u][2 * kp][0], sacc[u][2 * kp][1], sacc[u][2 * kp][2], sacc[u][2 * kp][3],
                              sacc[u][2 * kp + 1][0], sacc[u][2 * kp + 1][1], sacc[u][2 * kp + 1][2], sacc[u][2 * kp + 1][3]};
            pf[u][kp] = __builtin_convertvector(pv, bf16x8);
        }
    }
}
// O^T += V^T P^T ; k slots of key-pair tile kp: j < 4 -> key 32kp + 4g + j, j >= 4 -> key 32kp + 16 + 4g + (j - 4)
template <int QS, bool SWZ>
__device__ __forceinline__ void att_pv(const uint16_t* Vt, const bf16x8 (&pf)[QS][2], f32x4 (&o)[QS][4], int q16, int g) {
#pragma unroll
    for (int kp = 0; kp < 2; kp++)
#pragma unroll
        for (int dt = 0; dt < 4; dt++) {
            bf16x8 vf;
            if (SWZ) {                       // the lane's eight P entries of key-pair tile kp are the CONSECUTIVE keys 32 kp + 8 g .. + 7: one 16-byte chunk
                vf = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const char*>(Vt) + dt * 2048 + ((q16 * 128 + ((g ^ (q16 & 7)) << 4)) ^ (kp << 6)));
            } else {
                const uint16_t* vrow = &Vt[(16 * dt + q16) * AV_LD + 32 * kp + 4 * g];
                const uint2 lo = *reinterpret_cast<const uint2*>(vrow);
                const uint2 hi = *reinterpret_cast<const uint2*>(vrow + 16);
                vf = __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
            }
#pragma unroll
            for (int u = 0; u < QS; u++) {
                o[u][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[u][kp], o[u][dt], 0, 0, 0);
            }
        }
}
template <int QS, bool SWZ = false>
__device__ __forceinline__ void att_est_tile(const uint16_t* Kt, const uint16_t* Vt, const bf16x8 (&qf)[QS][2], f32x4 (&o)[QS][4],
                                             float (&mrun)[QS], float (&lrun)[QS], const int (&kmax_q)[QS], int kt, int q16, int g) {
    f32x4 sacc[QS][4];
    bf16x8 pf[QS][2];
    att_qk<QS, SWZ>(Kt, qf, sacc, q16, g);
    att_softmax<QS, SWZ>(sacc, pf, o, mrun, lrun, kmax_q, kt, g);
    att_pv<QS, SWZ>(Vt, pf, o, q16, g);
}
// QS query sub-tiles of 16 rows per wave, NW waves (block = 16*NW*QS rows): K / V^T fragments read from LDS once serve QS MFMAs;
// NW = 2 doubles the block count for single-utterance calls, whose 64-row tiles would not even fill the chip once.
// KSP = 2: a second group of NW waves works on the odd key tiles of the same queries (own LDS tiles, own register sets) and the two
// (max, sum, O) states are merged through LDS at the end.  One utterance gives only one wave per SIMD otherwise, and every LDS /
// MFMA / transcendental latency of the tile body is then exposed (phase stamps: ~1400 cycles of compute per 64-key tile).
template <int QS, int NW, int KSP = 1, bool CACHE = false>
__global__ __launch_bounds__(64 * NW * KSP) void k_attn_est(AttnEstArgs a) {
    __shared__ __attribute__((aligned(16))) uint16_t Ks_[KSP][2][64 * AK_LD];
    __shared__ __attribute__((aligned(16))) uint16_t Vs_[KSP][2][64 * AV_LD];
    const int lane = threadIdx.x & 63;
    const int kgrp = __builtin_amdgcn_readfirstlane((int)threadIdx.x / (64 * NW));      // key group of this wave
    const int tid = threadIdx.x - kgrp * 64 * NW, w = tid >> 6;                         // thread / wave index within the group
    uint16_t (*Ks)[64 * AK_LD] = Ks_[kgrp];
    uint16_t (*Vs)[64 * AV_LD] = Vs_[kgrp];
    constexpr int RB = 16 * NW;                                  // rows per query sub-tile group
    // grid (8 heads, query tiles): consecutive block ids go round the 8 XCDs, so XCD x serves head x only and a sequence's K / V^T of
    // that head is fetched into ONE L2 instead of all eight (r3_pmc_flow: 36 MB of fabric reads per launch before, for 6 MB of q/k/v)
#ifdef CV2_NO_XCD        // (A/B builds: tile-major block order, every XCD sees every head)
    const int lin_ = blockIdx.x + 8 * blockIdx.y, mt_ = gridDim.y;
    const int m0 = (lin_ % mt_) * RB * QS, h = lin_ / mt_;
#else
    const int m0 = blockIdx.y * RB * QS, h = blockIdx.x;
#endif
    const int q16 = lane & 15, g = lane >> 4;
    const int s = a.seq.tile_seq[m0 >> 6];           // sequences start on 128-row boundaries: one sequence per block
    uint16_t* orow[QS];
#pragma unroll
    for (int u = 0; u < QS; u++) orow[u] = a.out + (size_t)(m0 + RB * u + 16 * w + q16) * 512 + h * 64 + 4 * g;
    if (s < 0) {
#pragma unroll
        for (int u = 0; u < QS; u++)
#pragma unroll
            for (int dt = 0; dt < 4; dt++) *reinterpret_cast<uint2*>(orow[u] + 16 * dt) = make_uint2(0u, 0u);
        return;
    }
    const int start = a.seq.seq_start[s], len = a.seq.seq_len[s];
    const int t0 = m0 - start;
    if (CACHE && t0 >= len) {                          // padding rows of the sequence's last tile(s): zeros, as for a padding tile
#pragma unroll
        for (int u = 0; u < QS; u++)
#pragma unroll
            for (int dt = 0; dt < 4; dt++) *reinterpret_cast<uint2*>(orow[u] + 16 * dt) = make_uint2(0u, 0u);
        return;
    }
    // key source: the packed rows of this call, or the sequence's cache slot (keys 0 .. p0 + len)
    const int p0 = CACHE ? a.pos0[s] : 0;
    const int klen = p0 + len;
    const uint16_t* kbase = nullptr; const uint16_t* vbase = nullptr; long vld = 0;      // CACHE only; the packed-row form below is the round-1 code
    if (CACHE) {
        const long fr = a.kv_frames[s];
        kbase = a.kv[s] + a.slot * fr * 1024 + h * 64;
        vbase = a.kv[s] + a.slot * fr * 1024 + fr * 512 + (long)h * 64 * fr; vld = fr;
    }
    int tq[QS], kmax_q[QS];
#pragma unroll
    for (int u = 0; u < QS; u++) {
        tq[u] = t0 + RB * u + 16 * w + q16;                          // this lane's query frames
        kmax_q[u] = a.chunk > 0 ? min(klen, ((p0 + tq[u]) / a.chunk + 1) * a.chunk) : klen;
    }
    const int kmax_blk = a.chunk > 0 ? min(klen, ((p0 + t0 + RB * QS - 1) / a.chunk + 1) * a.chunk) : klen;
    const int ntiles = (kmax_blk + 63) / 64;

    bf16x8 qf[QS][2];
#pragma unroll
    for (int u = 0; u < QS; u++)
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
            qf[u][ks] = *reinterpret_cast<const bf16x8*>(a.qk + (size_t)(m0 + RB * u + 16 * w + q16) * 1024 + h * 64 + ks * 32 + g * 8);

    // staging: thread -> NCH K chunks (row, 16-B chunk) and NCH V^T chunks of the 64-key tile.  TWO register sets (A, B) keep the
    // next two tiles in flight: with one set the loop ran at the latency of a tile load (~3000 cycles per 64 keys at one
    // utterance, where only two small blocks share a CU), not at its cost.
    constexpr int NCH = 512 / (64 * NW);
    uint4 kregA0, kregA1, kregA2, kregA3, vregA0, vregA1, vregA2, vregA3;
    uint4 kregB0, kregB1, kregB2, kregB3, vregB0, vregB1, vregB2, vregB3;
    kregA2 = kregA3 = vregA2 = vregA3 = kregB2 = kregB3 = vregB2 = vregB3 = make_uint4(0u, 0u, 0u, 0u);
    kregA0 = kregA1 = vregA0 = vregA1 = kregB0 = kregB1 = vregB0 = vregB1 = make_uint4(0u, 0u, 0u, 0u);
#define ATT_GLOAD1(S, C, KROW)                                                                                               \
    if (NCH > C) {                                                                                                           \
        const int idx = tid + 64 * NW * C, kr = idx >> 3, kc = idx & 7;                                                       \
        if (CACHE) {                                                                                                          \
            kreg##S##C = *reinterpret_cast<const uint4*>(kbase + ((KROW) - start + kr) * 512 + kc * 8);                       \
            vreg##S##C = *reinterpret_cast<const uint4*>(vbase + kr * vld + ((KROW) - start) + kc * 8);                       \
        } else {                                                                                                              \
            kreg##S##C = *reinterpret_cast<const uint4*>(a.qk + (size_t)((KROW) + kr) * 1024 + 512 + h * 64 + kc * 8);        \
            vreg##S##C = *reinterpret_cast<const uint4*>(a.vt + (size_t)(h * 64 + kr) * a.R + (KROW) + kc * 8);               \
        }                                                                                                                     \
    }
#define ATT_GLOAD(S, KT) { const long krow_ = (long)start + (KT) * 64; ATT_GLOAD1(S, 0, krow_) ATT_GLOAD1(S, 1, krow_) ATT_GLOAD1(S, 2, krow_) ATT_GLOAD1(S, 3, krow_) }
#define ATT_LSTORE1(S, C, BUF)                                                                                               \
    if (NCH > C) {                                                                                                           \
        const int idx = tid + 64 * NW * C, kr = idx >> 3, kc = idx & 7;                                                       \
        *reinterpret_cast<uint4*>(&Ks[BUF][kr * AK_LD + kc * 8]) = kreg##S##C;                                                \
        uint2* vd = reinterpret_cast<uint2*>(&Vs[BUF][kr * AV_LD + kc * 8]);                                                  \
        vd[0] = make_uint2(vreg##S##C.x, vreg##S##C.y); vd[1] = make_uint2(vreg##S##C.z, vreg##S##C.w);                       \
    }
#define ATT_LSTORE(S, BUF) { ATT_LSTORE1(S, 0, BUF) ATT_LSTORE1(S, 1, BUF) ATT_LSTORE1(S, 2, BUF) ATT_LSTORE1(S, 3, BUF) }

    f32x4 o[QS][4];
    float mrun[QS], lrun[QS];
#pragma unroll
    for (int u = 0; u < QS; u++) {
        mrun[u] = -INFINITY; lrun[u] = 0.f;
#pragma unroll
        for (int dt = 0; dt < 4; dt++) o[u][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

    // tile kt sits in set A for even kt, in set B for odd kt; LDS buffer = kt & 1.  The loads are unconditional (a tile index past
    // the end is clamped to the last tile) so that the number in flight is fixed and the waits stay counted.
    // This group's tiles are kgrp, kgrp + KSP, ...; every group runs the same number of rounds (block barriers inside): a tile index
    // past the end is loaded from the last tile and fully masked (kt_eff >= ntiles makes every key invalid).
    const int last = ntiles - 1;
    const int rounds = (ntiles + KSP - 1) / KSP;
    int kmask_q[QS];                                                 // kmax_q, or 0 for a tile past the end
    SK_STAMP_DECL;
    SK_STAMP(0);
    ATT_GLOAD(A, min(kgrp, last))
    ATT_GLOAD(B, min(kgrp + KSP, last))
    SK_STAMP(1);
    // Both sets are consumed here (an empty statement the compiler has to wait in front of): hipcc issues the prologue's loads in an order
    // of its own, set A's after set B's, and the wait it places at the loop header covers the entry path too -- vmcnt(0) in EVERY
    // iteration, i.e. a wait for the set requested half an iteration earlier instead of the one requested two tiles ago (phase stamps at
    // one utterance: staging + wait 11 100 of a block's 23 500 cycles).  With nothing outstanding at the entry the header's wait is the
    // back edge's: the four loads of the other set stay in flight.
    asm volatile("" :: "v"(kregA0.x), "v"(kregA1.x), "v"(kregA2.x), "v"(kregA3.x), "v"(vregA0.x), "v"(vregA1.x), "v"(vregA2.x), "v"(vregA3.x),
                       "v"(kregB0.x), "v"(kregB1.x), "v"(kregB2.x), "v"(kregB3.x), "v"(vregB0.x), "v"(vregB1.x), "v"(vregB2.x), "v"(vregB3.x));
    for (int r = 0; r < rounds; r += 2) {
        {
            const int kt = r * KSP + kgrp;
#pragma unroll
            for (int u = 0; u < QS; u++) kmask_q[u] = kt < ntiles ? kmax_q[u] : 0;
            SK_ACC(2, ATT_LSTORE(A, 0) __syncthreads());             // [2] staging incl. the wait for the tile's loads
            ATT_GLOAD(A, min(kt + 2 * KSP, last))
            SK_ACC(3, att_est_tile<QS>(Ks[0], Vs[0], qf, o, mrun, lrun, kmask_q, kt, q16, g));     // [3] tile compute
        }
        if (r + 1 >= rounds) break;
        {
            const int kt = (r + 1) * KSP + kgrp;
#pragma unroll
            for (int u = 0; u < QS; u++) kmask_q[u] = kt < ntiles ? kmax_q[u] : 0;
            SK_ACC(2, ATT_LSTORE(B, 1) __syncthreads());
            ATT_GLOAD(B, min(kt + 2 * KSP, last))
            SK_ACC(3, att_est_tile<QS>(Ks[1], Vs[1], qf, o, mrun, lrun, kmask_q, kt, q16, g));
        }
    }
    SK_STAMP(4);
    if constexpr (KSP > 1) {
        // merge: the groups 1 .. KSP - 1 park (m, l, O) in LDS (each in its own tile buffers, free now), group 0 folds them in, in group order
        constexpr float SC = 0.125f * 1.4426950408889634f;
        static_assert((size_t)64 * NW * QS * 18 * sizeof(float) <= sizeof(Ks_[0]), "a group's tile buffers hold its parked state");
        __syncthreads();
        if (kgrp != 0) {
            float* mg = reinterpret_cast<float*>(&Ks_[kgrp][0][0]);      // [NW*64 threads][QS][18]
#pragma unroll
            for (int u = 0; u < QS; u++) {
                float* d = mg + ((size_t)tid * QS + u) * 18;
                d[0] = mrun[u]; d[1] = lrun[u];
#pragma unroll
                for (int dt = 0; dt < 4; dt++)
#pragma unroll
                    for (int e = 0; e < 4; e++) d[2 + dt * 4 + e] = o[u][dt][e];
            }
        }
        __syncthreads();
        if (kgrp != 0) return;
#pragma unroll
        for (int gq = 1; gq < KSP; gq++) {
            const float* mg = reinterpret_cast<const float*>(&Ks_[gq][0][0]);
#pragma unroll
            for (int u = 0; u < QS; u++) {
                const float* d = mg + ((size_t)tid * QS + u) * 18;
                const float m1 = d[0], l1 = d[1];
                const float M = fmaxf(mrun[u], m1);
                const float ms = M == -INFINITY ? 0.f : M;
                const float a0 = __builtin_amdgcn_exp2f((mrun[u] - ms) * SC), a1 = __builtin_amdgcn_exp2f((m1 - ms) * SC);
                lrun[u] = lrun[u] * a0 + l1 * a1;
                mrun[u] = M;
#pragma unroll
                for (int dt = 0; dt < 4; dt++)
#pragma unroll
                    for (int e = 0; e < 4; e++) o[u][dt][e] = o[u][dt][e] * a0 + d[2 + dt * 4 + e] * a1;
            }
        }
    }
#pragma unroll
    for (int u = 0; u < QS; u++) {
        float l = lrun[u];
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        const float inv = (tq[u] < len && l > 0.f) ? 1.f / l : 0.f;
#pragma unroll
        for (int dt = 0; dt < 4; dt++)
            *reinterpret_cast<uint2*>(orow[u] + 16 * dt) = make_uint2(pack_bf16x2(o[u][dt][0] * inv, o[u][dt][1] * inv), pack_bf16x2(o[u][dt][2] * inv, o[u][dt][3] * inv));
    }
#ifdef CV2_STAMPS
    SK_STAMP(5);
    if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) {
        for (int i_ = 0; i_ < 6; i_++) g_stamps[60][i_] = st_[i_];       // slot 60: last attention launch (tools/dbg_stamps_flow.py)
        g_stamps[60][6] = ntiles; g_stamps[60][7] = gridDim.x * gridDim.y;
    }
#endif
}

// The same attention with the K / V^T tiles going global -> LDS by DMA (4 waves, one key group): no staging registers (k_attn_est keeps
// two tiles = 64 VGPRs in flight per thread and stores them to LDS itself), LDS stages of 16 KB (round 4: three, two tiles in flight, three
// blocks per CU; round 5: two, one tile in flight, FOUR blocks per CU at 123 VGPRs -- the loop is bound by vector issue and the L2 -> LDS
// delivery, not by the tiles' latency: 176 -> 167 us per launch on a 32-utterance batch, profiles/r5_attention_experiments.txt).  A wave's DMA
// instruction writes 1 KiB = 8 rows of 128 B; lane l fetches chunk (l & 7) ^ (l >> 3) of row l >> 3, so the chunk c of row r lands in
// slot c ^ (r & 7) and the fragment reads (att_est_tile<.., true>) are conflict-free without padding.  The tile's keys are permuted among the
// score rows (att_est_tile): same products, another order inside a k group -- agreement with k_attn_est to fp32 round-off, not bit for bit.
#ifndef ATT_STAMP_BLOCK
#define ATT_STAMP_BLOCK 0
#endif
#ifndef ATT_STAGES
#define ATT_STAGES 2     // LDS stages of the DMA attention: 2 = 32 KB per block, four blocks per CU (3: two tiles in flight, three blocks; A/B)
#endif
template <int QS, bool CACHE = false>
__global__ __launch_bounds__(256) void k_attn_est_dma(AttnEstArgs a) {
    constexpr int NW = 4, RB = 16 * NW;
    __shared__ __attribute__((aligned(1024))) uint16_t Ks[ATT_STAGES][64 * 64];
    __shared__ __attribute__((aligned(1024))) uint16_t Vs[ATT_STAGES][64 * 64];
    const int lane = threadIdx.x & 63, tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef CV2_NO_XCD
    const int lin_ = blockIdx.x + 8 * blockIdx.y, mt_ = gridDim.y;
    const int m0 = (lin_ % mt_) * RB * QS, h = lin_ / mt_;
#else
    const int m0 = blockIdx.y * RB * QS, h = blockIdx.x;
#endif
    const int q16 = lane & 15, g = lane >> 4;
    const int s = a.seq.tile_seq[m0 >> 6];
    uint16_t* orow[QS];
#pragma unroll
    for (int u = 0; u < QS; u++) orow[u] = a.out + (size_t)(m0 + RB * u + 16 * w + q16) * 512 + h * 64 + 4 * g;
    const int start = s < 0 ? 0 : a.seq.seq_start[s], len = s < 0 ? 0 : a.seq.seq_len[s];
    const int t0 = m0 - start;
    if (s < 0 || (CACHE && t0 >= len)) {               // padding tile / padding rows of the sequence's last tile(s): zeros
#pragma unroll
        for (int u = 0; u < QS; u++)
#pragma unroll
            for (int dt = 0; dt < 4; dt++) *reinterpret_cast<uint2*>(orow[u] + 16 * dt) = make_uint2(0u, 0u);
        return;
    }
    const int p0 = CACHE ? a.pos0[s] : 0;
    const int klen = p0 + len;
    int tq[QS], kmax_q[QS];
#pragma unroll
    for (int u = 0; u < QS; u++) {
        tq[u] = t0 + RB * u + 16 * w + q16;
        kmax_q[u] = a.chunk > 0 ? min(klen, ((p0 + tq[u]) / a.chunk + 1) * a.chunk) : klen;
    }
    const int kmax_blk = a.chunk > 0 ? min(klen, ((p0 + t0 + RB * QS - 1) / a.chunk + 1) * a.chunk) : klen;
    const int ntiles = (kmax_blk + 63) / 64;
    // this lane's two K and two V^T DMA sources of tile 0 (wave w, instruction j: rows 8 (w + 4 j) .. + 7 of the tile); a tile further on is
    // 64 key rows (K) / 64 keys (V^T) further
    const int rr = lane >> 3, gc = (lane & 7) ^ rr;
    const uint16_t* ksrc[2]; const uint16_t* vsrc[2];
    long kstep;
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int r = 8 * (w + 4 * j) + rr;
        // key r of the tile is score row 4 ((r >> 3) & 3) + (r & 3) of its key tile (att_est_tile): its chunks are swizzled by that row's value
        const int gck = (lane & 7) ^ att_kswz(4 * ((r >> 3) & 3) + (r & 3));
        if (CACHE) {
            const long fr = a.kv_frames[s];
            ksrc[j] = a.kv[s] + a.slot * fr * 1024 + h * 64 + (long)r * 512 + gck * 8;
            vsrc[j] = a.kv[s] + a.slot * fr * 1024 + fr * 512 + (long)(h * 64 + r) * fr + gc * 8;
        } else {
            ksrc[j] = a.qk + (size_t)(start + r) * 1024 + 512 + h * 64 + gck * 8;
            vsrc[j] = a.vt + (size_t)(h * 64 + r) * a.R + start + gc * 8;
        }
    }
    kstep = CACHE ? 64 * 512 : 64 * 1024;
    // The DMA instructions are inline asm on purpose: hipcc orders every LDS read that may alias the destination of a DMA builtin behind
    // vmcnt(0) (here: each tile's fragment reads behind the DMA issued just before them, i.e. no tile in flight).  Written this way the
    // compiler knows nothing of the LDS writes and the waits below are the only ones; m0 is used by nothing else in this kernel (it is a
    // reserved register: hipcc rejects it on a clobber list, and itself writes it only for DMA builtins, LDS-direct and GWS operations).
    auto dma = [&](int kt, int st) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const unsigned kd = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(&Ks[st][(w + 4 * j) * 512]);
            const unsigned vd = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(&Vs[st][(w + 4 * j) * 512]);
            asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(kd), "v"(ksrc[j] + (long)kt * kstep) : "memory");
            asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(vd), "v"(vsrc[j] + (long)kt * 64) : "memory");
        }
    };
    f32x4 o[QS][4];
    float mrun[QS], lrun[QS];
#pragma unroll
    for (int u = 0; u < QS; u++) {
        mrun[u] = -INFINITY; lrun[u] = 0.f;
#pragma unroll
        for (int dt = 0; dt < 4; dt++) o[u][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const int last = ntiles - 1;
    SK_STAMP_DECL;
    SK_STAMP(0);
    dma(0, 0);
#if ATT_STAGES == 3
    dma(min(1, last), 1);
#endif
    // q fragments: loaded AFTER the first DMAs and consumed (empty asm) before the loop, so that the compiler's own wait for these loads sits
    // here -- placed at their first use inside the loop it would be a vmcnt(0) in every iteration, i.e. a wait for the DMA just issued
    bf16x8 qf[QS][2];
#pragma unroll
    for (int u = 0; u < QS; u++)
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
            qf[u][ks] = *reinterpret_cast<const bf16x8*>(a.qk + (size_t)(m0 + RB * u + 16 * w + q16) * 1024 + h * 64 + ks * 32 + g * 8);
#pragma unroll
    for (int u = 0; u < QS; u++)
#pragma unroll
        for (int ks = 0; ks < 2; ks++) asm volatile("" : "+v"(qf[u][ks]));
    int st = 0;
    SK_STAMP(1);
#if ATT_STAGES == 2
    // two stages (32 KB of LDS: four blocks per CU instead of three): one tile in flight while the other is worked on
    for (int kt = 0; kt < ntiles; kt++) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        dma(min(kt + 1, last), st ^ 1);
        att_est_tile<QS, true>(Ks[st], Vs[st], qf, o, mrun, lrun, kmax_q, kt, q16, g);
        st ^= 1;
    }
#else
    for (int kt = 0; kt < ntiles; kt++) {
        // four DMA instructions per tile and wave, always (a tile index past the end re-fetches the last tile): tile kt has landed when at
        // most the four of tile kt + 1 are outstanding; the barrier makes that true for every wave's share and says that every wave is
        // done with tile kt - 1 (its fragment reads have returned: lgkmcnt), whose stage the next DMA overwrites
        SK_TICK(ta_);
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        SK_TICK(tb_);
        __builtin_amdgcn_s_barrier();            // (no fence: __syncthreads() would add waits of its own)
        SK_TICK(tc_);
        dma(min(kt + 2, last), st >= 1 ? st - 1 : 2);
        SK_TICK(td_);
        att_est_tile<QS, true>(Ks[st], Vs[st], qf, o, mrun, lrun, kmax_q, kt, q16, g);
        SK_TICK(te_);
        SK_ADD(2, tb_ - ta_); SK_ADD(3, tc_ - tb_); SK_ADD(6, td_ - tc_); SK_ADD(7, te_ - td_);
        st = st == 2 ? 0 : st + 1;
    }
#endif
    SK_STAMP(4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the look-ahead DMAs target this block's LDS: they land before it is released)
#pragma unroll
    for (int u = 0; u < QS; u++) {
        float l = lrun[u];
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        const float inv = (tq[u] < len && l > 0.f) ? 1.f / l : 0.f;
#pragma unroll
        for (int dt = 0; dt < 4; dt++)
            *reinterpret_cast<uint2*>(orow[u] + 16 * dt) = make_uint2(pack_bf16x2(o[u][dt][0] * inv, o[u][dt][1] * inv), pack_bf16x2(o[u][dt][2] * inv, o[u][dt][3] * inv));
    }
#ifdef CV2_STAMPS
    SK_STAMP(5);
    if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == ATT_STAMP_BLOCK) {
        for (int i_ = 0; i_ < 8; i_++) g_stamps[61][i_] = st_[i_];       // slot 61: last DMA attention launch (tools/dbg_stamps_flow.py)
        g_stamps[62][0] = ntiles; g_stamps[62][1] = gridDim.x * gridDim.y;
    }
#endif
}

// One utterance, full context (round 5): the four-key-group form (k_attn_est<1, 4, 4>: 64 query rows x one head per 1024-thread block, group
// g works on the key tiles g, g + 4, ..., the four states merged through LDS) with its tiles by LDS DMA like k_attn_est_dma: no staging
// registers, no register -> LDS copies (phase stamps of the register-staged form at one utterance: staging + waits 9 800 of a block's 23 500
// cycles), two 16 KB stages per group.  The tile body is att_est_tile<1, true> (the DMA form's key permutation): agreement with k_attn_est to
// fp32 round-off, so -- like the other DMA forms -- it serves full-context calls only; a stream's chunk-masked recompute keeps the arithmetic
// of its cached continuation.
template <bool CACHE = false>
__global__ __launch_bounds__(1024) void k_attn_est_dma4(AttnEstArgs a) {
    constexpr int NW = 4, KSP = 4, RB = 64;
    __shared__ __attribute__((aligned(1024))) uint16_t Ts[KSP][2][2][64 * 64];          // [group][stage][K | V^T]: 128 KB
    const int lane = threadIdx.x & 63;
    const int kgrp = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);
    const int tid = threadIdx.x & 255;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.y * RB, h = blockIdx.x;
    const int q16 = lane & 15, g = lane >> 4;
    const int s = a.seq.tile_seq[m0 >> 6];
    uint16_t* orow[1] = {a.out + (size_t)(m0 + 16 * w + q16) * 512 + h * 64 + 4 * g};
    if (s < 0) {
        if (kgrp == 0) {
#pragma unroll
            for (int dt = 0; dt < 4; dt++) *reinterpret_cast<uint2*>(orow[0] + 16 * dt) = make_uint2(0u, 0u);
        }
        return;
    }
    const int start = a.seq.seq_start[s], len = a.seq.seq_len[s];
    const int t0 = m0 - start;
    if (CACHE && t0 >= len) {                          // padding rows of the sequence's last tile(s): zeros, as for a padding tile
        if (kgrp == 0) {
#pragma unroll
            for (int dt = 0; dt < 4; dt++) *reinterpret_cast<uint2*>(orow[0] + 16 * dt) = make_uint2(0u, 0u);
        }
        return;
    }
    // cached streaming (CACHE): keys / values come from the sequence's cache slot, the queries are the call's new frames at pos0 + t
    const int p0 = CACHE ? a.pos0[s] : 0;
    const int klen = p0 + len;
    int tq[1], kmax_q[1];
    tq[0] = t0 + 16 * w + q16;
    kmax_q[0] = a.chunk > 0 ? min(klen, ((p0 + tq[0]) / a.chunk + 1) * a.chunk) : klen;
    const int kmax_blk = a.chunk > 0 ? min(klen, ((p0 + t0 + RB - 1) / a.chunk + 1) * a.chunk) : klen;
    const int ntiles = (kmax_blk + 63) / 64;
    const int rounds = (ntiles + KSP - 1) / KSP, last = ntiles - 1;
    // this lane's DMA sources of tile 0 (wave w of the group, instruction j: rows 8 (w + 4 j) .. + 7 of the tile), as in k_attn_est_dma
    const int rr = lane >> 3, gc = (lane & 7) ^ rr;
    const uint16_t* ksrc[2]; const uint16_t* vsrc[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int r = 8 * (w + 4 * j) + rr;
        const int gck = (lane & 7) ^ att_kswz(4 * ((r >> 3) & 3) + (r & 3));
        if (CACHE) {
            const long fr = a.kv_frames[s];
            ksrc[j] = a.kv[s] + a.slot * fr * 1024 + h * 64 + (long)r * 512 + gck * 8;
            vsrc[j] = a.kv[s] + a.slot * fr * 1024 + fr * 512 + (long)(h * 64 + r) * fr + gc * 8;
        } else {
            ksrc[j] = a.qk + (size_t)(start + r) * 1024 + 512 + h * 64 + gck * 8;
            vsrc[j] = a.vt + (size_t)(h * 64 + r) * a.R + start + gc * 8;
        }
    }
    const long kstep = CACHE ? 64 * 512 : 64 * 1024;
    auto dma = [&](int kt, int st) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const unsigned kd = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(&Ts[kgrp][st][0][(w + 4 * j) * 512]);
            const unsigned vd = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(&Ts[kgrp][st][1][(w + 4 * j) * 512]);
            asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(kd), "v"(ksrc[j] + (long)kt * kstep) : "memory");
            asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(vd), "v"(vsrc[j] + (long)kt * 64) : "memory");
        }
    };
    f32x4 o[1][4];
    float mrun[1] = {-INFINITY}, lrun[1] = {0.f};
#pragma unroll
    for (int dt = 0; dt < 4; dt++) o[0][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    dma(min(kgrp, last), 0);
    bf16x8 qf[1][2];
#pragma unroll
    for (int ks = 0; ks < 2; ks++)
        qf[0][ks] = *reinterpret_cast<const bf16x8*>(a.qk + (size_t)(m0 + 16 * w + q16) * 1024 + h * 64 + ks * 32 + g * 8);
#pragma unroll
    for (int ks = 0; ks < 2; ks++) asm volatile("" : "+v"(qf[0][ks]));          // (the compiler's wait for q sits here, not inside the loop)
    int st = 0;
    for (int r = 0; r < rounds; r++) {
        const int kt = r * KSP + kgrp;                        // a tile index past the end re-fetches the last tile and is fully masked
        int kmask_q[1] = {kt < ntiles ? kmax_q[0] : 0};
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                         // every wave's share of tile r has landed; every wave is done with tile r - 1
        dma(min(kt + KSP, last), st ^ 1);
        att_est_tile<1, true>(Ts[kgrp][st][0], Ts[kgrp][st][1], qf, o, mrun, lrun, kmask_q, kt, q16, g);
        st ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    // merge: the groups 1 .. 3 park (m, l, O) in their own tile buffers (free now), group 0 folds them in, in group order (k_attn_est's merge)
    constexpr float SC = 0.125f * 1.4426950408889634f;
    static_assert((size_t)64 * NW * 18 * sizeof(float) <= sizeof(Ts[0]), "a group's tile buffers hold its parked state");
    __syncthreads();
    if (kgrp != 0) {
        float* d = reinterpret_cast<float*>(&Ts[kgrp][0][0][0]) + (size_t)tid * 18;
        d[0] = mrun[0]; d[1] = lrun[0];
#pragma unroll
        for (int dt = 0; dt < 4; dt++)
#pragma unroll
            for (int e = 0; e < 4; e++) d[2 + dt * 4 + e] = o[0][dt][e];
    }
    __syncthreads();
    if (kgrp != 0) return;
#pragma unroll
    for (int gq = 1; gq < KSP; gq++) {
        const float* d = reinterpret_cast<const float*>(&Ts[gq][0][0][0]) + (size_t)tid * 18;
        const float m1 = d[0], l1 = d[1];
        const float M = fmaxf(mrun[0], m1);
        const float ms = M == -INFINITY ? 0.f : M;
        const float a0 = __builtin_amdgcn_exp2f((mrun[0] - ms) * SC), a1 = __builtin_amdgcn_exp2f((m1 - ms) * SC);
        lrun[0] = lrun[0] * a0 + l1 * a1;
        mrun[0] = M;
#pragma unroll
        for (int dt = 0; dt < 4; dt++)
#pragma unroll
            for (int e = 0; e < 4; e++) o[0][dt][e] = o[0][dt][e] * a0 + d[2 + dt * 4 + e] * a1;
    }
    float l = lrun[0];
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    const float inv = (tq[0] < len && l > 0.f) ? 1.f / l : 0.f;
#pragma unroll
    for (int dt = 0; dt < 4; dt++)
        *reinterpret_cast<uint2*>(orow[0] + 16 * dt) = make_uint2(pack_bf16x2(o[0][dt][0] * inv, o[0][dt][1] * inv), pack_bf16x2(o[0][dt][2] * inv, o[0][dt][3] * inv));
}

// =========================================================================== host side
struct Layout {
    int S = 0, rows = 0;
    std::vector<int> start, len, ext;          // ext = padded extent in rows (multiple of 128)
    int* d_tile_seq = nullptr; int* d_start = nullptr; int* d_len = nullptr;   // device copies (workspace slices)
    const int4* d_tile_info = nullptr;
    SeqTable tab() const { return SeqTable{d_tile_seq, d_start, d_len, d_tile_info}; }
};
static int pad_rows(int len) { return (len + 8 + 127) / 128 * 128; }
// ints per layout slice of the int-table region: per-tile records (int4) + tile_seq + per-sequence tables, 16-B aligned
static size_t itab_per(int R, int max_seqs) { return ((size_t)5 * (R / 64) + 4 * max_seqs + 64 + 3) / 4 * 4; }

struct cv2_flow {
    cv2_flow_dims d;
    cv2_flow_weights w;
    hipStream_t stream0;
    // workspace
    char* ws; size_t ws_bytes;
    int R;                       // row capacity
    int TPmax, Pmax;
    float* temb_tab;             // [n_timesteps][14][256]
    float* t_steps_dev;          // [n_timesteps]
    float* tmp_t;                // [32][1024] x 3 scratch for the time MLP
    std::vector<float> t_host, dt_host;
    // row buffers
    uint16_t *a0, *hb, *xa, *xbuf, *cat, *lnb, *qk, *vt, *att, *ff;      // bf16
    float *xf, *rf, *vf, *xs, *mu, *spk, *tmpf;                           // fp32
    float *tail_part; int *tail_ticket;                                  // k_tail_panel<S > 1>: [128 panels][4 parts][16][256] partial tiles, [128] arrival counters (zero between launches)
    float *ac, *bd; uint16_t* probs; uint16_t *pos, *posp;
    // encoder buffers (C = 512)
    uint16_t *e_a, *e_b, *e_ln, *e_qkv, *e_vt, *e_att, *e_ff; float *e_x, *e_tmp;
    int* itab;                   // int tables region
    void* ptab;                  // pointer tables region
    std::vector<char> host_stage;
    // hipGraph replay of whole cv2_flow_inference calls (round 6): every kernel argument of a call is a function of its shape key -- the
    // utterances' token / prompt lengths, the flags -- and of engine-owned device addresses (the callers' pointers and the layouts travel
    // through device tables uploaded before the launches), so the ~2 300 launches of a call with a shape seen before are ONE graph launch.
    struct FlowGraph { hipGraphExec_t exec = nullptr; unsigned long last = 0; };
    std::map<std::vector<int>, FlowGraph> graphs;       // captured shapes (LRU, FG_MAX)
    std::map<std::vector<int>, int> seen;               // uses of shapes not captured yet (bounded)
    unsigned long graph_tick = 0;
};
#define FG_MAX 6

struct Carver {
    char* base; size_t off = 0;
    template <typename T> T* take(size_t n) {
        size_t o = off; off += (n * sizeof(T) + 255) & ~(size_t)255;
        return base ? reinterpret_cast<T*>(base + o) : nullptr;
    }
};

static size_t flow_carve(const cv2_flow_dims& d, cv2_flow* h, char* base) {
    Carver c{base};
    const size_t R = (size_t)d.max_rows, RG = R + GUARD + 8;
    const int TP = pad_rows(d.max_len), P = (2 * d.max_len - 1 + 127) / 128 * 128;
    cv2_flow tmp_{};
    cv2_flow& f = h ? *h : tmp_;
    f.R = (int)R; f.TPmax = TP; f.Pmax = P;
    f.temb_tab = c.take<float>((size_t)32 * 14 * 256);
    f.t_steps_dev = c.take<float>(32);
    f.tmp_t = c.take<float>((size_t)3 * 32 * 1024);
    f.a0 = c.take<uint16_t>(RG * 320); f.hb = c.take<uint16_t>(RG * 256); f.xa = c.take<uint16_t>(RG * 256);
    f.xbuf = c.take<uint16_t>(RG * 256); f.cat = c.take<uint16_t>(RG * 512); f.lnb = c.take<uint16_t>(RG * 256);
    f.qk = c.take<uint16_t>(RG * 1024); f.vt = c.take<uint16_t>((size_t)(512 + 64) * RG); f.att = c.take<uint16_t>(RG * 512);
    f.ff = c.take<uint16_t>(RG * 1024);
    f.tail_part = c.take<float>((size_t)128 * 4 * 16 * 256); f.tail_ticket = c.take<int>(128);
    f.xf = c.take<float>(R * 256); f.rf = c.take<float>(R * 256); f.vf = c.take<float>(R * 80); f.xs = c.take<float>(R * 80);
    f.mu = c.take<float>(R * 80); f.spk = c.take<float>((size_t)d.max_seqs * 80); f.tmpf = c.take<float>(R * 512);
    f.ac = c.take<float>((size_t)8 * TP * TP); f.bd = c.take<float>((size_t)8 * TP * P); f.probs = c.take<uint16_t>((size_t)8 * TP * TP);
    f.pos = c.take<uint16_t>((size_t)P * 512); f.posp = c.take<uint16_t>((size_t)P * 512);
    f.e_a = c.take<uint16_t>(RG * 512); f.e_b = c.take<uint16_t>(RG * 512); f.e_ln = c.take<uint16_t>(RG * 512);
    f.e_qkv = c.take<uint16_t>(RG * 1536); f.e_vt = c.take<uint16_t>((size_t)(512 + 64) * RG); f.e_att = c.take<uint16_t>(RG * 512);
    f.e_ff = c.take<uint16_t>(RG * 2048); f.e_x = c.take<float>(R * 512); f.e_tmp = c.take<float>(R * 512);
    f.itab = c.take<int>((size_t)8 * itab_per(R, d.max_seqs));      // layouts 0-3, ints 4; cached streaming: layouts 5-6, ints 7
    f.ptab = c.take<void*>((size_t)8 * d.max_seqs + 64);
    return c.off;
}

#ifdef CV2_STAMPS
extern "C" int cv2_debug_stamps_flow(unsigned long long* out_host) {       // this translation unit's copy of the stamp ring
    CV2_HIP(hipDeviceSynchronize());
    CV2_HIP(hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 64 * 8));
    return 0;
}
#endif
extern "C" size_t cv2_flow_workspace_bytes(const cv2_flow_dims* d) { return flow_carve(*d, nullptr, nullptr); }

// guard-offset views: bf16 row buffers are addressed from their first real row
#define GB(p, C) ((p) + (size_t)GUARD * (C))

static int upload_layout(cv2_flow* h, Layout& L, int slot, hipStream_t s) {
    // slot selects a disjoint slice of the int-table region
    const int per = (int)itab_per(h->R, h->d.max_seqs);
    int* base = h->itab + (size_t)slot * per;
    const int nt = L.rows / 64;
    std::vector<int> host(4 * nt + nt + 2 * L.S);              // [tile_info int4 x nt][tile_seq nt][start S][len S]
    int* ts = host.data() + 4 * nt;
    for (int i = 0; i < nt; i++) { ts[i] = -1; host[4 * i] = -1; host[4 * i + 1] = 0; host[4 * i + 2] = 0; host[4 * i + 3] = 0; }
    for (int q = 0; q < L.S; q++) {
        for (int r = L.start[q]; r < L.start[q] + L.ext[q]; r += 64) {
            ts[r / 64] = q;
            host[4 * (r / 64)] = q; host[4 * (r / 64) + 1] = L.start[q]; host[4 * (r / 64) + 2] = L.len[q];
        }
        ts[nt + q] = L.start[q];
        ts[nt + L.S + q] = L.len[q];
    }
    CV2_CHECK((int)host.size() <= per, "flow: layout table overflow");
    CV2_HIP(hipMemcpyAsync(base, host.data(), host.size() * sizeof(int), hipMemcpyHostToDevice, s));
    CV2_HIP(hipStreamSynchronize(s));            // host vector goes out of scope
    L.d_tile_info = reinterpret_cast<const int4*>(base);
    L.d_tile_seq = base + 4 * nt; L.d_start = L.d_tile_seq + nt; L.d_len = L.d_start + L.S;
    return 0;
}

static Layout make_layout(const std::vector<int>& lens, int lead = 0) {
    Layout L;
    L.S = (int)lens.size();
    int r = lead;                               // rows [0, lead) belong to no sequence
    for (int l : lens) { L.start.push_back(r); L.len.push_back(l); L.ext.push_back(pad_rows(l)); r += pad_rows(l); }
    L.rows = r;
    return L;
}

static int time_tables(cv2_flow* h, const float* t_dev, int n, float* tab_out, hipStream_t s) {
    // temb = Linear2(SiLU(Linear1(sinus(t)))) ; per resnet: Linear(Mish(temb))  (decoder.py:420-421, matcha decoder.py:58)
    float* e0 = h->tmp_t; float* e1 = e0 + 32 * 1024; float* e2 = e1 + 32 * 1024;
    hipLaunchKernelGGL(k_sinus, dim3(n), dim3(160), 0, s, t_dev, e0, n);
    if (skinny_gemm_launch(h->w.time1.w, h->w.time1.b, e0, e1, n, 1024, 320, s)) return -1;
    hipLaunchKernelGGL(k_act_inplace, dim3((n * 1024 + 255) / 256), dim3(256), 0, s, e1, n * 1024, ACT_SILU);
    if (skinny_gemm_launch(h->w.time2.w, h->w.time2.b, e1, e2, n, 1024, 1024, s)) return -1;
    hipLaunchKernelGGL(k_act_inplace, dim3((n * 1024 + 255) / 256), dim3(256), 0, s, e2, n * 1024, ACT_MISH);
    // 14 resnets: down, mid 0..11, up ; output [n][14][256] -> skinny writes [n][256] with row stride 256, so go via e0
    for (int r = 0; r < 14; r++) {
        const cv2_resnet& rn = r == 0 ? h->w.down.rn : r == 13 ? h->w.up_blk.rn : h->w.mid[r - 1].rn;
        if (skinny_gemm_launch(rn.mlp.w, rn.mlp.b, e2, e0, n, 256, 1024, s)) return -1;
        CV2_HIP(hipMemcpy2DAsync(tab_out + r * 256, (size_t)14 * 256 * 4, e0, 256 * 4, 256 * 4, n, hipMemcpyDeviceToDevice, s));
    }
    CV2_LAUNCH_CHECK();
    return 0;
}

extern "C" int cv2_flow_create(const cv2_flow_dims* d, const cv2_flow_weights* w, void* ws, size_t ws_bytes, void* stream, cv2_flow** out) {
    CV2_CHECK(d && w && ws && out, "cv2_flow_create: null argument");
    CV2_CHECK(d->max_rows % 128 == 0 && d->max_rows >= 256, "cv2_flow_create: max_rows must be a multiple of 128");
    CV2_CHECK(d->n_timesteps >= 1 && d->n_timesteps <= 31, "cv2_flow_create: n_timesteps out of range");
    CV2_CHECK(ws_bytes >= cv2_flow_workspace_bytes(d), "cv2_flow_create: workspace too small");
    cv2_flow* h = new cv2_flow();
    h->d = *d; h->w = *w; h->ws = (char*)ws; h->ws_bytes = ws_bytes;
    flow_carve(*d, h, (char*)ws);
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(ws, 0, cv2_flow_workspace_bytes(d), s) != hipSuccess) { delete h; return cv2_fail("cv2_flow_create: memset failed"); }
    // t schedule exactly as solve_euler walks it (flow_matching.py:91-121): t_span = 1 - cos(linspace(0,1,n+1) * pi/2) in fp32
    const int n = d->n_timesteps;
    std::vector<float> span(n + 1);
    for (int i = 0; i <= n; i++) {
        // torch.linspace (fp32): start + i*step for the first half, end - (n-i)*step for the second
        const float step = 1.0f / (float)n;
        const float lin = i < (n + 1) / 2 ? 0.0f + step * (float)i : 1.0f - step * (float)(n - i);
        span[i] = 1.0f - cosf(lin * 0.5f * 3.14159265358979323846f);
    }
    h->t_host.resize(n); h->dt_host.resize(n);
    float t = span[0], dt = span[1] - span[0];
    for (int st = 1; st <= n; st++) {
        h->t_host[st - 1] = t; h->dt_host[st - 1] = dt;
        t = t + dt;
        if (st < n) dt = span[st + 1] - t;
    }
    if (hipMemcpyAsync(h->t_steps_dev, h->t_host.data(), n * sizeof(float), hipMemcpyHostToDevice, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess) { delete h; return cv2_fail("cv2_flow_create: upload failed"); }
    if (time_tables(h, h->t_steps_dev, n, h->temb_tab, s)) { delete h; return -1; }
    *out = h;
    return 0;
}
extern "C" int cv2_flow_destroy(cv2_flow* h) {
    if (h) for (auto& g : h->graphs) if (g.second.exec) (void)hipGraphExecDestroy(g.second.exec);
    delete h;
    return 0;
}

// ------------------------------------------------------------------ estimator core
struct EstCtx { cv2_flow* h; const Layout* L; const float* temb; int chunk; hipStream_t s;
                const IncTabs* inc = nullptr; int step = 0, conv_i = 0, tb_i = 0;         // cached streaming: tables, Euler step, site counters
                bool qkv_chained = false; };                                              // the next transformer block's q / k / v^T are already there (k_tail_rows2<true>)

// cached streaming: hand the conv at this site its left context and keep this call's last two input rows
static int est_conv_tail(EstCtx& c, uint16_t* buf, int C) {
    if (!c.inc) return 0;
    const int nst = c.h->d.n_timesteps, ci = c.conv_i++;
    CV2_CHECK(ci < INC_CONVS, "flow: conv site counter overflow");
    ConvTailArgs a{buf, C, c.L->tab(), *c.inc, (0 * nst + c.step) * INC_CONVS + ci, (1 * nst + c.step) * INC_CONVS + ci};
    hipLaunchKernelGGL(k_conv_tail, dim3(c.L->S), dim3(256), 0, c.s, a);
    return 0;
}

static int est_gemm(EstCtx& c, GemmArgs a, int cfg) {
    a.seq = c.L->tab(); a.mask = 1;
    return gemm_launch_cfg(a, cfg, 1, true, c.s);
}

// resnet: A = bf16 input (guard-offset pointer) with C_in channels; leaves x (fp32, xf) and LN'd bf16 (lnb) for the first tblock
static int est_resnet(EstCtx& c, const cv2_resnet& rn, int ridx, uint16_t* A, int cin, const cv2_ln& next_ln) {
    cv2_flow* h = c.h; const int M = c.L->rows;
    if (est_conv_tail(c, A, cin)) return -1;
    {   // res_conv 1x1
        GemmArgs a = gemm_args(A, cin, 0, rn.res.w, M, 256, cin);
        a.bias = rn.res.b; a.out_f32 = h->rf; a.ldo = 256;
        if (est_gemm(c, a, 1)) return -1;
    }
    {   // block1: conv k3 -> LN -> Mish -> + time embedding
        GemmArgs a = gemm_args(A, cin, -2, rn.conv1.w, M, 256, 3 * cin);
        a.bias = rn.conv1.b; a.ln1_g = rn.ln1.g; a.ln1_b = rn.ln1.b; a.ln1_eps = 1e-5f; a.act = ACT_MISH;
        a.rowadd = c.temb + ridx * 256; a.rowadd_ld = 0;
        a.out_bf16 = GB(h->hb, 256); a.ldo16 = 256;
        if (est_gemm(c, a, 1)) return -1;
    }
    if (est_conv_tail(c, GB(h->hb, 256), 256)) return -1;
    {   // block2 + residual; second LN = norm1 of the first transformer block
        GemmArgs a = gemm_args(GB(h->hb, 256), 256, -2, rn.conv2.w, M, 256, 768);
        a.bias = rn.conv2.b; a.ln1_g = rn.ln2.g; a.ln1_b = rn.ln2.b; a.ln1_eps = 1e-5f; a.act = ACT_MISH;
        a.res = h->rf; a.ldres = 256; a.out_f32 = h->xf; a.ldo = 256;
        a.ln2_g = next_ln.g; a.ln2_b = next_ln.b; a.ln2_eps = 1e-5f; a.out_ln2 = GB(h->lnb, 256); a.ldo_ln2 = 256;
        if (est_gemm(c, a, 1)) return -1;
    }
    return 0;
}

// test hook: 1 / 0 = the four-wave estimator attention with / without LDS DMA staging (agreement to fp32 round-off, see k_attn_est_dma),
// -1 = the default (CV2_ATT_DMA)
static std::atomic<int> g_att_dma{-1};
extern "C" int cv2_flow_debug_attn_dma(int32_t on) { g_att_dma = on < 0 ? -1 : (on != 0); return 0; }
// test hook: 1 / 0 = cv2_flow_inference replays / does not replay a hipGraph for repeated shapes, -1 = the environment (CV2_FLOW_GRAPH, default off)
static std::atomic<int> g_flow_graph{-1};
extern "C" int cv2_flow_debug_graph(int32_t on) { g_flow_graph = on < 0 ? -1 : (on != 0); return 0; }

// transformer block; next_ln == null: last of its group -> bf16 copy of x goes to (xout, ldx).  have_qkv: the block's q / k / v^T are
// there already (the previous block's tail kernel chained this block's QKV projection on); next_tb != null: the block that follows in the
// group -- its QKV projection may be chained onto this block's tail (c.qkv_chained says whether it was).
static int est_tblock(EstCtx& c, const cv2_tblock& tb, const cv2_ln* next_ln, uint16_t* xout, long ldx, const cv2_tblock* next_tb = nullptr) {
    cv2_flow* h = c.h; const int M = c.L->rows;
    const bool have_qkv = c.qkv_chained;
    c.qkv_chained = false;
    if (!have_qkv) {
        GemmArgs a = gemm_args(GB(h->lnb, 256), 256, 0, tb.qkv.w, M, 1536, 256);
        a.out_bf16 = GB(h->qk, 1024); a.ldo16 = 1024; a.n_store = 1024;
        a.vt = h->vt + GUARD; a.vt_ld = h->R + GUARD + 8; a.vt_n0 = 1024;
        if (c.inc) {                 // cached streaming: the epilogue also files this call's keys / values in the cache slot of (Euler step, block)
            CV2_CHECK(c.tb_i < INC_TBLOCKS, "flow: transformer block counter overflow");
            a.kvc = c.inc->kv; a.kvc_frames = c.inc->kv_frames; a.kvc_pos0 = c.inc->pos0; a.kvc_slot = (long)c.step * INC_TBLOCKS + c.tb_i; a.kvc_k0 = 512;
        }
        if (est_gemm(c, a, 0)) return -1;
    }
    {
        AttnEstArgs a{GB(h->qk, 1024), h->vt + GUARD, GB(h->att, 512), c.L->tab(), c.chunk, (long)(h->R + GUARD + 8), nullptr, nullptr, nullptr, 0};
        if (c.inc) {                 // cached streaming: this call's keys / values join the cache, the attention reads all of it
            const int bi = c.tb_i++;
            CV2_CHECK(bi < INC_TBLOCKS, "flow: transformer block counter overflow");
            a.kv = c.inc->kv; a.kv_frames = c.inc->kv_frames; a.pos0 = c.inc->pos0; a.slot = (long)c.step * INC_TBLOCKS + bi;
        }
        // the four-wave forms stage their tiles by LDS DMA (k_attn_est_dma); CV2_ATT_DMA=0 (A/B, diagnostics): through registers as the two-group form
        static const bool dma_env = !(getenv("CV2_ATT_DMA") && getenv("CV2_ATT_DMA")[0] == '0');
        // below 4096 rows the keys are split over four wave groups of a 64-row block (one 10 s utterance: flow 26.5 -> 24.9 ms against two
        // groups of a 32-row block; a stream's chunk alone: first chunk 51.3 -> 50.8 ms); CV2_ATT_KSP=2 (A/B, diagnostics): the two-group form
        static const bool ksp4 = !(getenv("CV2_ATT_KSP") && getenv("CV2_ATT_KSP")[0] == '2');
        // (without the DMA forms: two key groups of four waves put two blocks on a CU above 256 blocks; CV2_ATT_KSP=4: four groups regardless)
        // One utterance above 2 048 packed rows (10.2 .. 20 s: 257 .. 511 blocks of the four-group form): a 1024-thread block fills a CU, so that
        // grid ran in TWO rounds (attention 13 -> 26 us per launch from 2 048 to 2 304 rows: flow 24.6 -> 33.5 ms for 6 % more frames).  From 257
        // blocks on the one-group DMA form runs (four 256-thread blocks per CU): 33.5 / 36.3 / 39.3 / 41.8 ms at 1 070 / 1 310 / 1 610 / 1 910
        // frames -> 28.7 / 31.5 / 35.0 / 37.3 (two key groups of a 512-thread block, the other candidate: 30.2 / 33.0 / 36.0 / 38.7);
        // tools/exp_flow_len.py.  Full-context calls only: a streaming call's chunk-masked recompute keeps the four-group form, the one its
        // cached continuation runs (k_attn_est<.., CACHE>), so that recomputed and cached chunks of a stream stay bit-identical -- any other
        // summation order moves the mel by ~3e-3 of its range through the 10 Euler steps.  CV2_ATT_DMA1_MIN=512: the round-4 threshold (A/B)
        static const int dma1_min = getenv("CV2_ATT_DMA1_MIN") ? atoi(getenv("CV2_ATT_DMA1_MIN")) : 257;
        static const bool dma4s = !(getenv("CV2_ATT_DMA4S") && getenv("CV2_ATT_DMA4S")[0] == '0');    // (A/B: streaming calls keep the register-staged form)
        static const bool dma4 = !(getenv("CV2_ATT_DMA4") && getenv("CV2_ATT_DMA4")[0] == '0');      // (A/B: the register-staged four-group form)
        static const bool ksp42 = !(getenv("CV2_ATT_KSP") && getenv("CV2_ATT_KSP")[0] == '4');
        const int dma_dbg = g_att_dma.load();
        const bool dma = dma_dbg < 0 ? dma_env : dma_dbg != 0;
        if (c.inc) {
            if (M / 128 * 8 >= 512) { if (dma) hipLaunchKernelGGL((k_attn_est_dma<2, true>), dim3(8, M / 128), dim3(256), 0, c.s, a); else hipLaunchKernelGGL((k_attn_est<2, 4, 1, true>), dim3(8, M / 128), dim3(256), 0, c.s, a); }
            else if (M / 64 * 8 >= 512) { if (dma) hipLaunchKernelGGL((k_attn_est_dma<1, true>), dim3(8, M / 64), dim3(256), 0, c.s, a); else hipLaunchKernelGGL((k_attn_est<1, 4, 1, true>), dim3(8, M / 64), dim3(256), 0, c.s, a); }
            else if (ksp4 && dma && dma4 && dma4s) hipLaunchKernelGGL(k_attn_est_dma4<true>, dim3(8, M / 64), dim3(1024), 0, c.s, a);
            else if (ksp4) hipLaunchKernelGGL((k_attn_est<1, 4, 4, true>), dim3(8, M / 64), dim3(1024), 0, c.s, a);     // (9 .. 15 streams' chunks, 257 .. 511 blocks: two groups of a 512-thread block measured +1 %: 12 streams 165 -> 167 audio-s/s; not taken)
            else hipLaunchKernelGGL((k_attn_est<1, 2, 2, true>), dim3(8, M / 32), dim3(256), 0, c.s, a);
        }
        else if (M / 128 * 8 >= 512) { if (dma) hipLaunchKernelGGL((k_attn_est_dma<2>), dim3(8, M / 128), dim3(256), 0, c.s, a); else hipLaunchKernelGGL((k_attn_est<2, 4>), dim3(8, M / 128), dim3(256), 0, c.s, a); }   // enough blocks to fill the chip twice
        else if (M / 64 * 8 >= (c.chunk > 0 ? 512 : dma1_min)) { if (dma) hipLaunchKernelGGL((k_attn_est_dma<1>), dim3(8, M / 64), dim3(256), 0, c.s, a); else hipLaunchKernelGGL((k_attn_est<1, 4>), dim3(8, M / 64), dim3(256), 0, c.s, a); }
        // below 512 blocks: the four-group form with its tiles by LDS DMA.  Chunk-masked (streaming) calls take it together with their cached
        // continuation (dma4s) over this whole range, so that recomputed and cached chunks of a stream keep ONE arithmetic whatever their row counts
        else if (ksp4 && dma && dma4 && (c.chunk == 0 || dma4s)) hipLaunchKernelGGL(k_attn_est_dma4<false>, dim3(8, M / 64), dim3(1024), 0, c.s, a);
        else if (ksp4 && (M / 64 * 8 <= 256 || !ksp42 || c.chunk > 0)) hipLaunchKernelGGL((k_attn_est<1, 4, 4>), dim3(8, M / 64), dim3(1024), 0, c.s, a);   // one utterance: the keys split over four wave groups of a 64-row block
        else if (ksp4) hipLaunchKernelGGL((k_attn_est<1, 4, 2>), dim3(8, M / 64), dim3(512), 0, c.s, a);     // more blocks than CUs (a 1024-thread block fills one): two groups, two blocks per CU
        else hipLaunchKernelGGL((k_attn_est<1, 2, 2>), dim3(8, M / 32), dim3(256), 0, c.s, a);       // (over two groups of a 32-row block)
    }
    static const bool tail_rows_off = getenv("CV2_FLOW_TAIL_ROWS") && getenv("CV2_FLOW_TAIL_ROWS")[0] == '0';     // A/B switch (diagnostics)
    static const long tail_rows_min = getenv("CV2_FLOW_TAIL_ROWS_MIN") ? atol(getenv("CV2_FLOW_TAIL_ROWS_MIN")) : 96;    // 64-row tiles from which the 64-row blocks are used (8 streaming chunks: 192 -> 184 ms first chunk)
    if ((long)(M / 64) < 200 || !tail_rows_off) {
        // O-projection, norm3, feed-forward and the next block's norm1 are row-local -> one launch per row panel: 16-row panels at one
        // utterance (128 blocks), 64-row blocks for batches (every weight fragment then feeds four MFMAs)
        TailArgs t{};
        t.att = GB(h->att, 512); t.lda = 512;
        t.Wo = tb.out.w; t.bo = tb.out.b; t.g3 = tb.norm3.g; t.b3 = tb.norm3.b; t.eps3 = 1e-5f;
        t.W1 = tb.ff1.w; t.b1 = tb.ff1.b; t.W2 = tb.ff2.w; t.b2 = tb.ff2.b;
        t.gn = next_ln ? next_ln->g : nullptr; t.bn = next_ln ? next_ln->b : nullptr; t.epsn = 1e-5f;
        t.xf = h->xf; t.out_ln = GB(h->lnb, 256); t.ldo_ln = 256; t.out_x = xout; t.ldo_x = ldx;
        t.seq = c.L->tab(); t.M_valid = M;
        t.part = h->tail_part; t.ticket = h->tail_ticket;
        t.row0 = c.inc && M > INC_LEAD ? INC_LEAD : 0;           // (cached chunks: skip the lead panels, see TailArgs::row0)
        if ((long)(M / 64) < tail_rows_min) return tail_panel_go(t, M, c.s);
        // batches: k_tail_rows2 (round 5); CV2_FLOW_TAIL_ROWS2=0: k_tail_rows<4>, =1: k_tail_rows2 without the chained QKV projection (A/B, diagnostics)
        static const int rows2 = getenv("CV2_FLOW_TAIL_ROWS2") ? atoi(getenv("CV2_FLOW_TAIL_ROWS2")) : 2;
        if (rows2 == 0) return tail_rows_go(t, M, c.s);
        const bool chain = rows2 >= 2 && next_tb && next_ln && !c.inc;      // (cached streaming chunks file their keys / values in the QKV GEMM's epilogue)
        if (chain) {
            t.Wqkv = next_tb->qkv.w; t.qk = GB(h->qk, 1024); t.vt = h->vt + GUARD; t.vt_ld = h->R + GUARD + 8;
            c.qkv_chained = true;
        }
        return tail_rows2_go(t, M, chain, c.s);
    }
    {
        GemmArgs a = gemm_args(GB(h->att, 512), 512, 0, tb.out.w, M, 256, 512);
        a.bias = tb.out.b; a.res = h->xf; a.ldres = 256; a.out_f32 = h->xf; a.ldo = 256;
        a.ln2_g = tb.norm3.g; a.ln2_b = tb.norm3.b; a.ln2_eps = 1e-5f; a.out_ln2 = GB(h->lnb, 256); a.ldo_ln2 = 256;
        if (est_gemm(c, a, 1)) return -1;
    }
    {
        GemmArgs a = gemm_args(GB(h->ff, 1024), 1024, 0, tb.ff2.w, M, 256, 1024);
        a.bias = tb.ff2.b; a.res = h->xf; a.ldres = 256;
        if (next_ln) {
            a.out_f32 = h->xf; a.ldo = 256;
            a.ln2_g = next_ln->g; a.ln2_b = next_ln->b; a.ln2_eps = 1e-5f; a.out_ln2 = GB(h->lnb, 256); a.ldo_ln2 = 256;
        } else {
            a.out_bf16 = xout; a.ldo16 = ldx;
        }
        GemmArgs a1 = gemm_args(GB(h->lnb, 256), 256, 0, tb.ff1.w, M, 1024, 256);
        a1.bias = tb.ff1.b; a1.act = ACT_GELU; a1.out_bf16 = GB(h->ff, 1024); a1.ldo16 = 1024;
        if (est_gemm(c, a1, 0)) return -1;
        if (est_gemm(c, a, 1)) return -1;
    }
    return 0;
}

static int est_block(EstCtx& c, const cv2_unet_block& b, int ridx, uint16_t* A, int cin, uint16_t* xout, long ldx) {
    if (est_resnet(c, b.rn, ridx, A, cin, b.tb[0].norm1)) return -1;
    for (int j = 0; j < 4; j++)
        if (est_tblock(c, b.tb[j], j < 3 ? &b.tb[j + 1].norm1 : nullptr, xout, ldx, j < 3 ? &b.tb[j + 1] : nullptr)) return -1;
    return 0;
}

// a0 (bf16 [rows][320]) -> vf (fp32 [rows][80])      CausalConditionalDecoder.forward, decoder.py:405-494
static int estimator_core(EstCtx& c) {
    cv2_flow* h = c.h; const int M = c.L->rows;
    const cv2_flow_weights& w = h->w;
    c.conv_i = 0; c.tb_i = 0;
    if (est_block(c, w.down, 0, GB(h->a0, 320), 320, GB(h->xa, 256), 256)) return -1;
    // skip connection: xa -> cat[:, 256:512]
    CV2_HIP(hipMemcpy2DAsync(GB(h->cat, 512) + 256, 512 * 2, GB(h->xa, 256), 256 * 2, 256 * 2, M, hipMemcpyDeviceToDevice, c.s));
    if (est_conv_tail(c, GB(h->xa, 256), 256)) return -1;
    {   // downsample tail: CausalConv1d(256,256,3)
        GemmArgs a = gemm_args(GB(h->xa, 256), 256, -2, w.down.tail.w, M, 256, 768);
        a.bias = w.down.tail.b; a.out_bf16 = GB(h->xbuf, 256); a.ldo16 = 256;
        if (est_gemm(c, a, 1)) return -1;
    }
    for (int i = 0; i < 12; i++) {
        uint16_t* A = i == 0 ? GB(h->xbuf, 256) : GB(h->xa, 256);
        uint16_t* xo = i == 11 ? GB(h->cat, 512) : GB(h->xa, 256);
        if (est_block(c, w.mid[i], 1 + i, A, 256, xo, i == 11 ? 512 : 256)) return -1;
    }
    if (est_block(c, w.up_blk, 13, GB(h->cat, 512), 512, GB(h->xa, 256), 256)) return -1;
    if (est_conv_tail(c, GB(h->xa, 256), 256)) return -1;
    {   // upsample tail
        GemmArgs a = gemm_args(GB(h->xa, 256), 256, -2, w.up_blk.tail.w, M, 256, 768);
        a.bias = w.up_blk.tail.b; a.out_bf16 = GB(h->xbuf, 256); a.ldo16 = 256;
        if (est_gemm(c, a, 1)) return -1;
    }
    if (est_conv_tail(c, GB(h->xbuf, 256), 256)) return -1;
    {   // final_block: conv -> LN -> Mish
        GemmArgs a = gemm_args(GB(h->xbuf, 256), 256, -2, w.final_conv.w, M, 256, 768);
        a.bias = w.final_conv.b; a.ln1_g = w.final_ln.g; a.ln1_b = w.final_ln.b; a.ln1_eps = 1e-5f; a.act = ACT_MISH;
        a.out_bf16 = GB(h->hb, 256); a.ldo16 = 256;
        if (est_gemm(c, a, 1)) return -1;
    }
    {   // final_proj 256 -> 80 (weights padded to 128 rows)
        GemmArgs a = gemm_args(GB(h->hb, 256), 256, 0, w.final_proj.w, M, 128, 256);
        a.bias = w.final_proj.b; a.out_f32 = h->vf; a.ldo = 80; a.n_store = 80;
        if (est_gemm(c, a, 0)) return -1;
    }
    CV2_LAUNCH_CHECK();
    return 0;
}

extern "C" int cv2_flow_estimator(cv2_flow* h, float* x, const float* mask, const float* mu, const float* t, const float* spks,
                                  const float* cond, int32_t T, int32_t streaming, void* stream) {
    CV2_CHECK(h && x && mu && t && spks && cond, "cv2_flow_estimator: null argument");
    (void)mask;   // the reference always passes an all-ones mask of length T (flow.py:272); lengths come from T
    hipStream_t s = (hipStream_t)stream;
    Layout L = make_layout({T, T});
    CV2_CHECK(L.rows <= h->R, "cv2_flow_estimator: T=%d exceeds the workspace (rows %d > %d)", T, L.rows, h->R);
    if (upload_layout(h, L, 0, s)) return -1;
    // time embedding for this t (both rows carry the same t, flow_matching.py:108)
    float* tab = h->temb_tab + (size_t)31 * 14 * 256;
    if (time_tables(h, t, 1, tab, s)) return -1;
    SeamPackArgs p{x, mu, spks, cond, T, L.start[1], GB(h->a0, 320), L.rows};
    hipLaunchKernelGGL(k_seam_pack, dim3(((long)L.rows * 320 + 255) / 256), dim3(256), 0, s, p);
    EstCtx c{h, &L, tab, streaming ? 50 : 0, s};
    if (estimator_core(c)) return -1;
    hipLaunchKernelGGL(k_seam_unpack, dim3((2L * 80 * T + 255) / 256), dim3(256), 0, s, (const float*)h->vf, x, (int)T, L.start[1]);
    CV2_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ streaming encoder cache (round 4)
// With streaming masks the encoder is chunk-causal too (upsample_encoder.py:243-306: static chunks of 25 tokens / 50 frames, the
// pre-lookahead conv reads 3 tokens ahead -- the call's look-ahead tokens --, every other convolution is causal), so the rows of
// finished chunks never change.  Per stream, behind the estimator's part of the cache: for each of the 10 conformer layers the keys
// K [cap][512] and values V^T [512][cap] of all positions so far (6 token-rate layers, cap = capT; 4 mel-rate layers, cap = capE),
// and the last two input rows of the two causal convolutions (pre_lookahead conv2; the k = 5 conv behind the nearest-neighbour
// upsampling, as the two token rows it repeats), double-buffered on the generation like the estimator's tails.
#define ENC_LAYERS 10
static long enc_capE(long frames) { return (frames + 127) / 128 * 128; }
static long enc_capT(long frames) { return (frames / 2 + 127) / 128 * 128; }
static size_t enc_layer_off(long frames, int l) {              // elements from the start of the encoder part
    const size_t t = (size_t)enc_capT(frames) * 1024, e = (size_t)enc_capE(frames) * 1024;
    return l < 6 ? l * t : 6 * t + (l - 6) * e;
}
static size_t enc_tail_off(long frames) { return enc_layer_off(frames, ENC_LAYERS); }
static size_t enc_elems(long frames) { return enc_tail_off(frames) + (size_t)2 * 2 * 2 * 512; }      // + [gen][conv][2 rows][512]

// this call's keys / values of one sequence -> its cache: K rows n0 .. n0 + n - 1, V^T columns likewise
struct EncAppendArgs { const uint16_t* qkv; const uint16_t* vt; long vt_ld; int start, n, n0; uint16_t* K; uint16_t* VT; long cap; };
__global__ __launch_bounds__(256) void k_enc_append(EncAppendArgs a) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.y == 0) {                                       // K: 64 x 16 B per row
        if (idx >= (long)a.n * 64) return;
        const long t = idx >> 6; const int c = (int)(idx & 63) * 8;
        *reinterpret_cast<uint4*>(a.K + (a.n0 + t) * 512 + c) = *reinterpret_cast<const uint4*>(a.qkv + (a.start + t) * 1536 + 1024 + c);
    } else {                                                     // V^T: one element per (feature, frame)
        if (idx >= (long)a.n * 512) return;
        const long c = idx / a.n, t = idx - c * a.n;
        a.VT[c * a.cap + a.n0 + t] = a.vt[c * a.vt_ld + a.start + t];
    }
}
// left context of a causal convolution: rows [start - nrep * 2, start) of `buf` <- the cached two rows (each repeated nrep times: the
// upsampled signal repeats a token row twice), and this call's last two rows of `src` (bf16 `buf` itself, or the fp32 stage output) -> cache
struct EncTailArgs { uint16_t* buf; const float* src_f32; const uint16_t* src_b16; long src_start; int start, n_src, n0, nrep; const uint16_t* rd; uint16_t* wr; };
__global__ __launch_bounds__(256) void k_enc_tail(EncTailArgs a) {
    for (int e = threadIdx.x; e < 2 * 512; e += 256) {
        const int j = e >> 9, c = e & 511;
        const uint16_t o = a.n0 > 0 ? a.rd[j * 512 + c] : (uint16_t)0;
        for (int r = 0; r < a.nrep; r++) a.buf[(long)(a.start - 2 * a.nrep + j * a.nrep + r) * 512 + c] = o;
        const int pos = a.n_src - 2 + j;                         // source row of the new tail (pos < 0: fewer than two new rows)
        uint16_t v;
        if (pos >= 0) v = a.src_f32 ? (uint16_t)(pack_bf16x2(a.src_f32[(a.src_start + pos) * 512 + c], 0.f) & 0xffffu) : a.src_b16[(a.src_start + pos) * 512 + c];     // (k_repeat2's conversion)
        else v = pos == -1 && a.n0 > 0 ? a.rd[512 + c] : (uint16_t)0;
        a.wr[j * 512 + c] = v;
    }
}
struct EncCacheCopyArgs { const uint16_t* src; uint16_t* dst; long scap, dcap; int n; };
__global__ __launch_bounds__(256) void k_enc_cache_copy(EncCacheCopyArgs a) {      // one layer: K rows [0, n), V^T columns [0, n)
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.y == 0) {
        if (idx >= (long)a.n * 64) return;
        *reinterpret_cast<uint4*>(a.dst + idx * 8) = *reinterpret_cast<const uint4*>(a.src + idx * 8);
    } else {
        if (idx >= (long)a.n * 512) return;
        const long c = idx / a.n, t = idx - c * a.n;
        a.dst[a.dcap * 512 + c * a.dcap + t] = a.src[a.scap * 512 + c * a.scap + t];
    }
}

// ------------------------------------------------------------------ encoder
struct EncInc {                                   // cached streaming: one entry per sequence of the call
    std::vector<uint16_t*> base;                  // the encoder part of the sequence's cache
    std::vector<long> frames;                     // its capacity in mel frames
    std::vector<int> n0_tok;                      // tokens cached before this call
    std::vector<int> gen;
};
struct EncCtx { cv2_flow* h; const Layout* L; int chunk; hipStream_t s; const EncInc* inc = nullptr; int layer = 0; int rate = 1; };

static int enc_gemm(EncCtx& c, GemmArgs a, int cfg) {
    a.seq = c.L->tab(); a.mask = 1;
    return gemm_launch_cfg(a, cfg, 1, true, c.s);
}
static int enc_ln(EncCtx& c, const float* x, const cv2_ln& ln, float eps, float scale, float* of, uint16_t* ob) {
    LnArgs a{x, 512, 512, ln.g, ln.b, eps, scale, c.L->tab(), c.L->rows, of, 512, ob, 512};
    hipLaunchKernelGGL(k_layernorm<8>, dim3((c.L->rows + 3) / 4), dim3(256), 0, c.s, a);
    return 0;
}

// one ConformerEncoderLayer over x (fp32 e_x, in place)      encoder_layer.py:160-236, attention.py:249-330
static int conformer_layer(EncCtx& c, const cv2_conformer& cl) {
    cv2_flow* h = c.h; const int M = c.L->rows; const long RG = h->R + GUARD + 8;
    enc_ln(c, h->e_x, cl.norm_mha, 1e-12f, 1.f, nullptr, GB(h->e_ln, 512));
    {
        GemmArgs a = gemm_args(GB(h->e_ln, 512), 512, 0, cl.qkv.w, M, 2048, 512);
        a.bias = cl.qkv.b; a.out_bf16 = GB(h->e_qkv, 1536); a.ldo16 = 1536; a.n_store = 1536;
        a.vt = h->e_vt + GUARD; a.vt_ld = RG; a.vt_n0 = 1536;
        if (enc_gemm(c, a, 0)) return -1;
    }
    int pos_T = -1;
    for (int q = 0; c.inc && q < c.L->S; q++) {
        // cached streaming: the rows of the call are positions n0 .. T - 1 of the sequence; their keys / values join the cache and the
        // attention runs over all of it (same products, same softmax row, same chunk mask as the whole-prefix form)
        const int nn = c.L->len[q], r0 = c.L->start[q], Mp = pad_rows(nn);
        const int n0 = c.inc->n0_tok[q] * c.rate, T = n0 + nn, Tk = (T + 127) / 128 * 128, P = (2 * T - 1 + 127) / 128 * 128;
        const long fr = c.inc->frames[q], cap = c.rate == 2 ? enc_capE(fr) : enc_capT(fr);
        uint16_t* Kc = c.inc->base[q] + enc_layer_off(fr, c.layer);
        uint16_t* VTc = Kc + (size_t)cap * 512;
        CV2_CHECK(Tk <= cap && (size_t)Mp * Tk <= (size_t)h->TPmax * h->TPmax && P <= h->Pmax, "flow: encoder cache capacity (T %d, cap %ld)", T, cap);
        {
            EncAppendArgs a{GB(h->e_qkv, 1536), h->e_vt + GUARD, RG, r0, nn, n0, Kc, VTc, cap};
            hipLaunchKernelGGL(k_enc_append, dim3(((long)nn * 512 + 255) / 256, 2), dim3(256), 0, c.s, a);
        }
        if (T != pos_T) {
            hipLaunchKernelGGL(k_pos_emb, dim3(P), dim3(256), 0, c.s, h->pos, T, P);
            GemmArgs a = gemm_args(h->pos, 512, 0, cl.pos.w, P, 512, 512);
            a.out_bf16 = h->posp; a.ldo16 = 512;
            if (gemm_launch_cfg(a, 0, 1, true, c.s)) return -1;
            pos_T = T;
        }
        const uint16_t* qkv = GB(h->e_qkv, 1536) + (size_t)r0 * 1536;
        {   // ac[h] = (q + u) K^T over the cache
            GemmArgs a = gemm_args(qkv, 1536, 0, Kc, Mp, Tk, 64);
            a.a_bstride = 64; a.ldw = 512; a.w_bstride = 64;
            a.out_f32 = h->ac; a.ldo = Tk; a.o_bstride = (long)Mp * Tk;
            if (gemm_launch_cfg(a, 0, 8, false, c.s)) return -1;
        }
        {   // bd[h] = (q + v) p^T
            GemmArgs a = gemm_args(qkv + 512, 1536, 0, h->posp, Mp, P, 64);
            a.a_bstride = 64; a.ldw = 512; a.w_bstride = 64;
            a.out_f32 = h->bd; a.ldo = P; a.o_bstride = (long)Mp * P;
            if (gemm_launch_cfg(a, 0, 8, false, c.s)) return -1;
        }
        {
            RelSmArgs a{h->ac, h->bd, h->probs, T, Mp, Tk, P, c.chunk, n0, nn};
            hipLaunchKernelGGL(k_relsoftmax, dim3(Mp, 8), dim3(256), (size_t)Tk * 4, c.s, a);
        }
        {   // att[:, h*64:(h+1)*64] = probs[h] V[h] from the cache
            GemmArgs a = gemm_args(h->probs, Tk, 0, VTc, Mp, 64, Tk);
            a.a_bstride = (long)Mp * Tk; a.ldw = cap; a.w_bstride = 64 * cap;
            a.out_bf16 = GB(h->e_att, 512) + (size_t)r0 * 512; a.ldo16 = 512; a.o16_bstride = 64;
            if (gemm_launch_cfg(a, 2, 8, false, c.s)) return -1;
        }
    }
    for (int q = 0; !c.inc && q < c.L->S; q++) {
        const int T = c.L->len[q], r0 = c.L->start[q], Tp = pad_rows(T), P = (2 * T - 1 + 127) / 128 * 128;
        if (T != pos_T) {       // linear_pos(pos_emb(T)) for this layer
            hipLaunchKernelGGL(k_pos_emb, dim3(P), dim3(256), 0, c.s, h->pos, T, P);
            GemmArgs a = gemm_args(h->pos, 512, 0, cl.pos.w, P, 512, 512);
            a.out_bf16 = h->posp; a.ldo16 = 512;
            if (gemm_launch_cfg(a, 0, 1, true, c.s)) return -1;
            pos_T = T;
        }
        const uint16_t* qkv = GB(h->e_qkv, 1536) + (size_t)r0 * 1536;
        {   // ac[h] = (q + u) k^T
            GemmArgs a = gemm_args(qkv, 1536, 0, qkv + 1024, Tp, Tp, 64);
            a.a_bstride = 64; a.ldw = 1536; a.w_bstride = 64;
            a.out_f32 = h->ac; a.ldo = Tp; a.o_bstride = (long)Tp * Tp;
            if (gemm_launch_cfg(a, 0, 8, false, c.s)) return -1;
        }
        {   // bd[h] = (q + v) p^T
            GemmArgs a = gemm_args(qkv + 512, 1536, 0, h->posp, Tp, P, 64);
            a.a_bstride = 64; a.ldw = 512; a.w_bstride = 64;
            a.out_f32 = h->bd; a.ldo = P; a.o_bstride = (long)Tp * P;
            if (gemm_launch_cfg(a, 0, 8, false, c.s)) return -1;
        }
        {
            RelSmArgs a{h->ac, h->bd, h->probs, T, Tp, Tp, P, c.chunk, 0, T};
            hipLaunchKernelGGL(k_relsoftmax, dim3(Tp, 8), dim3(256), (size_t)Tp * 4, c.s, a);
        }
        {   // att[:, h*64:(h+1)*64] = probs[h] V[h]
            GemmArgs a = gemm_args(h->probs, Tp, 0, h->e_vt + GUARD + r0, Tp, 64, Tp);
            a.a_bstride = (long)Tp * Tp; a.ldw = RG; a.w_bstride = 64 * RG;
            a.out_bf16 = GB(h->e_att, 512) + (size_t)r0 * 512; a.ldo16 = 512; a.o16_bstride = 64;
            if (gemm_launch_cfg(a, 2, 8, false, c.s)) return -1;
        }
    }
    {
        GemmArgs a = gemm_args(GB(h->e_att, 512), 512, 0, cl.out.w, M, 512, 512);
        a.bias = cl.out.b; a.res = h->e_x; a.ldres = 512; a.out_f32 = h->e_x; a.ldo = 512;
        if (enc_gemm(c, a, 0)) return -1;
    }
    enc_ln(c, h->e_x, cl.norm_ff, 1e-12f, 1.f, nullptr, GB(h->e_ln, 512));
    {
        GemmArgs a = gemm_args(GB(h->e_ln, 512), 512, 0, cl.w1.w, M, 2048, 512);
        a.bias = cl.w1.b; a.act = ACT_SILU; a.out_bf16 = GB(h->e_ff, 2048); a.ldo16 = 2048;
        if (enc_gemm(c, a, 0)) return -1;
    }
    {
        GemmArgs a = gemm_args(GB(h->e_ff, 2048), 2048, 0, cl.w2.w, M, 512, 2048);
        a.bias = cl.w2.b; a.res = h->e_x; a.ldres = 512; a.out_f32 = h->e_x; a.ldo = 512;
        if (enc_gemm(c, a, 0)) return -1;
    }
    CV2_LAUNCH_CHECK();
    return 0;
}

// Token-rate bf16 input rows (e_a; with look-ahead context rows when LA.len = T + 3) -> mel-rate rows:
//   final fp32 [rows2][512] after after_norm (out_f32, may be null) and/or bf16 (e_ln) for encoder_proj.
// LA = layout with the embed-stage lengths, LT = token-rate layout (T), L2 = mel-rate layout (2T); same starts for LA/LT.
// inc != null (cached streaming): the layouts hold only the NEW tokens / frames of every sequence (after INC_LEAD lead rows), the rest
// comes from the sequences' encoder caches.
static int encoder_core(cv2_flow* h, Layout& LA, Layout& LT, Layout& L2, int streaming, float* out_f32, hipStream_t s, const EncInc* inc = nullptr) {
    const cv2_flow_weights& w = h->w;
    const float SQ = sqrtf(512.f);
    EncCtx ca{h, &LA, streaming ? 25 : 0, s}, ct{h, &LT, streaming ? 25 : 0, s, inc, 0, 1}, c2{h, &L2, streaming ? 50 : 0, s, inc, 6, 2};
    const int M = LT.rows;
    auto tail = [&](int q, int kind, bool write) {      // the sequence's cached conv tail: [generation][conv][2 rows][512]
        const int g = (inc->gen[q] + (write ? 1 : 0)) & 1;
        return inc->base[q] + enc_tail_off(inc->frames[q]) + (size_t)((g * 2 + kind) * 2) * 512;
    };
    {   // embed: Linear -> LayerNorm(1e-5) -> * sqrt(512)     subsampling.py:69-113, embedding.py:268
        GemmArgs a = gemm_args(GB(h->e_a, 512), 512, 0, w.embed.w, M, 512, 512);
        a.bias = w.embed.b; a.out_f32 = h->e_tmp; a.ldo = 512;
        if (enc_gemm(ca, a, 0)) return -1;
        enc_ln(ca, h->e_tmp, w.embed_ln, 1e-5f, SQ, h->e_x, GB(h->e_b, 512));
    }
    {   // pre_lookahead: conv k4 over [t, t+3] -> leaky_relu -> causal conv k3 -> + x      upsample_encoder.py:82-102
        GemmArgs a = gemm_args(GB(h->e_b, 512), 512, 0, w.pre1.w, M, 512, 2048);
        a.bias = w.pre1.b; a.act = ACT_LRELU; a.act_slope = 0.01f; a.out_bf16 = GB(h->e_a, 512); a.ldo16 = 512;
        if (enc_gemm(ct, a, 0)) return -1;
        for (int q = 0; inc && q < LT.S; q++) {      // conv2's left context: the previous call's last two conv1 outputs
            EncTailArgs t{GB(h->e_a, 512), nullptr, GB(h->e_a, 512), LT.start[q], LT.start[q], LT.len[q], inc->n0_tok[q], 1, tail(q, 0, false), tail(q, 0, true)};
            hipLaunchKernelGGL(k_enc_tail, dim3(1), dim3(256), 0, s, t);
        }
        GemmArgs b = gemm_args(GB(h->e_a, 512), 512, -2, w.pre2.w, M, 512, 1536);
        b.bias = w.pre2.b; b.res = h->e_x; b.ldres = 512; b.out_f32 = h->e_x; b.ldo = 512;
        if (enc_gemm(ct, b, 0)) return -1;
    }
    for (int i = 0; i < 6; i++) { ct.layer = i; if (conformer_layer(ct, w.enc[i])) return -1; }
    {   // Upsample1D: nearest x2, left pad 4, conv k5   upsample_encoder.py:37-63 ; then up_embed
        RepeatArgs r{h->e_x, LT.tab(), L2.tab(), L2.rows, GB(h->e_a, 512)};
        hipLaunchKernelGGL(k_repeat2, dim3((L2.rows + 1) / 2), dim3(256), 0, s, r);
        for (int q = 0; inc && q < LT.S; q++) {      // the k = 5 conv's left context: the previous call's last two token rows, each repeated twice
            EncTailArgs t{GB(h->e_a, 512), h->e_x, nullptr, LT.start[q], L2.start[q], LT.len[q], inc->n0_tok[q], 2, tail(q, 1, false), tail(q, 1, true)};
            hipLaunchKernelGGL(k_enc_tail, dim3(1), dim3(256), 0, s, t);
        }
        GemmArgs a = gemm_args(GB(h->e_a, 512), 512, -4, w.up_conv.w, L2.rows, 512, 2560);
        a.bias = w.up_conv.b; a.out_bf16 = GB(h->e_b, 512); a.ldo16 = 512;
        if (enc_gemm(c2, a, 0)) return -1;
        GemmArgs b = gemm_args(GB(h->e_b, 512), 512, 0, w.up_embed.w, L2.rows, 512, 512);
        b.bias = w.up_embed.b; b.out_f32 = h->e_tmp; b.ldo = 512;
        if (enc_gemm(c2, b, 0)) return -1;
        enc_ln(c2, h->e_tmp, w.up_embed_ln, 1e-5f, SQ, h->e_x, nullptr);
    }
    for (int i = 0; i < 4; i++) { c2.layer = 6 + i; if (conformer_layer(c2, w.up[i])) return -1; }
    enc_ln(c2, h->e_x, w.after_norm, 1e-5f, 1.f, out_f32, GB(h->e_ln, 512));
    CV2_LAUNCH_CHECK();
    return 0;
}

static int check_rows(cv2_flow* h, const Layout& L, const char* who) {
    CV2_CHECK(L.rows <= h->R, "%s: %d packed rows exceed the workspace capacity %d", who, L.rows, h->R);
    for (int l : L.len) CV2_CHECK(l >= 1 && l <= h->d.max_len + 3, "%s: sequence length %d out of range (max %d)", who, l, h->d.max_len);
    CV2_CHECK(L.S <= h->d.max_seqs, "%s: too many sequences", who);
    return 0;
}

extern "C" int cv2_flow_encoder(cv2_flow* h, const float* xs, int32_t T, const float* context, int32_t streaming, float* out, void* stream) {
    CV2_CHECK(h && xs && out && T >= 1, "cv2_flow_encoder: bad argument");
    hipStream_t s = (hipStream_t)stream;
    Layout LT = make_layout({T}), L2 = make_layout({2 * T});
    Layout LA = LT;                                   // same rows; the 3 context rows sit in the zero tail of the sequence
    if (context) LA.len[0] += 3;
    if (check_rows(h, L2, "cv2_flow_encoder")) return -1;
    if (upload_layout(h, LA, 0, s) || upload_layout(h, LT, 1, s) || upload_layout(h, L2, 2, s)) return -1;
    // fp32 rows -> bf16 e_a (context rows appended)
    int zero = 0;
    int* off = h->itab + (size_t)4 * itab_per(h->R, h->d.max_seqs);
    CV2_HIP(hipMemcpyAsync(off, &zero, sizeof(int), hipMemcpyHostToDevice, s));
    CV2_HIP(hipStreamSynchronize(s));
    CastArgs ca{xs, GB(h->e_a, 512), LT.tab(), LT.rows, 512, off};
    hipLaunchKernelGGL(k_cast_rows, dim3(((long)LT.rows * 128 + 255) / 256), dim3(256), 0, s, ca);
    if (context) hipLaunchKernelGGL(k_ctx_rows, dim3(6), dim3(256), 0, s, context, GB(h->e_a, 512) + (size_t)T * 512);
    if (encoder_core(h, LA, LT, L2, streaming, h->e_tmp, s)) return -1;
    CV2_HIP(hipMemcpyAsync(out, h->e_tmp, (size_t)2 * T * 512 * sizeof(float), hipMemcpyDeviceToDevice, s));
    return 0;
}

extern "C" int cv2_flow_inference(cv2_flow* h, const cv2_flow_utt* utts, int32_t U, int32_t streaming, int32_t finalize, void* stream) {
    CV2_CHECK(h && utts && U >= 1, "cv2_flow_inference: bad argument");
    CV2_CHECK(2 * U <= h->d.max_seqs, "cv2_flow_inference: %d utterances exceed max_seqs/2", U);
    hipStream_t s = (hipStream_t)stream;
    const int la = finalize ? 0 : 3;                      // pre_lookahead_len (flow.py:260-263)
    std::vector<int> lensT, lens2, lensE;
    for (int u = 0; u < U; u++) {
        const int T = utts[u].n_tok - la;
        CV2_CHECK(T >= 1, "cv2_flow_inference: utterance %d has no tokens", u);
        CV2_CHECK(utts[u].n_prompt_feat >= 0 && utts[u].n_prompt_feat <= 2 * T, "cv2_flow_inference: prompt_feat longer than the mel (%d > %d)", utts[u].n_prompt_feat, 2 * T);
        lensT.push_back(T); lens2.push_back(2 * T);
    }
    Layout LT = make_layout(lensT), L2 = make_layout(lens2);
    Layout LA = LT;                                   // embed-stage lengths include the look-ahead rows (zero tail of each sequence)
    for (int u = 0; u < U; u++) LA.len[u] += la;
    lensE = lens2; lensE.insert(lensE.end(), lens2.begin(), lens2.end());
    Layout LE = make_layout(lensE);
    if (check_rows(h, LE, "cv2_flow_inference")) return -1;
    const int RU = L2.rows;
    if (upload_layout(h, LA, 0, s) || upload_layout(h, LT, 1, s) || upload_layout(h, L2, 2, s) || upload_layout(h, LE, 3, s)) return -1;
    // pointer / int tables
    {
        std::vector<const void*> ptrs(4 * U);
        std::vector<int> ints(U);
        for (int u = 0; u < U; u++) {
            ptrs[u] = utts[u].embedding; ptrs[U + u] = utts[u].prompt_feat; ptrs[2 * U + u] = utts[u].mel_out; ptrs[3 * U + u] = utts[u].tokens;
            ints[u] = utts[u].n_prompt_feat;
        }
        CV2_HIP(hipMemcpyAsync(h->ptab, ptrs.data(), ptrs.size() * sizeof(void*), hipMemcpyHostToDevice, s));
        int* ibase = h->itab + (size_t)4 * itab_per(h->R, h->d.max_seqs);
        CV2_HIP(hipMemcpyAsync(ibase, ints.data(), ints.size() * sizeof(int), hipMemcpyHostToDevice, s));
        CV2_HIP(hipStreamSynchronize(s));
        const void* const* dp = (const void* const*)h->ptab;
        auto launches = [&](hipStream_t s) -> int {
        // speaker projection
        SpkArgs sp{(const float* const*)dp, h->w.spk_w, h->w.spk_b, h->spk};
        hipLaunchKernelGGL(k_spk, dim3(U), dim3(128), 0, s, sp);
        // token embedding rows
        EmbedPtrArgs ea{(const int* const*)(dp + 3 * U), h->w.input_embedding, LA.tab(), LA.rows, GB(h->e_a, 512)};
        hipLaunchKernelGGL(k_embed_tokens_ptr, dim3((LA.rows + 1) / 2), dim3(256), 0, s, ea);
        if (encoder_core(h, LA, LT, L2, streaming, nullptr, s)) return -1;
        {   // encoder_proj 512 -> 80 -> mu
            GemmArgs a = gemm_args(GB(h->e_ln, 512), 512, 0, h->w.enc_proj.w, L2.rows, 128, 512);
            a.bias = h->w.enc_proj.b; a.out_f32 = h->mu; a.ldo = 80; a.n_store = 80; a.seq = L2.tab(); a.mask = 1;
            if (gemm_launch_cfg(a, 0, 1, true, s)) return -1;
        }
        // Euler loop (flow_matching.py:91-121)
        EstCtx c{h, &LE, nullptr, streaming ? 50 : 0, s};
        for (int st = 0; st < h->d.n_timesteps; st++) {
            PackArgs p{h->xs, h->mu, h->spk, (const float* const*)(dp + U), ibase, st == 0 ? h->w.rand_noise : nullptr,
                       st == 0 ? nullptr : h->vf, st == 0 ? 0.f : h->dt_host[st - 1], h->d.cfg_rate, L2.tab(), RU, U, GB(h->a0, 320),
                       nullptr, nullptr, RU, 0};
            hipLaunchKernelGGL(k_euler_pack, dim3(((long)RU * 20 + 255) / 256), dim3(256), 0, s, p);
            c.temb = h->temb_tab + (size_t)st * 14 * 256;
            if (estimator_core(c)) return -1;
        }
        {   // last update without a following pack: reuse k_euler_pack (a0 is scratch now)
            const int st = h->d.n_timesteps;
            PackArgs p{h->xs, h->mu, h->spk, (const float* const*)(dp + U), ibase, nullptr, h->vf, h->dt_host[st - 1], h->d.cfg_rate,
                       L2.tab(), RU, U, GB(h->a0, 320), nullptr, nullptr, RU, 0};
            hipLaunchKernelGGL(k_euler_pack, dim3(((long)RU * 20 + 255) / 256), dim3(256), 0, s, p);
        }
        int maxn2 = 1;
        for (int u = 0; u < U; u++) maxn2 = std::max(maxn2, lens2[u] - utts[u].n_prompt_feat);
        MelOutArgs mo{h->xs, (float* const*)(dp + 2 * U), ibase, L2.tab()};
        hipLaunchKernelGGL(k_mel_out, dim3(((long)maxn2 * 80 + 255) / 256, U), dim3(256), 0, s, mo);
        return 0;
        };
        // ---- CV2_FLOW_GRAPH=1 / cv2_flow_debug_graph(1): one graph launch for a shape seen before.  A shape is captured at its
        // (CV2_FLOW_GRAPH_AFTER + 1)-th use (default: the second), so a workload whose lengths never repeat pays nothing.  The key holds
        // everything the launches' arguments and kernel choices depend on beside engine-owned addresses.  OFF by default: measured at
        // configs[1] (profiles/r6_flow_graph_ab.txt) the replay is 0.4 ms SLOWER than the 2 300 launches (flow 24.73 against 24.32 ms; the
        // host stays ahead of the device with plain launches, and a graph's kernel nodes follow each other no faster than stream launches do)
        static const bool graph_env = getenv("CV2_FLOW_GRAPH") && getenv("CV2_FLOW_GRAPH")[0] == '1';
        static const int graph_after = getenv("CV2_FLOW_GRAPH_AFTER") ? atoi(getenv("CV2_FLOW_GRAPH_AFTER")) : 1;
        const int graph_dbg = g_flow_graph.load();
        const bool graph_on = graph_dbg < 0 ? graph_env : graph_dbg != 0;
        if (!graph_on) { if (launches(s)) return -1; CV2_LAUNCH_CHECK(); return 0; }
        std::vector<int> key{U, streaming, finalize, g_att_dma.load()};
        for (int u = 0; u < U; u++) { key.push_back(utts[u].n_tok); key.push_back(utts[u].n_prompt_feat); }
        auto it = h->graphs.find(key);
        if (it == h->graphs.end()) {
            int& n = h->seen[key];
            if (n++ < graph_after) {
                if (h->seen.size() > 256) h->seen.clear();
                if (launches(s)) return -1;
                CV2_LAUNCH_CHECK();
                return 0;
            }
            h->seen.erase(key);
            CV2_HIP(hipStreamSynchronize(s));                 // (the capture stream's work must follow what `s` holds; it is replayed on `s` below)
            hipStream_t cs = nullptr;
            CV2_HIP(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
            hipGraph_t g = nullptr;
            hipError_t e = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
            int rc = -1;
            if (e == hipSuccess) {
                rc = launches(cs);
                e = hipStreamEndCapture(cs, &g);
            }
            (void)hipStreamDestroy(cs);
            if (rc) return rc;
            CV2_HIP(e);
            hipGraphExec_t ge = nullptr;
            e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
            (void)hipGraphDestroy(g);
            CV2_HIP(e);
            if ((int)h->graphs.size() >= FG_MAX) {            // drop the least recently used shape
                auto old = h->graphs.begin();
                for (auto j = h->graphs.begin(); j != h->graphs.end(); ++j) if (j->second.last < old->second.last) old = j;
                (void)hipGraphExecDestroy(old->second.exec);
                h->graphs.erase(old);
            }
            cv2_flow::FlowGraph fg; fg.exec = ge;
            it = h->graphs.emplace(key, fg).first;
        }
        it->second.last = ++h->graph_tick;
        CV2_HIP(hipGraphLaunch(it->second.exec, s));
    }
    CV2_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ cached streaming
static size_t inc_kv_elems(const cv2_flow* h, long frames) { return (size_t)2 * h->d.n_timesteps * INC_TBLOCKS * frames * 1024; }
static size_t inc_tail_elems(const cv2_flow* h) { return (size_t)2 * 2 * h->d.n_timesteps * INC_CONVS * 1024; }

static size_t enc_part_off(const cv2_flow* h, long frames) { return inc_kv_elems(h, frames) + inc_tail_elems(h); }     // elements in front of the encoder part
extern "C" size_t cv2_flow_cache_bytes(const cv2_flow* h, int32_t frames) {
    if (!h || frames < 64 || frames % 64) return 0;
    return (enc_part_off(h, frames) + enc_elems(frames)) * sizeof(uint16_t);
}

// first n frames of every slot (and the convolution tails) of one cache -> another cache of a different capacity: a stream that
// starts from a prompt another call has already run (same prompt tokens, prompt mel, speaker embedding) begins with that prefix
struct CacheCopyArgs { const uint16_t* src; uint16_t* dst; long sfr, dfr; int n; };
__global__ __launch_bounds__(256) void k_cache_copy(CacheCopyArgs a) {      // grid (slots, 2 * 8): y < 8: K rows, y >= 8: V^T rows, 64 channels each
    const long slot = blockIdx.x;
    const int part = blockIdx.y >> 3, hd = blockIdx.y & 7, tid = threadIdx.x;
    const uint16_t* s = a.src + slot * a.sfr * 1024;
    uint16_t* d = a.dst + slot * a.dfr * 1024;
    if (part == 0) {                                               // K [frames][512]: 8 x 16 B of this head per frame
        for (long i = tid; i < (long)a.n * 8; i += 256) {
            const long r = i >> 3; const int ch = hd * 64 + (int)(i & 7) * 8;
            *reinterpret_cast<uint4*>(d + r * 512 + ch) = *reinterpret_cast<const uint4*>(s + r * 512 + ch);
        }
    } else {                                                       // V^T [512][frames]: n is even (whole chunks of 50)
        const int np = a.n >> 1;
        for (long i = tid; i < (long)64 * np; i += 256) {
            const long ch = hd * 64 + i / np, pr = i % np;
            *reinterpret_cast<uint32_t*>(d + a.dfr * 512 + ch * a.dfr + 2 * pr) = *reinterpret_cast<const uint32_t*>(s + a.sfr * 512 + ch * a.sfr + 2 * pr);
        }
    }
}
extern "C" int cv2_flow_cache_copy(const cv2_flow* h, const void* src, int32_t src_frames, void* dst, int32_t dst_frames, int32_t n_frames,
                                   void* stream) {
    CV2_CHECK(h && src && dst && src != dst, "cv2_flow_cache_copy: bad argument");
    CV2_CHECK(src_frames >= 64 && src_frames % 64 == 0 && dst_frames >= 64 && dst_frames % 64 == 0, "cv2_flow_cache_copy: capacities must be multiples of 64");
    CV2_CHECK(n_frames >= 0 && n_frames % 50 == 0 && n_frames <= src_frames && n_frames <= dst_frames, "cv2_flow_cache_copy: %d frames (whole chunks of 50, within both capacities)", n_frames);
    hipStream_t s = (hipStream_t)stream;
    const int slots = 2 * h->d.n_timesteps * INC_TBLOCKS;
    if (n_frames > 0) {
        CacheCopyArgs a{(const uint16_t*)src, (uint16_t*)dst, (long)src_frames, (long)dst_frames, (int)n_frames};
        hipLaunchKernelGGL(k_cache_copy, dim3(slots, 16), dim3(256), 0, s, a);
    }
    CV2_HIP(hipMemcpyAsync((uint16_t*)dst + inc_kv_elems(h, dst_frames), (const uint16_t*)src + inc_kv_elems(h, src_frames),
                           inc_tail_elems(h) * sizeof(uint16_t), hipMemcpyDeviceToDevice, s));
    {   // the encoder part: keys / values of the first n_frames / 2 tokens (layers 0-5) and n_frames frames (layers 6-9), the conv tails
        const uint16_t* se = (const uint16_t*)src + enc_part_off(h, src_frames);
        uint16_t* de = (uint16_t*)dst + enc_part_off(h, dst_frames);
        for (int l = 0; l < ENC_LAYERS && n_frames > 0; l++) {
            const int n = l < 6 ? n_frames / 2 : n_frames;
            EncCacheCopyArgs a{se + enc_layer_off(src_frames, l), de + enc_layer_off(dst_frames, l), l < 6 ? enc_capT(src_frames) : enc_capE(src_frames),
                               l < 6 ? enc_capT(dst_frames) : enc_capE(dst_frames), n};
            hipLaunchKernelGGL(k_enc_cache_copy, dim3(((long)n * 512 + 255) / 256, 2), dim3(256), 0, s, a);
        }
        CV2_HIP(hipMemcpyAsync(de + enc_tail_off(dst_frames), se + enc_tail_off(src_frames), (size_t)2 * 2 * 2 * 512 * sizeof(uint16_t), hipMemcpyDeviceToDevice, s));
    }
    CV2_LAUNCH_CHECK();
    return 0;
}

extern "C" int cv2_flow_inference_chunk(cv2_flow* h, const cv2_flow_utt* utts, const cv2_flow_cache_ref* refs, int32_t U, int32_t finalize,
                                        void* stream) {
    CV2_CHECK(h && utts && refs && U >= 1, "cv2_flow_inference_chunk: bad argument");
    CV2_CHECK(2 * U <= h->d.max_seqs, "cv2_flow_inference_chunk: %d utterances exceed max_seqs/2", U);
    hipStream_t s = (hipStream_t)stream;
    const int la = finalize ? 0 : 3;
    std::vector<int> lensT, lens2, lensN, lensNE;
    for (int u = 0; u < U; u++) {
        const int T = utts[u].n_tok - la, nc = refs[u].n_cached;
        CV2_CHECK(T >= 1, "cv2_flow_inference_chunk: utterance %d has no tokens", u);
        CV2_CHECK(utts[u].n_prompt_feat >= 0 && utts[u].n_prompt_feat <= 2 * T, "cv2_flow_inference_chunk: prompt_feat longer than the mel (%d > %d)", utts[u].n_prompt_feat, 2 * T);
        CV2_CHECK(refs[u].cache && refs[u].cache_frames >= 64 && refs[u].cache_frames % 64 == 0, "cv2_flow_inference_chunk: utterance %d: cache of %d frames (need a multiple of 64)", u, refs[u].cache_frames);
        CV2_CHECK(nc >= 0 && nc % 50 == 0 && nc < 2 * T, "cv2_flow_inference_chunk: utterance %d: %d cached frames must be whole chunks of 50 and fewer than the %d frames of this call", u, nc, 2 * T);
        CV2_CHECK(2 * T <= refs[u].cache_frames, "cv2_flow_inference_chunk: utterance %d: %d frames exceed the cache capacity %d", u, 2 * T, refs[u].cache_frames);
        CV2_CHECK(finalize || (2 * T) % 50 == 0, "cv2_flow_inference_chunk: utterance %d: a non-final call must end on a chunk boundary (%d frames)", u, 2 * T);
        lensT.push_back(T); lens2.push_back(2 * T); lensN.push_back(2 * T - nc);
    }
    // the encoder too runs over the new tokens only (its keys / values and conv tails of the prefix are in the cache) unless CV2_ENC_CACHE=0
    static const bool enc_cache = !(getenv("CV2_ENC_CACHE") && getenv("CV2_ENC_CACHE")[0] == '0');
    std::vector<int> lensTn;
    for (int u = 0; u < U; u++) lensTn.push_back(lensT[u] - refs[u].n_cached / 2);
    Layout LT = enc_cache ? make_layout(lensTn, INC_LEAD) : make_layout(lensT);
    Layout L2 = enc_cache ? make_layout(lensN, INC_LEAD) : make_layout(lens2);
    Layout LA = LT;
    for (int u = 0; u < U; u++) LA.len[u] += la;
    Layout LI = make_layout(lensN, INC_LEAD);
    lensNE = lensN; lensNE.insert(lensNE.end(), lensN.begin(), lensN.end());
    Layout LIE = make_layout(lensNE, INC_LEAD);
    if (check_rows(h, L2, "cv2_flow_inference_chunk") || check_rows(h, LIE, "cv2_flow_inference_chunk")) return -1;
    const int RI = LI.rows, twin = RI - INC_LEAD;
    if (upload_layout(h, LA, 0, s) || upload_layout(h, LT, 1, s) || upload_layout(h, L2, 2, s) || upload_layout(h, LI, 5, s) ||
        upload_layout(h, LIE, 6, s)) return -1;
    const size_t per = itab_per(h->R, h->d.max_seqs);
    CV2_CHECK((size_t)12 * U + 16 <= per, "cv2_flow_inference_chunk: int table overflow");
    // pointer tables: [emb U][prompt U][mel_out U][tokens U][kv 2U][tails 2U]; int tables: [n_prompt U][skip U][mu_start U][pos0 2U][frames 2U][gen 2U]
    std::vector<const void*> ptrs(8 * U);
    std::vector<int> ints(9 * U);
    int maxn2 = 1;
    for (int u = 0; u < U; u++) {
        ptrs[u] = utts[u].embedding; ptrs[U + u] = utts[u].prompt_feat; ptrs[2 * U + u] = utts[u].mel_out;
        ptrs[3 * U + u] = utts[u].tokens + (enc_cache ? refs[u].n_cached / 2 : 0);       // (the embedding kernel starts at the first new token)
        const long fr = refs[u].cache_frames;
        uint16_t* base = (uint16_t*)refs[u].cache;
        uint16_t* tails = base + inc_kv_elems(h, fr);
        ptrs[4 * U + u] = base; ptrs[5 * U + u] = base + inc_kv_elems(h, fr) / 2;
        ptrs[6 * U + u] = tails; ptrs[7 * U + u] = tails + inc_tail_elems(h) / 2;
        const int nc = refs[u].n_cached, skip = std::max(utts[u].n_prompt_feat - nc, 0);
        ints[u] = utts[u].n_prompt_feat; ints[U + u] = skip; ints[2 * U + u] = enc_cache ? L2.start[u] - nc : L2.start[u];      // mu row of frame pt: this + pt
        ints[3 * U + u] = ints[4 * U + u] = nc;
        ints[5 * U + u] = ints[6 * U + u] = (int)fr;
        ints[7 * U + u] = ints[8 * U + u] = refs[u].gen;
        maxn2 = std::max(maxn2, lensN[u] - skip);
    }
    CV2_HIP(hipMemcpyAsync(h->ptab, ptrs.data(), ptrs.size() * sizeof(void*), hipMemcpyHostToDevice, s));
    int* ibase = h->itab + (size_t)7 * per;
    CV2_HIP(hipMemcpyAsync(ibase, ints.data(), ints.size() * sizeof(int), hipMemcpyHostToDevice, s));
    CV2_HIP(hipStreamSynchronize(s));
    const void* const* dp = (const void* const*)h->ptab;
    IncTabs tabs{(uint16_t* const*)(dp + 4 * U), ibase + 5 * U, ibase + 3 * U, (uint16_t* const*)(dp + 6 * U), ibase + 7 * U};
    // encoder: over the new tokens with the per-stream cache (round 4), or over the whole prefix (its chunk masks make the rows of
    // finished chunks reproduce exactly)
    SpkArgs sp{(const float* const*)dp, h->w.spk_w, h->w.spk_b, h->spk};
    hipLaunchKernelGGL(k_spk, dim3(U), dim3(128), 0, s, sp);
    EmbedPtrArgs ea{(const int* const*)(dp + 3 * U), h->w.input_embedding, LA.tab(), LA.rows, GB(h->e_a, 512)};
    hipLaunchKernelGGL(k_embed_tokens_ptr, dim3((LA.rows + 1) / 2), dim3(256), 0, s, ea);
    EncInc einc;
    for (int u = 0; enc_cache && u < U; u++) {
        einc.base.push_back((uint16_t*)refs[u].cache + enc_part_off(h, refs[u].cache_frames));
        einc.frames.push_back(refs[u].cache_frames); einc.n0_tok.push_back(refs[u].n_cached / 2); einc.gen.push_back(refs[u].gen);
    }
    if (encoder_core(h, LA, LT, L2, 1, nullptr, s, enc_cache ? &einc : nullptr)) return -1;
    {
        GemmArgs a = gemm_args(GB(h->e_ln, 512), 512, 0, h->w.enc_proj.w, L2.rows, 128, 512);
        a.bias = h->w.enc_proj.b; a.out_f32 = h->mu; a.ldo = 80; a.n_store = 80; a.seq = L2.tab(); a.mask = 1;
        if (gemm_launch_cfg(a, 0, 1, true, s)) return -1;
    }
    // Euler loop over the new frames only
    EstCtx c{h, &LIE, nullptr, 50, s};
    c.inc = &tabs;
    for (int st = 0; st <= h->d.n_timesteps; st++) {
        PackArgs p{h->xs, h->mu, h->spk, (const float* const*)(dp + U), ibase, st == 0 ? h->w.rand_noise : nullptr,
                   st == 0 ? nullptr : h->vf, st == 0 ? 0.f : h->dt_host[st - 1], h->d.cfg_rate, LI.tab(), RI, U, GB(h->a0, 320),
                   ibase + 3 * U, ibase + 2 * U, twin, INC_LEAD};
        hipLaunchKernelGGL(k_euler_pack, dim3(((long)RI * 20 + 255) / 256), dim3(256), 0, s, p);
        if (st == h->d.n_timesteps) break;                       // last update: no estimator call follows
        c.temb = h->temb_tab + (size_t)st * 14 * 256;
        c.step = st;
        if (estimator_core(c)) return -1;
    }
    MelOutArgs mo{h->xs, (float* const*)(dp + 2 * U), ibase + U, LI.tab()};
    hipLaunchKernelGGL(k_mel_out, dim3(((long)maxn2 * 80 + 255) / 256, U), dim3(256), 0, s, mo);
    CV2_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ test hook: the epilogue activations as the GEMM kernels evaluate them
__global__ void k_dbg_act(const float* x, float* out, int n, int act, float slope) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = act_apply(x[i], act, slope);
}
extern "C" int cv2_dbg_act(const float* x, float* out, int64_t n, int32_t act, float slope, void* stream) {
    CV2_CHECK(x && out && n > 0 && n < (1ll << 30), "cv2_dbg_act: bad argument");
    hipLaunchKernelGGL(k_dbg_act, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, out, (int)n, (int)act, slope);
    CV2_LAUNCH_CHECK();
    return 0;
}

