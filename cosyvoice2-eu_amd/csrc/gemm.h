// bf16 MFMA GEMM with an LDS-staged row epilogue: the MFMA workhorse of stage 2 (flow).
//
//   out[m][n] = epilogue( sum_k A[m + a_row_off][k] * W[n][k] )          A bf16 row-major, W bf16
//
// Roofline: MFMA (bf16 16x16x32, fp32 accumulate).  Tile BM x BN x 64; both operand tiles are staged by LDS-DMA
// (global_load_lds, 16 B per lane) into two LDS stages as 16-row x 32-k sub-tiles of 1 KiB in MFMA operand order:
//   * activations (row-major, any leading dimension; a causal Conv1d over time-major rows is this GEMM with
//     lda = C and K = taps*C, the tap window being contiguous memory) are gathered with a per-lane SOURCE address
//     and the st_16x32 XOR swizzle (conflict-free ds_read_b128 fragment reads);
//   * weights are pre-packed in HBM in exactly that order (include/cv2_amd.h), so a stage is a linear stream.
// The prefetch of K-step k+1 stays in flight across the barrier (counted vmcnt + raw s_barrier).
// Orientation: the MFMA "A" operand is the weight fragment, "B" the activation fragment, so a lane ends with 4
// consecutive output features of one row; the accumulator tile is then staged through LDS (re-using the operand
// stages) and finished row-wise: bias, LayerNorm over the row, activation, per-sequence vector add, sequence
// mask, fp32 residual, fp32 / bf16 stores, a second LayerNorm for the next GEMM's operand, or a transposed
// bf16 store (V^T for attention).  Rows live in the packed ragged layout described in flow.hip.
#pragma once
#include "common.h"

enum { ACT_NONE = 0, ACT_GELU = 1, ACT_SILU = 2, ACT_MISH = 3, ACT_LRELU = 4 };

struct SeqTable {
    const int* tile_seq;    // [rows / 64]: sequence id of every 64-row tile, -1 = padding tile
    const int* seq_start;   // [S] first row (multiple of 128)
    const int* seq_len;     // [S] valid rows
    const int4* tile_info;  // [rows / 64]: {sequence id or -1, its first row, its valid rows, 0}: the three lookups above in one load
};

struct GemmArgs {
    const uint16_t* A; long lda; long a_row_off; long a_bstride;   // bf16 activations; row m reads A[(m + a_row_off) * lda ..]
    const uint16_t* A_lo;                                           // SPLITA kernels: low half of the operand split, same layout
    const uint16_t* W; long ldw; long w_bstride;                    // packed (WPACKED) or row-major [N][ldw]
    int M, N, K;                                                    // M % BM == 0 (padded rows), K % 64 == 0
    SeqTable seq;                                                   // tile_seq == null: every row m < M_valid is valid
    int M_valid;
    // epilogue, in this order
    const float* bias;                                              // [N]
    const float* ln1_g; const float* ln1_b; float ln1_eps;          // LayerNorm over the row (BN == N)
    int act; float act_slope;
    const float* rowadd; int rowadd_ld;                             // + rowadd[seq][n]
    const float* res; long ldres;                                   // + res[m][n] (fp32)
    int mask;                                                       // rows beyond their sequence -> 0
    float out_scale;                                                // * out_scale (applied to acc before bias when != 1)
    float* out_f32; long ldo; long o_bstride;
    uint16_t* out_bf16; long ldo16; long o16_bstride;
    const float* ln2_g; const float* ln2_b; float ln2_eps; float ln2_scale; uint16_t* out_ln2; long ldo_ln2;
    uint16_t* vt; long vt_ld; int vt_n0;                            // features n >= vt_n0: vt[(n - vt_n0) * vt_ld + m]
    int n_store;                                                    // only features n < n_store are written (N padded up)
    // cached streaming (flow: cv2_flow_inference_chunk): the QKV projection also files the call's keys / values in each sequence's cache
    // slot -- K rows [frames][512] for features [kvc_k0, vt_n0), V^T [512][frames] for features >= vt_n0 -- at frames pos0[s] + t
    // (what a separate k_kv_append launch per transformer block did: 7.5 us x 560 per chunk round)
    uint16_t* const* kvc; const int* kvc_frames; const int* kvc_pos0; long kvc_slot; int kvc_k0;
    // split-K as a batch (the one-prompt prefill's down projection: blockIdx.z = K slice, a_bstride = w.r.t. columns, o_bstride = one
    // partial matrix): the packed W's row-tile stride in k blocks stays that of the WHOLE K
    int w_ks;                                                       // 0: K / 32
};

__device__ __forceinline__ float act_apply(float v, int act, float slope) {
    switch (act) {
        case ACT_GELU: {                                   // exact-erf GELU; erf by Abramowitz-Stegun 7.1.26 (|err| < 7e-7 in fp32, one exp + one rcp)
            const float z = v * 0.70710678118654752f, az = fabsf(z);
            const float t = __builtin_amdgcn_rcpf(1.f + 0.3275911f * az);      // v_rcp_f32 (1 ulp): __fdividef / a plain division compile to the ten-instruction IEEE
                                                                               // sequence (v_div_scale x 2, v_rcp, 4 fma, v_div_fmas, v_div_fixup) -- a third of this function,
                                                                               // and the feed-forward's GELU is the largest VALU item of the batch tail kernel
            const float poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
            const float e = 1.f - poly * __expf(-az * az);
            return 0.5f * v * (1.f + copysignf(e, z));
        }
        case ACT_SILU: return v * __builtin_amdgcn_rcpf(1.f + __expf(-v));
        case ACT_MISH: {                                   // x * tanh(softplus(x)); tanh(log(1 + e^x)) = n / (n + 2), n = e^x (e^x + 2)
            const float e = __expf(fminf(v, 20.f));        // torch's softplus switches to the identity above 20: the ratio is 1 there
            const float nn = e * (e + 2.f);
            return v * (nn * __builtin_amdgcn_rcpf(nn + 2.f));
        }
        case ACT_LRELU: return v > 0.f ? v : v * slope;
        default: return v;
    }
}

__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    const bf16x2 h = __builtin_convertvector((f32x2){a, b}, bf16x2);
    return __builtin_bit_cast(uint32_t, h);
}

// LDS byte offset of (row, 16-B chunk) inside a 16 x 32 bf16 sub-tile, st_16x32 swizzle
__device__ __forceinline__ int subtile_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 3) << 1)) << 4); }

// s_waitcnt vmcnt(N) with a compile-time N (the immediate must be a literal)
template <int N>
__device__ __forceinline__ void vmcnt_wait() {
    static_assert(N >= 0 && N <= 24, "add the literal below");
    if (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else if (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (N == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
    else if (N == 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    else if (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // conservative for counts without a literal
}

// sum over aligned groups of LPR lanes, result in every lane of the group (DPP row operations for 64 / 32 lanes, see common.h)
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
    if (LPR == 64) return wave_sum(v);
    if (LPR == 32) {
        v += dpp_mov_f32<0x111, 0xf>(0.f, v);
        v += dpp_mov_f32<0x112, 0xf>(0.f, v);
        v += dpp_mov_f32<0x114, 0xf>(0.f, v);
        v += dpp_mov_f32<0x118, 0xf>(0.f, v);
        v += dpp_mov_f32<0x142, 0xa>(0.f, v);                      // lanes 31 / 63 now hold the two half-wave totals
        const float lo = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 31));
        const float hi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
        return (threadIdx.x & 32) ? hi : lo;
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// XCD-aware tile order (cdna_hip_programming.md T1, bijective form): consecutive block ids are dealt round the 8 XCDs, so the N
// tiles of one row tile would land in 8 different L2s and its A rows would cross the fabric up to 8 times.  Blocks with equal
// id % 8 take one contiguous share of the tile list instead (n fastest): an A row tile is fetched by one L2.  A speed choice only.
__device__ __forceinline__ void xcd_tile(int& bx, int& by) {
    const int gx = gridDim.x, T = gx * gridDim.y;
    bx = blockIdx.x; by = blockIdx.y;
    if (gridDim.z != 1 || gx == 1 || T < 16) return;
#ifdef CV2_NO_XCD        // (A/B builds)
    return;
#endif
    const int L = bx + by * gx, x = L & 7, q = T >> 3, r = T & 7;
    const int Lp = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (L >> 3);
    bx = Lp % gx; by = Lp / gx;
}

// EPI = 1: the row epilogue of a GEMM that only stores bf16, rows beyond their sequence as zeros (no bias / LayerNorm / activation / residual / row vector / fp32 copy /
// cache append -- the estimator's QKV projection): the general epilogue below is ~1 000 executed instructions per wave and tile with all
// its run-time switches (phase stamps at 32 utterances: 7 560 of a block's 17 600 cycles for the 128 x 128 x 256 QKV tile); this one is
// 8 x {LDS read, two conversions, store}.
template <int BM, int BN, int WM, int WN, bool WPACKED, bool SPLITA = false, int NSTAGE = 2, int EPI = 0>
__global__ __launch_bounds__(WM * WN * 64) void k_gemm(GemmArgs a) {
    constexpr int NW = WM * WN, NT_ = NW * 64;
    constexpr int TM = BM / WM, TN = BN / WN, MT = TM / 16, NT = TN / 16;
    // 1 KiB pieces per stage: A (hi) [, A lo: the operand split x = hi + lo keeps ~16 mantissa bits through two bf16 MFMAs], W
    constexpr int NA = BM / 16 * 2, NAP = SPLITA ? 2 * NA : NA, NB = BN / 16 * 2, NP = NAP + NB;
    constexpr int STAGE = NP * 1024;
    constexpr int LDC = BN + 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    int bx_, by_;
    xcd_tile(bx_, by_);
    const int m0 = by_ * BM, n0 = bx_ * BN;
    const uint16_t* A = a.A + (size_t)blockIdx.z * a.a_bstride;
    const uint16_t* W = a.W + (size_t)blockIdx.z * a.w_bstride;
    const int KS = a.w_ks > 0 ? a.w_ks : a.K / 32;                            // 32-wide k blocks in the packed W
    const int nk = a.K / 64;
    const uint16_t* A_lo = SPLITA ? a.A_lo + (size_t)blockIdx.z * a.a_bstride : nullptr;

    // per-lane source coordinates of an LDS-DMA piece (lane writes LDS byte lane*16 of the sub-tile)
    const int srow = lane >> 2, schunk = (lane & 3) ^ ((lane >> 5) << 1);

    // Pieces are dealt round-robin to the waves; every wave issues the same count PER_WAVE (the counted vmcnt below relies on
    // it), a wave whose last slot falls beyond NP re-issues the last piece (same bytes to the same LDS address: harmless).
    constexpr int PER_WAVE = (NP + NW - 1) / NW;
    auto issue = [&](int kt, int buf) {
        char* base = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < PER_WAVE; i++) {
            int p = wave + i * NW;
            if (NP % NW != 0) p = p < NP ? p : NP - 1;
            const void* src;
            if (p < NAP) {                                // wave-uniform
                const int pp = SPLITA ? p % NA : p, sub = pp >> 1, ks = pp & 1;
                const uint16_t* Ab = (SPLITA && p >= NA) ? A_lo : A;
                src = Ab + ((long)(m0 + sub * 16 + srow) + a.a_row_off) * a.lda + kt * 64 + ks * 32 + schunk * 8;
            } else {
                const int q = p - NAP, sub = q >> 1, ks = q & 1;
                if (WPACKED)
                    src = W + (((size_t)(n0 / 16 + sub) * KS + kt * 2 + ks) * 64 + lane) * 8;
                else
                    src = W + (long)(n0 + sub * 16 + srow) * a.ldw + kt * 64 + ks * 32 + schunk * 8;
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(base + p * 1024), 16, 0, 0);
        }
    };

    f32x4 acc[NT][MT];
#pragma unroll
    for (int j = 0; j < NT; j++)
#pragma unroll
        for (int i = 0; i < MT; i++) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int a_off = subtile_off(lane & 15, lane >> 4);
    const int w_off = WPACKED ? lane * 16 : a_off;

    SK_STAMP_DECL;
    SK_STAMP(0);
    issue(0, 0);
    // epilogue operands that do not depend on the row: fetched now, consumed after the K loop
    constexpr int LPR = BN / 4, RPI = NT_ / LPR, NIT = BM / RPI;   // lanes per row, rows per epilogue iteration, iterations
    static_assert(LPR <= 64, "BN <= 256");
    const int rsub = tid / LPR, cl = tid % LPR;
    const int n = n0 + cl * 4;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    const f32x4 ep_bias = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + n) : z4;
    const f32x4 ep_g1 = a.ln1_g ? *reinterpret_cast<const f32x4*>(a.ln1_g + n) : z4;
    const f32x4 ep_b1 = a.ln1_g ? *reinterpret_cast<const f32x4*>(a.ln1_b + n) : z4;
    const f32x4 ep_g2 = a.ln2_g ? *reinterpret_cast<const f32x4*>(a.ln2_g + n) : z4;
    const f32x4 ep_b2 = a.ln2_g ? *reinterpret_cast<const f32x4*>(a.ln2_b + n) : z4;
    constexpr int NTL = (BM + 63) / 64;                            // 64-row sequence tiles touched by this block
    // one record per tile (sequence id, start, length): a single load here instead of the chain tile_seq -> seq_start / seq_len
    // -> rowadd, which held the K loop back by two memory round trips; rowadd (needs the id) is fetched inside the loop
    int ep_start[NTL], ep_len[NTL], ep_sq[NTL];
    f32x4 ep_radd[NTL];
#pragma unroll
    for (int t = 0; t < NTL; t++) {
        ep_start[t] = 0; ep_len[t] = a.M_valid; ep_sq[t] = 0; ep_radd[t] = z4;
        if (a.seq.tile_seq) {                           // padding tiles carry {-1, 0, 0}: branch-free, so the record stays ONE 16-byte load
            // block-uniform address, read-only table: a scalar load (constant address space), which nothing has to wait for until
            // the values are used; as a vector load hipcc waits for it (and the stage-0 DMA before it) right here
            typedef int i32x4_t __attribute__((ext_vector_type(4)));
            typedef const __attribute__((address_space(4))) i32x4_t k_i32x4;
            const i32x4_t ti = *reinterpret_cast<k_i32x4*>(reinterpret_cast<uintptr_t>(a.seq.tile_info + (m0 >> 6) + t));
            ep_sq[t] = max(ti.x, 0); ep_start[t] = ti.y; ep_len[t] = ti.z;
        }
    }
    // first residual row of this lane: the 16-wave (small-grid, latency-bound) configurations request it now, so that it has long
    // arrived when the epilogue starts; the 8-wave configurations cannot afford its 4 registers across the K loop (they sit at the
    // 128-VGPR step that decides whether two blocks share a CU) and request it after the loop
    constexpr bool EARLY_RES = NW >= 16;
    const bool use_res = a.res && !(a.vt && n0 >= a.vt_n0);
    f32x4 res_next = z4;
    if (EARLY_RES && use_res) res_next = *reinterpret_cast<const f32x4*>(a.res + (size_t)(m0 + rsub) * a.ldres + n);
#pragma unroll
    for (int st = 1; st < NSTAGE - 1; st++) if (st < nk) issue(st, st);
    SK_STAMP(1);                                         // prologue loads issued
    for (int kt = 0, buf = 0; kt < nk; kt++, buf = (buf + 1 == NSTAGE ? 0 : buf + 1)) {
        // keep NSTAGE - 1 K-steps in flight: wait only until stage kt has landed (counted vmcnt, loads retire in order)
        const int ahead = NSTAGE - 1;
        if (kt + ahead < nk) issue(kt + ahead, (buf + ahead) % NSTAGE);
        const int inflight = (nk - 1 - kt) < ahead ? (nk - 1 - kt) : ahead;      // younger stages that may stay outstanding
        if (NSTAGE > 3 && inflight == 3) vmcnt_wait<3 * PER_WAVE>();
        else if (NSTAGE > 2 && inflight == 2) vmcnt_wait<2 * PER_WAVE>();
        else if (inflight == 1) vmcnt_wait<PER_WAVE>();
        else vmcnt_wait<0>();
        __builtin_amdgcn_s_barrier();                    // stage kt has landed for every wave
        if (kt == 0) {
            SK_STAMP(2);                                 // first stage landed (and every older prologue load)
            if (EARLY_RES && a.rowadd) {
#pragma unroll
                for (int t = 0; t < NTL; t++) ep_radd[t] = *reinterpret_cast<const f32x4*>(a.rowadd + (size_t)ep_sq[t] * a.rowadd_ld + n);
            }
        }
        const char* base = smem + buf * STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            bf16x8 af[MT], al[SPLITA ? MT : 1], wf[NT];
#pragma unroll
            for (int i = 0; i < MT; i++) {
                af[i] = *reinterpret_cast<const bf16x8*>(base + ((wm * MT + i) * 2 + ks) * 1024 + a_off);
                if (SPLITA) al[i] = *reinterpret_cast<const bf16x8*>(base + (NA + (wm * MT + i) * 2 + ks) * 1024 + a_off);
            }
#pragma unroll
            for (int j = 0; j < NT; j++)
                wf[j] = *reinterpret_cast<const bf16x8*>(base + (NAP + (wn * NT + j) * 2 + ks) * 1024 + w_off);
#pragma unroll
            for (int j = 0; j < NT; j++)
#pragma unroll
                for (int i = 0; i < MT; i++) {
                    acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[j][i], 0, 0, 0);
                    if (SPLITA) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], al[i], acc[j][i], 0, 0, 0);
                }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                    // everyone is done reading this stage
    }

    SK_STAMP(3);                                         // K loop done
    // ---- accumulators -> LDS C tile [BM][LDC] ----
    float* C = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int j = 0; j < NT; j++)
#pragma unroll
        for (int i = 0; i < MT; i++) {
            const int m = wm * TM + i * 16 + (lane & 15);
            const int n = wn * TN + j * 16 + 4 * (lane >> 4);
            *reinterpret_cast<f32x4*>(&C[m * LDC + n]) = acc[j][i] * a.out_scale;
        }
    if (!EARLY_RES && use_res) res_next = *reinterpret_cast<const f32x4*>(a.res + (size_t)(m0 + rsub) * a.ldres + n);
    if (!EARLY_RES && a.rowadd) {
#pragma unroll
        for (int t = 0; t < NTL; t++) ep_radd[t] = *reinterpret_cast<const f32x4*>(a.rowadd + (size_t)ep_sq[t] * a.rowadd_ld + n);
    }
    __syncthreads();
    SK_STAMP(4);                                         // C tile staged

    float* out_f32 = a.out_f32 ? a.out_f32 + (size_t)blockIdx.z * a.o_bstride : nullptr;
    uint16_t* out_bf16 = a.out_bf16 ? a.out_bf16 + (size_t)blockIdx.z * a.o16_bstride : nullptr;

    if (a.vt && n0 >= a.vt_n0) {
        // transposed bf16 store: 8 consecutive rows of one feature = 16 B
        for (int it = tid; it < BN * (BM / 8); it += NT_) {
            const int n = it % BN, rg = it / BN;
            const int mrow = m0 + rg * 8;
            int s = 0, start = 0, len = a.M_valid;
            if (a.seq.tile_seq) { s = a.seq.tile_seq[mrow >> 6]; if (s >= 0) { start = a.seq.seq_start[s]; len = a.seq.seq_len[s]; } }
            const float b = a.bias ? a.bias[n0 + n] : 0.f;
            uint32_t pk[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                float v0 = C[(rg * 8 + 2 * r) * LDC + n] + b, v1 = C[(rg * 8 + 2 * r + 1) * LDC + n] + b;
                if (a.mask) {
                    if (s < 0 || mrow + 2 * r - start >= len) v0 = 0.f;
                    if (s < 0 || mrow + 2 * r + 1 - start >= len) v1 = 0.f;
                }
                pk[r] = pack_bf16x2(v0, v1);
            }
            *reinterpret_cast<uint4*>(a.vt + (size_t)(n0 + n - a.vt_n0) * a.vt_ld + mrow) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
            if (a.kvc && s >= 0) {                         // the same 8 frames of this feature -> the sequence's V^T cache (pos0 and t are even: 4-byte pairs)
                const long fr = a.kvc_frames[s];
                const int t = mrow - start;
                uint16_t* d = a.kvc[s] + a.kvc_slot * fr * 1024 + fr * 512 + (size_t)(n0 + n - a.vt_n0) * fr + a.kvc_pos0[s] + t;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    if (t + 2 * r + 1 < len) *reinterpret_cast<uint32_t*>(d + 2 * r) = pk[r];
                    else if (t + 2 * r < len) d[2 * r] = (uint16_t)(pk[r] & 0xffffu);
                }
            }
        }
        return;
    }

    if (EPI == 1) {
        const bool wr = n < a.n_store;
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int ml = it * RPI + rsub;
            const int tl = (BM > 64 && ml >= 64) ? 1 : 0;
            const bool valid = !a.mask || (m0 + ml - ep_start[BM > 64 ? tl : 0]) < ep_len[BM > 64 ? tl : 0];
            f32x4 v = *reinterpret_cast<const f32x4*>(&C[ml * LDC + cl * 4]);
            if (!valid) v = z4;
            if (wr) *reinterpret_cast<uint2*>(out_bf16 + (size_t)(m0 + ml) * a.ldo16 + n) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
        }
        SK_STAMP(5);
        SK_STAMP_FLUSH_RING(((unsigned long long)a.K << 32) | (unsigned)a.N,
                            ((unsigned long long)BM << 48) | ((unsigned long long)BN << 32) | (unsigned)(gridDim.x * gridDim.y * gridDim.z));
        return;
    }
    // ---- row-wise epilogue: LPR lanes per row, 4 consecutive features per lane ----
    // Everything that does not depend on the row (this lane's 4 columns of bias / LayerNorm parameters, the sequence
    // records of the block's 64-row tiles) was loaded into registers BEFORE the K loop (ep_* below), and the residual rows
    // are fetched in one batch, so the loop itself only touches LDS and issues stores.
#pragma unroll 1
    for (int it = 0; it < NIT; it++) {
        const int ml = it * RPI + rsub, m = m0 + ml;
        const f32x4 res_cur = res_next;                  // residual row fetched one iteration ahead
        if (use_res && it + 1 < NIT) res_next = *reinterpret_cast<const f32x4*>(a.res + (size_t)(m + RPI) * a.ldres + n);
        const int tl = BM > 64 ? (ml >> 6) : 0;
        const bool valid = tl == 0 ? (m - ep_start[0]) < ep_len[0] : (m - ep_start[BM > 64 ? 1 : 0]) < ep_len[BM > 64 ? 1 : 0];
        f32x4 v = *reinterpret_cast<const f32x4*>(&C[ml * LDC + cl * 4]);
        v += ep_bias;
        if (a.ln1_g) {
            const float mean = group_sum<LPR>(v[0] + v[1] + v[2] + v[3]) * (1.f / BN);
            const f32x4 d = v - mean;
            const float var = group_sum<LPR>(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.f / BN);
            const float rstd = rsqrtf(var + a.ln1_eps);
            v = d * rstd * ep_g1 + ep_b1;
        }
        if (a.act != ACT_NONE) {
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = act_apply(v[e], a.act, a.act_slope);
        }
        if (a.rowadd) v += tl == 0 ? ep_radd[0] : ep_radd[BM > 64 ? 1 : 0];
        if (use_res) v += res_cur;
        if (a.mask && !valid) v = (f32x4){0.f, 0.f, 0.f, 0.f};
        const bool wr = n < a.n_store;
        if (out_f32 && wr) *reinterpret_cast<f32x4*>(out_f32 + (size_t)m * a.ldo + n) = v;
        if (out_bf16 && wr)
            *reinterpret_cast<uint2*>(out_bf16 + (size_t)m * a.ldo16 + n) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
        if (a.kvc && n >= a.kvc_k0 && wr && valid && a.seq.tile_seq) {       // this row's key channels -> the sequence's K cache
            const int s = a.seq.tile_seq[m >> 6];
            const long fr = a.kvc_frames[s];
            uint16_t* d = a.kvc[s] + a.kvc_slot * fr * 1024 + (size_t)(a.kvc_pos0[s] + m - a.seq.seq_start[s]) * 512 + (n - a.kvc_k0);
            *reinterpret_cast<uint2*>(d) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
        }
        if (a.ln2_g) {
            const float mean = group_sum<LPR>(v[0] + v[1] + v[2] + v[3]) * (1.f / BN);
            const f32x4 d = v - mean;
            const float var = group_sum<LPR>(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.f / BN);
            const float rstd = rsqrtf(var + a.ln2_eps);
            f32x4 y = (d * rstd * ep_g2 + ep_b2) * a.ln2_scale;
            if (a.mask && !valid) y = (f32x4){0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<uint2*>(a.out_ln2 + (size_t)m * a.ldo_ln2 + n) = make_uint2(pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3]));
        }
    }
    SK_STAMP(5);                                         // row epilogue done
    SK_STAMP_FLUSH_RING(((unsigned long long)a.K << 32) | (unsigned)a.N,
                        ((unsigned long long)BM << 48) | ((unsigned long long)BN << 32) | (unsigned)(gridDim.x * gridDim.y * gridDim.z));
}

// ------------------------------------------------------------------ row-panel GEMM (few rows: one utterance)
// Block = 16 rows x 256 columns, 16 waves, wave w owns columns [16 w, 16 w + 16).  Phase stamps of k_gemm<32, 256> on M = 2048
// showed the K loop bound by LDS traffic (every wave re-reads A and W fragments: 96 KB per 64-wide step against 36 KB staged)
// and two block barriers per step, and the row epilogue bound by VALU work on only 64 CUs.  Here
//   * the packed W fragments go straight from global memory to registers (they ARE the MFMA operand: one 1 KiB wave-load each,
//     every byte of W crosses the CU's vector-memory path once), a ring of CH fragments per wave in flight;
//   * the block's whole A panel [16][K] is staged in LDS once by DMA: ONE barrier before the K loop and none inside;
//   * twice as many blocks share the row epilogue (LayerNorm needs the whole row, so the N = 256 columns stay in one block).
// Same arguments and epilogue semantics as k_gemm (no transposed-V path, packed weights, K % (32 CH) == 0).
// KS_T = K / 32 at compile time: the K loop is straight-line code, so the waits hipcc derives for the ring are exact counted
// ones (around a run-time loop it falls back to vmcnt(0) at the loop head, which serialises load and use chunk by chunk).
template <int CH, int KS_T>
__global__ __launch_bounds__(1024) void k_gemm_panel(GemmArgs a) {
    constexpr int BM = 16, BN = 256, NW = 16, LDC = BN + 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const uint16_t* A = a.A + (size_t)blockIdx.z * a.a_bstride;
    const uint16_t* W = a.W + (size_t)blockIdx.z * a.w_bstride;
    constexpr int KS = KS_T;
    SK_STAMP_DECL;
    SK_STAMP(0);
    const int rsub = tid >> 6, cl = lane;                // epilogue: wave = row, lane = 4 columns
    const int n = n0 + cl * 4;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    // A panel -> LDS: 1 KiB piece p = rows m0..m0+15, k [32 p, 32 p + 32), st_16x32 swizzle on the source side.  Requested first.
    {
        const int srow = lane >> 2, schunk = (lane & 3) ^ ((lane >> 5) << 1);
        const uint16_t* asrc = A + ((long)(m0 + srow) + a.a_row_off) * a.lda + schunk * 8;
        for (int p = wave; p < KS; p += NW)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc + p * 32),
                                             (__attribute__((address_space(3))) void*)(smem + p * 1024), 16, 0, 0);
    }
    // the five per-column epilogue vectors (bias, LayerNorm 1 / 2 scale and shift; 1 KiB each) go to LDS once per block (wave w
    // fetches vector w) instead of once per wave: as register loads they were 80 KB of the block's vector-memory traffic
    constexpr size_t PAR_OFF = (size_t)KS * 1024 > (size_t)BM * LDC * 4 ? (size_t)KS * 1024 : (size_t)BM * LDC * 4;
    float* par = reinterpret_cast<float*>(smem + PAR_OFF);
    if (wave < 5) {
        const float* src = wave == 0 ? a.bias : wave == 1 ? a.ln1_g : wave == 2 ? a.ln1_b : wave == 3 ? a.ln2_g : a.ln2_b;
        if (wave >= 1 && wave <= 2 && !a.ln1_g) src = nullptr;
        if (wave >= 3 && !a.ln2_g) src = nullptr;
        if (src)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + n0 + lane * 4),
                                             (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(par) + wave * 1024), 16, 0, 0);
    }
    const f32x4 ep_res = a.res ? *reinterpret_cast<const f32x4*>(a.res + (size_t)(m0 + rsub) * a.ldres + n) : z4;
    // every wave's panel / vector requests are queued before any weight request (instruction streams only: nothing is waited for)
    __builtin_amdgcn_s_barrier();
    const s16x8* wp = reinterpret_cast<const s16x8*>(W) + ((size_t)(n0 / 16 + wave) * KS) * 64 + lane;
    // ring of CH fragments: slot i is refilled with fragment kb + CH as soon as fragment kb has been consumed, so CH wave-loads
    // (CH KiB per wave) stay in flight; the last CH fragments are consumed without refills (fixed load counts on both paths)
    s16x8 wr[CH];
#pragma unroll
    for (int i = 0; i < CH; i++) wr[i] = wp[(size_t)i * 64];     // plain loads: every block of the launch reads the same W, it must stay in L2
    int ep_start = 0, ep_len = a.M_valid, ep_sq = 0;
    if (a.seq.tile_seq) {                                // scalar load, requested after everything above: nothing queues behind its miss
        typedef int i32x4_t __attribute__((ext_vector_type(4)));
        typedef const __attribute__((address_space(4))) i32x4_t k_i32x4;
        const i32x4_t ti = *reinterpret_cast<k_i32x4*>(reinterpret_cast<uintptr_t>(a.seq.tile_info + (m0 >> 6)));
        ep_sq = max(ti.x, 0); ep_start = ti.y; ep_len = ti.z;
    }
    SK_STAMP(1);
    vmcnt_wait<CH>();                                    // everything older than the CH weight loads: the panel, the vectors, the residual
    __builtin_amdgcn_s_barrier();                        // (raw: __syncthreads would also wait for the weight loads)
    SK_STAMP(2);                                         // A panel staged
    const f32x4 ep_radd = a.rowadd ? *reinterpret_cast<const f32x4*>(a.rowadd + (size_t)ep_sq * a.rowadd_ld + n) : z4;
    const char* ap = smem + subtile_off(lane & 15, lane >> 4);
    f32x4 acc = z4;
#pragma unroll
    for (int kb = 0; kb < KS; kb++) {
        const bf16x8 af = *reinterpret_cast<const bf16x8*>(ap + (size_t)kb * 1024);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wr[kb % CH]), af, acc, 0, 0, 0);
        if (kb + CH < KS) wr[kb % CH] = wp[(size_t)(kb + CH) * 64];
    }
    SK_STAMP(3);                                         // K loop done
    __syncthreads();                                     // every wave is done reading the panel: its LDS becomes the C tile
    float* C = reinterpret_cast<float*>(smem);
    *reinterpret_cast<f32x4*>(&C[(lane & 15) * LDC + wave * 16 + 4 * (lane >> 4)]) = acc * a.out_scale;
    __syncthreads();
    SK_STAMP(4);                                         // C tile staged
    // ---- row epilogue: one wave per row, 4 consecutive features per lane (the order of k_gemm's) ----
    {
        const int m = m0 + rsub;
        const bool valid = (m - ep_start) < ep_len;
        f32x4 v = *reinterpret_cast<const f32x4*>(&C[rsub * LDC + cl * 4]);
        const f32x4* pv = reinterpret_cast<const f32x4*>(par) + cl;          // [5][64] f32x4: bias, g1, b1, g2, b2
        if (a.bias) v += pv[0];
        const f32x4 ep_g1 = pv[64], ep_b1 = pv[128], ep_g2 = pv[192], ep_b2 = pv[256];   // (unused ones: stale LDS, never consumed)
        if (a.ln1_g) {
            const float mean = wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.f / BN);
            const f32x4 d = v - mean;
            const float var = wave_sum(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.f / BN);
            const float rstd = rsqrtf(var + a.ln1_eps);
            v = d * rstd * ep_g1 + ep_b1;
        }
        if (a.act != ACT_NONE) {
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = act_apply(v[e], a.act, a.act_slope);
        }
        if (a.rowadd) v += ep_radd;
        if (a.res) v += ep_res;
        if (a.mask && !valid) v = z4;
        const bool wr = n < a.n_store;
        float* out_f32 = a.out_f32 ? a.out_f32 + (size_t)blockIdx.z * a.o_bstride : nullptr;
        uint16_t* out_bf16 = a.out_bf16 ? a.out_bf16 + (size_t)blockIdx.z * a.o16_bstride : nullptr;
        if (out_f32 && wr) *reinterpret_cast<f32x4*>(out_f32 + (size_t)m * a.ldo + n) = v;
        if (out_bf16 && wr)
            *reinterpret_cast<uint2*>(out_bf16 + (size_t)m * a.ldo16 + n) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
        if (a.ln2_g) {
            const float mean = wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.f / BN);
            const f32x4 d = v - mean;
            const float var = wave_sum(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.f / BN);
            const float rstd = rsqrtf(var + a.ln2_eps);
            f32x4 y = (d * rstd * ep_g2 + ep_b2) * a.ln2_scale;
            if (a.mask && !valid) y = z4;
            *reinterpret_cast<uint2*>(a.out_ln2 + (size_t)m * a.ldo_ln2 + n) = make_uint2(pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3]));
        }
    }
    SK_STAMP(5);
    SK_STAMP_FLUSH_RING(((unsigned long long)a.K << 32) | (unsigned)a.N,
                        ((unsigned long long)BM << 48) | ((unsigned long long)BN << 32) | (unsigned)(gridDim.x * gridDim.y * gridDim.z));
}

// LDS hand-offs only: wait for this wave's LDS operations, then the block barrier.  __syncthreads() also waits for every outstanding
// global operation (s_waitcnt vmcnt(0)): weight fragments requested to fly through an epilogue would be waited for at the epilogue's first barrier,
// and a barrier behind a store phase would sit out the stores' completion.
#define TR2_BARRIER do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
// ------------------------------------------------------------------ transformer-block tail for few rows (one utterance)
// Everything of a BasicTransformerBlock after the attention is row-local: x += att Wo^T + bo; y = LN3(x); x += GELU(y W1^T + b1) W2^T
// + b2; next = LN(x).  One launch per 16-row panel chains the three GEMMs (K = 512, 256, 1024) with the weights streamed
// global -> registers, the operands of GEMM 2 and 3 written to LDS in MFMA operand layout by the previous epilogue, and the
// residual row kept in registers.  Replaces three launches (k_gemm_panel<K=512>, k_gemm<64,128> for FF1, k_gemm_panel<K=1024>), their
// prologues, the HBM round trip of the 1024-wide hidden activations and one of xf.
struct TailArgs {
    const uint16_t* att; long lda;                           // attention output [M][512] bf16
    const uint16_t* Wo; const float* bo;                     // packed [256][512]
    const float* g3; const float* b3; float eps3;            // norm3
    const uint16_t* W1; const float* b1;                     // packed [1024][256]
    const uint16_t* W2; const float* b2;                     // packed [256][1024]
    const float* gn; const float* bn; float epsn;            // next block's norm1 (null: last block of the group)
    float* xf;                                               // residual stream fp32 [M][256], read and (gn != null) rewritten
    uint16_t* out_ln; long ldo_ln;                           // gn != null: LN'd bf16 [M][256]
    uint16_t* out_x; long ldo_x;                             // gn == null: x as bf16
    SeqTable seq; int M_valid;
    float* part; int* ticket;                                // k_tail_panel<S > 1>: partial FF2 tiles [panels][S][16][256] and the panels' arrival counters (zero between launches)
    // k_tail_rows2<true>: the NEXT transformer block's QKV projection chained on (row-local like the rest: K = 256, N = 1536, no bias): Q / K
    // features [0, 1024) -> qk [M][1024] bf16, V features -> vt[(n - 1024) * vt_ld + m] (transposed for the attention), rows beyond their
    // sequence as zeros -- what k_gemm<.., EPI = 1> + its transposed-V path store; out_ln is not written then (nothing else reads it)
    const uint16_t* Wqkv; uint16_t* qk; uint16_t* vt; long vt_ld;
    int row0;                                                // k_tail_panel: first row of the launch (cached streaming chunks: the 128 lead rows in front of the first sequence
                                                             // hold nothing but the convolution tails -- 8 padding panels that pushed 8 streams' 128 real panels over the S = 2 limit)
};
// S > 1: S workgroups share a panel, each takes 1024 / S hidden columns of the feed-forward (its slice of W1 and of W2's K range: one
// workgroup ingests ~100 GB/s, and at few rows the 1.25 MB of weights per panel were the launch: 64 panels = 64 CUs busy for 20 us).
// Everything up to norm3 is repeated by every part (W_o: 256 KB); the S partial FF2 tiles meet in the panel's last-arriving workgroup
// (write-through partials, arrival counter, sc1 loads; cdna_hip_programming.md Guideline 16), which sums them in part order -- the
// result does not depend on who arrives last -- and runs the row epilogue.  grid = (S, panels).
template <int S>
__global__ __launch_bounds__(1024) void k_tail_panel(TailArgs a) {
    constexpr int NW = 16, LDC = 260, KS0 = 16, KS1 = 8, KS2 = 32, CH = 8;
    constexpr size_t ATT_BYTES = (size_t)KS0 * 1024, X_OFF = 17 * 1024, H_OFF = X_OFF + KS1 * 1024, PAR_OFF = H_OFF + KS2 * 1024;
    extern __shared__ __attribute__((aligned(16))) char smem[];        // [att panel 16 K | C tile 16.25 K (overlays it)][x panel 8 K][h panel 32 K][6 vectors]
    char* xs = smem + X_OFF;
    char* hs = smem + H_OFF;
    float* par = reinterpret_cast<float*>(smem + PAR_OFF);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int part = S > 1 ? (int)blockIdx.x : 0;
    constexpr int TPW = 4 / S, KS2P = KS2 / S;              // hidden column tiles per wave / k-steps of FF2 of one part
    const int m0 = a.row0 + blockIdx.y * 16, m = m0 + wave, n = lane * 4;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    {   // attention panel: piece kb = 16 rows x 32 k, one per wave
        const int srow = lane >> 2, schunk = (lane & 3) ^ ((lane >> 5) << 1);
        const uint16_t* src = a.att + (long)(m0 + srow) * a.lda + wave * 32 + schunk * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(smem + wave * 1024), 16, 0, 0);
    }
    if (wave < 6) {                                          // per-column vectors: bo, g3, b3, b2, gn, bn
        const float* src = wave == 0 ? a.bo : wave == 1 ? a.g3 : wave == 2 ? a.b3 : wave == 3 ? a.b2 : wave == 4 ? a.gn : a.bn;
        if (src)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + lane * 4),
                                             (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(par) + wave * 1024), 16, 0, 0);
    }
    f32x4 xrow = *reinterpret_cast<const f32x4*>(a.xf + (size_t)m * 256 + n);       // residual row of this wave
    __builtin_amdgcn_s_barrier();
    const s16x8* wop = reinterpret_cast<const s16x8*>(a.Wo) + ((size_t)wave * KS0) * 64 + lane;
    s16x8 wr[CH];
#pragma unroll
    for (int i = 0; i < CH; i++) wr[i] = wop[(size_t)i * 64];
    int ep_start = 0, ep_len = a.M_valid;
    if (a.seq.tile_seq) {
        typedef int i32x4_t __attribute__((ext_vector_type(4)));
        typedef const __attribute__((address_space(4))) i32x4_t k_i32x4;
        const i32x4_t ti = *reinterpret_cast<k_i32x4*>(reinterpret_cast<uintptr_t>(a.seq.tile_info + (m0 >> 6)));
        ep_start = ti.y; ep_len = ti.z;
    }
    const bool valid = (m - ep_start) < ep_len;
    vmcnt_wait<CH>();
    __builtin_amdgcn_s_barrier();
    const int a_off = subtile_off(lane & 15, lane >> 4);
    const f32x4* pv = reinterpret_cast<const f32x4*>(par) + lane;
    float* C = reinterpret_cast<float*>(smem);
    // ---- GEMM 0: O-projection, wave owns output columns [16 w, 16 w + 16)
    f32x4 acc = z4;
#pragma unroll
    for (int kb = 0; kb < KS0; kb++) {
        const bf16x8 af = *reinterpret_cast<const bf16x8*>(smem + (size_t)kb * 1024 + a_off);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wr[kb % CH]), af, acc, 0, 0, 0);
        if (kb + CH < KS0) wr[kb % CH] = wop[(size_t)(kb + CH) * 64];
    }
    // first fragments of W1 requested now: they fly during the epilogue below
    const s16x8* w1p = reinterpret_cast<const s16x8*>(a.W1) + ((size_t)(part * (64 / S) + wave * TPW) * KS1) * 64 + lane;
    s16x8 wa[KS1], wb[KS1];
#pragma unroll
    for (int i = 0; i < KS1; i++) wa[i] = w1p[(size_t)i * 64];
    __syncthreads();                                         // attention panel fully read: its LDS becomes the C tile
    *reinterpret_cast<f32x4*>(&C[(lane & 15) * LDC + wave * 16 + 4 * (lane >> 4)]) = acc;
    __syncthreads();
    {   // x += o + bo ; y = LN3(x) -> x panel (bf16, operand layout of GEMM 1: row = wave, 4 consecutive k per lane)
        f32x4 v = *reinterpret_cast<const f32x4*>(&C[wave * LDC + n]) + pv[0] + xrow;
        if (!valid) v = z4;
        xrow = v;
        const float mean = wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.f / 256);
        const f32x4 d = v - mean;
        const float var = wave_sum(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.f / 256);
        f32x4 y = d * rsqrtf(var + a.eps3) * pv[64] + pv[128];
        if (!valid) y = z4;
        char* dst = xs + (size_t)(n >> 5) * 1024 + subtile_off(wave, (n & 31) >> 3) + (n & 7) * 2;
        *reinterpret_cast<uint2*>(dst) = make_uint2(pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3]));
    }
    __syncthreads();
    // ---- GEMM 1: 4 column tiles of 16 hidden columns per wave, bias + GELU, hidden panel in operand layout
#pragma unroll
    for (int t = 0; t < TPW; t++) {
        s16x8 (&cur)[KS1] = (t & 1) ? wb : wa;
        s16x8 (&nxt)[KS1] = (t & 1) ? wa : wb;
        if (t + 1 < TPW) {
#pragma unroll
            for (int i = 0; i < KS1; i++) nxt[i] = w1p[(size_t)((t + 1) * KS1 + i) * 64];
        }
        f32x4 a1 = z4;
#pragma unroll
        for (int kb = 0; kb < KS1; kb++) {
            const bf16x8 xf_ = *reinterpret_cast<const bf16x8*>(xs + (size_t)kb * 1024 + a_off);
            a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, cur[kb]), xf_, a1, 0, 0, 0);
        }
        const int c = wave * (16 * TPW) + t * 16 + 4 * (lane >> 4);        // hidden column within the part
        f32x4 v = a1 + *reinterpret_cast<const f32x4*>(a.b1 + part * (1024 / S) + c);
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] = act_apply(v[e], ACT_GELU, 0.f);
        char* d = hs + (size_t)(c >> 5) * 1024 + subtile_off(lane & 15, (c & 31) >> 3) + (c & 7) * 2;
        *reinterpret_cast<uint2*>(d) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
    }
    // ---- GEMM 2
    const s16x8* w2p = reinterpret_cast<const s16x8*>(a.W2) + ((size_t)wave * KS2 + part * KS2P) * 64 + lane;
#pragma unroll
    for (int i = 0; i < CH; i++) wr[i] = w2p[(size_t)i * 64];
    __syncthreads();                                         // hidden panel complete (and the C tile of GEMM 0 consumed)
    // FF2 is summed as four chains of 8 k-steps combined pairwise, (q0 + q1) + (q2 + q3), in EVERY variant (one workgroup holds 4, 2 or 1
    // of the chains): a panel's result does not depend on how many workgroups shared it, so a cached streaming chunk (few rows, S = 4)
    // and the whole-prefix recompute (S = 2 or 1) stay bit-identical.
    f32x4 q2[KS2P / 8];
#pragma unroll
    for (int c8 = 0; c8 < KS2P / 8; c8++) q2[c8] = z4;
#pragma unroll
    for (int kb = 0; kb < KS2P; kb++) {
        const bf16x8 hf = *reinterpret_cast<const bf16x8*>(hs + (size_t)kb * 1024 + a_off);
        q2[kb / 8] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wr[kb % CH]), hf, q2[kb / 8], 0, 0, 0);
        if (kb + CH < KS2P) wr[kb % CH] = w2p[(size_t)(kb + CH) * 64];
    }
    f32x4 acc2;
    if (S == 1) acc2 = (q2[0] + q2[1 % (KS2P / 8)]) + (q2[2 % (KS2P / 8)] + q2[3 % (KS2P / 8)]);
    else if (S == 2) acc2 = q2[0] + q2[1 % (KS2P / 8)];
    else acc2 = q2[0];
    *reinterpret_cast<f32x4*>(&C[(lane & 15) * LDC + wave * 16 + 4 * (lane >> 4)]) = acc2;
    __syncthreads();
    f32x4 ff = *reinterpret_cast<const f32x4*>(&C[wave * LDC + n]);
    if (S > 1) {
        float* slab = a.part + (((size_t)blockIdx.y * S + part) * 16 + wave) * 256 + n;
        asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(slab), "v"(ff) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int* last = reinterpret_cast<int*>(C);                  // (the C tile has been read: its LDS carries the verdict)
        if (tid == 0) {
            const int old = __hip_atomic_fetch_add(a.ticket + blockIdx.y, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *last = old == S - 1;
            if (*last) __hip_atomic_store(a.ticket + blockIdx.y, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (!*last) return;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.part + (size_t)blockIdx.y * S * 16 * 256), 0, S * 16 * 256 * 4, 0x00020000);
        f32x4 pp[S];
#pragma unroll
        for (int p = 0; p < S; p++) {                           // every part's tile (this workgroup's own from its registers)
            const f32x4 o = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ((p * 16 + wave) * 256 + n) * 4, 0, 16));     // aux 16 = sc1
            pp[p] = p == part ? ff : o;
        }
        ff = S == 2 ? pp[0] + pp[1 % S] : (pp[0] + pp[1 % S]) + (pp[2 % S] + pp[3 % S]);      // the same tree, whoever folds
    }
    {   // x += ff + b2 ; stores
        f32x4 v = ff + pv[192] + xrow;
        if (!valid) v = z4;
        if (a.gn) {
            *reinterpret_cast<f32x4*>(a.xf + (size_t)m * 256 + n) = v;
            const float mean = wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.f / 256);
            const f32x4 d = v - mean;
            const float var = wave_sum(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.f / 256);
            f32x4 y = d * rsqrtf(var + a.epsn) * pv[256] + pv[320];
            if (!valid) y = z4;
            *reinterpret_cast<uint2*>(a.out_ln + (size_t)m * a.ldo_ln + n) = make_uint2(pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3]));
        } else {
            *reinterpret_cast<uint2*>(a.out_x + (size_t)m * a.ldo_x + n) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
        }
    }
}
constexpr size_t tail_panel_smem() { return (size_t)(17 + 8 + 32 + 6) * 1024; }

// The same chain for MANY rows (batched utterances): RT row tiles of 16 rows per block, so that every weight fragment fetched from
// L2 (1.25 MB per block) feeds RT MFMAs instead of one: at RT = 4 a block's 84 MFLOP take about as long on the CU's matrix cores as its
// weights take through the CU's 64 B/clk vector-memory path, where the three separate GEMM launches it replaces ran at ~12 % of the MFMA
// roof (N = 256 / K <= 1024 tiles, fp32 activations through HBM between them).  The 1024-wide hidden activations are produced and
// consumed in four quarters of 256 columns (LDS: attention panel 16 RT KiB | x panel 8 RT | hidden quarter 8 RT), the FF2
// accumulators stay in registers across the quarters.  Wave w owns output columns [16 w, 16 w + 16) of the N = 256 GEMMs and hidden
// columns [16 w, 16 w + 16) of each quarter; rows: wave w finishes rows w, w + 16, ... (tile i, row w) in the row epilogues.
template <int RT>
__global__ __launch_bounds__(1024) void k_tail_rows(TailArgs a) {
    constexpr int LDC = 260, KS0 = 16, KS1 = 8, KS2 = 32, CH = 8, QK = 8;      // QK = k-steps of one hidden quarter
    constexpr size_t C_BYTES = (size_t)RT * 16 * LDC * 4, ATT_BYTES = (size_t)KS0 * RT * 1024;
    constexpr size_t X_OFF = ((C_BYTES > ATT_BYTES ? C_BYTES : ATT_BYTES) + 1023) / 1024 * 1024, H_OFF = X_OFF + (size_t)KS1 * RT * 1024,
                     PAR_OFF = H_OFF + (size_t)QK * RT * 1024;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* xs = smem + X_OFF;
    char* hs = smem + H_OFF;
    float* par = reinterpret_cast<float*>(smem + PAR_OFF);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.y * (16 * RT), n = lane * 4;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    {   // attention panel: piece (kb, rt) = 16 rows x 32 k; wave w fetches k-step w of every row tile
        const int srow = lane >> 2, schunk = (lane & 3) ^ ((lane >> 5) << 1);
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
            const uint16_t* src = a.att + (long)(m0 + rt * 16 + srow) * a.lda + wave * 32 + schunk * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(smem + (size_t)(wave * RT + rt) * 1024), 16, 0, 0);
        }
    }
    if (wave < 6) {                                          // per-column vectors: bo, g3, b3, b2, gn, bn
        const float* src = wave == 0 ? a.bo : wave == 1 ? a.g3 : wave == 2 ? a.b3 : wave == 3 ? a.b2 : wave == 4 ? a.gn : a.bn;
        if (src)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + lane * 4),
                                             (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(par) + wave * 1024), 16, 0, 0);
    }
    f32x4 xrow[RT];                                           // residual rows of this wave: row wave of every tile
#pragma unroll
    for (int i = 0; i < RT; i++) xrow[i] = *reinterpret_cast<const f32x4*>(a.xf + (size_t)(m0 + 16 * i + wave) * 256 + n);
    __builtin_amdgcn_s_barrier();
    const s16x8* wop = reinterpret_cast<const s16x8*>(a.Wo) + ((size_t)wave * KS0) * 64 + lane;
    s16x8 wr[CH];
#pragma unroll
    for (int i = 0; i < CH; i++) wr[i] = wop[(size_t)i * 64];
    int ep_start = 0, ep_len = a.M_valid;
    if (a.seq.tile_seq) {
        typedef int i32x4_t __attribute__((ext_vector_type(4)));
        typedef const __attribute__((address_space(4))) i32x4_t k_i32x4;
        const i32x4_t ti = *reinterpret_cast<k_i32x4*>(reinterpret_cast<uintptr_t>(a.seq.tile_info + (m0 >> 6)));
        ep_start = ti.y; ep_len = ti.z;
    }
    vmcnt_wait<CH>();
    __builtin_amdgcn_s_barrier();
    const int a_off = subtile_off(lane & 15, lane >> 4);
    const f32x4* pv = reinterpret_cast<const f32x4*>(par) + lane;
    float* C = reinterpret_cast<float*>(smem);
    // ---- GEMM 0: O-projection
    f32x4 acc[RT];
#pragma unroll
    for (int rt = 0; rt < RT; rt++) acc[rt] = z4;
#pragma unroll
    for (int kb = 0; kb < KS0; kb++) {
        const bf16x8 wf = __builtin_bit_cast(bf16x8, wr[kb % CH]);
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
            const bf16x8 af = *reinterpret_cast<const bf16x8*>(smem + (size_t)(kb * RT + rt) * 1024 + a_off);
            acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af, acc[rt], 0, 0, 0);
        }
        if (kb + CH < KS0) wr[kb % CH] = wop[(size_t)(kb + CH) * 64];
    }
    // W1 fragments of the first hidden quarter: they fly during the epilogue below
    s16x8 w1[KS1], w2[QK];
    {
        const s16x8* w1p = reinterpret_cast<const s16x8*>(a.W1) + ((size_t)wave * KS1) * 64 + lane;
#pragma unroll
        for (int i = 0; i < KS1; i++) w1[i] = w1p[(size_t)i * 64];
    }
    __syncthreads();                                         // attention panel fully read: its LDS becomes the C tile
#pragma unroll
    for (int rt = 0; rt < RT; rt++) *reinterpret_cast<f32x4*>(&C[(rt * 16 + (lane & 15)) * LDC + wave * 16 + 4 * (lane >> 4)]) = acc[rt];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RT; i++) {   // x += o + bo ; y = LN3(x) -> x panel (bf16, operand layout of GEMM 1: tile i, row = wave)
        const bool valid = (m0 + 16 * i + wave - ep_start) < ep_len;
        f32x4 v = *reinterpret_cast<const f32x4*>(&C[(16 * i + wave) * LDC + n]) + pv[0] + xrow[i];
        if (!valid) v = z4;
        xrow[i] = v;
        const float mean = wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.f / 256);
        const f32x4 d = v - mean;
        const float var = wave_sum(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.f / 256);
        f32x4 y = d * rsqrtf(var + a.eps3) * pv[64] + pv[128];
        if (!valid) y = z4;
        char* dst = xs + (size_t)((n >> 5) * RT + i) * 1024 + subtile_off(wave, (n & 31) >> 3) + (n & 7) * 2;
        *reinterpret_cast<uint2*>(dst) = make_uint2(pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3]));
    }
    __syncthreads();
    // ---- feed-forward, one hidden quarter at a time; FF2 accumulates over the quarters
#pragma unroll
    for (int rt = 0; rt < RT; rt++) acc[rt] = z4;
#pragma unroll
    for (int hq = 0; hq < 4; hq++) {
        // GEMM 1 quarter: hidden columns 256 hq + [16 w, 16 w + 16), bias + GELU -> hidden panel (operand layout of GEMM 2)
        f32x4 a1[RT];
#pragma unroll
        for (int rt = 0; rt < RT; rt++) a1[rt] = z4;
#pragma unroll
        for (int kb = 0; kb < KS1; kb++) {
            const bf16x8 wf = __builtin_bit_cast(bf16x8, w1[kb]);
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
                const bf16x8 xf_ = *reinterpret_cast<const bf16x8*>(xs + (size_t)(kb * RT + rt) * 1024 + a_off);
                a1[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf_, a1[rt], 0, 0, 0);
            }
        }
        {   // this quarter's W2 fragments (k-steps 8 hq .. 8 hq + 7 of output tile `wave`), then the next quarter's W1
            const s16x8* w2p = reinterpret_cast<const s16x8*>(a.W2) + ((size_t)wave * KS2 + hq * QK) * 64 + lane;
#pragma unroll
            for (int i = 0; i < QK; i++) w2[i] = w2p[(size_t)i * 64];
        }
        const int cq = wave * 16 + 4 * (lane >> 4);                      // column inside the quarter
        const f32x4 b1v = *reinterpret_cast<const f32x4*>(a.b1 + hq * 256 + cq);
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
            f32x4 v = a1[rt] + b1v;
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = act_apply(v[e], ACT_GELU, 0.f);
            char* d = hs + (size_t)((cq >> 5) * RT + rt) * 1024 + subtile_off(lane & 15, (cq & 31) >> 3) + (cq & 7) * 2;
            *reinterpret_cast<uint2*>(d) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
        }
        if (hq + 1 < 4) {
            const s16x8* w1p = reinterpret_cast<const s16x8*>(a.W1) + ((size_t)((hq + 1) * 16 + wave) * KS1) * 64 + lane;
#pragma unroll
            for (int i = 0; i < KS1; i++) w1[i] = w1p[(size_t)i * 64];
        }
        __syncthreads();                                     // hidden quarter complete
#pragma unroll
        for (int kb = 0; kb < QK; kb++) {
            const bf16x8 wf = __builtin_bit_cast(bf16x8, w2[kb]);
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
                const bf16x8 hf = *reinterpret_cast<const bf16x8*>(hs + (size_t)(kb * RT + rt) * 1024 + a_off);
                acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, hf, acc[rt], 0, 0, 0);
            }
        }
        __syncthreads();                                     // hidden panel consumed: the next quarter may overwrite it
    }
#pragma unroll
    for (int rt = 0; rt < RT; rt++) *reinterpret_cast<f32x4*>(&C[(rt * 16 + (lane & 15)) * LDC + wave * 16 + 4 * (lane >> 4)]) = acc[rt];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RT; i++) {   // x += ff + b2 ; stores
        const int m = m0 + 16 * i + wave;
        const bool valid = (m - ep_start) < ep_len;
        f32x4 v = *reinterpret_cast<const f32x4*>(&C[(16 * i + wave) * LDC + n]) + pv[192] + xrow[i];
        if (!valid) v = z4;
        if (a.gn) {
            *reinterpret_cast<f32x4*>(a.xf + (size_t)m * 256 + n) = v;
            const float mean = wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.f / 256);
            const f32x4 d = v - mean;
            const float var = wave_sum(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.f / 256);
            f32x4 y = d * rsqrtf(var + a.epsn) * pv[256] + pv[320];
            if (!valid) y = z4;
            *reinterpret_cast<uint2*>(a.out_ln + (size_t)m * a.ldo_ln + n) = make_uint2(pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3]));
        } else {
            *reinterpret_cast<uint2*>(a.out_x + (size_t)m * a.ldo_x + n) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
        }
    }
}
// ---- round 5: the same chain with 32-column wave tiles and ONE weight stream (k_tail_rows2)
// k_tail_rows<4> spends ~1 400 cycles per 32-wide k-step where its matrix-core work is 256: 16 waves x (16 columns x 64 rows) re-read the
// block's four activation row tiles from LDS once per 16 columns (64 KB per k-step = 512 cycles of LDS bandwidth), every GEMM of the chain
// starts with its weight fragments' L2 round trip exposed, and every phase ends in a block barrier with all 16 waves in step.  Here:
//   * 8 waves, wave w owns columns [32 w, 32 w + 32) x 64 rows: two weight fragments and four activation fragments feed eight MFMAs, so a
//     k-step moves 32 KB through LDS (256 cycles), 16 KB of weights through the CU's vector-memory path (256 cycles) and holds the matrix
//     cores 256 cycles -- balanced instead of LDS-bound;
//   * the block's GEMMs (O projection, 4 x {FF1 quarter, FF2 quarter}, optionally six 256-column tiles of the next block's QKV projection)
//     are ONE sequence for the weight ring: a slot freed at k-step kb is refilled with the fragment 8 k-steps on, whichever GEMM that
//     belongs to, so the stream runs through the epilogues and barriers;
//   * the hidden quarters are double-buffered (in the LDS of the attention panel, dead by then): one barrier per quarter, FF2 of quarter
//     q and FF1 of quarter q + 1 run back to back, and the two waves of a SIMD drift apart (one in GELU, one in MFMAs).
// Every sum keeps k_tail_rows<4>'s order (k-steps in order in one accumulator per output tile), the epilogue expressions are the same:
// the results are bit-identical to it (tests/test_flow_gpu.py), and the chained QKV projection is k_gemm's sum (8 k-steps in order).
#define TR2_CH 8
struct Tr2Ring { s16x8 r0[TR2_CH], r1[TR2_CH]; };
// fragment `frag` (1 KiB) behind a wave-uniform pointer, this lane's 16 bytes: scalar base + ONE per-lane byte offset (written as
// w[frag * 64 + lane] hipcc materialises a VGPR offset per distinct fragment index: 60 of them live across the kernel, 370 spilled)
__device__ __forceinline__ s16x8 tr2_ld(const s16x8* w, int frag, unsigned lane16) {
    const char* base = reinterpret_cast<const char*>(w) + (size_t)frag * 1024;
    return *reinterpret_cast<const s16x8*>(base + lane16);
}
// fragments kb = 0 .. min(CH, NKS) - 1 of a GEMM's two column tiles into the ring (the very first GEMM of the block)
template <int NKS>
__device__ __forceinline__ void tr2_prime(Tr2Ring& R, const s16x8* w0, const s16x8* w1, unsigned lane16) {
#pragma unroll
    for (int i = 0; i < (NKS < TR2_CH ? NKS : TR2_CH); i++) { R.r0[i] = tr2_ld(w0, i, lane16); R.r1[i] = tr2_ld(w1, i, lane16); }
}
// acc[ct][rt] += W tiles (w0, w1: this wave's two 16-column tiles, NKS k-steps) x panel rows; ring slots are refilled with this GEMM's
// later fragments, then with the NEXT GEMM's first ones (nw0 / nw1, NNKS k-steps; null: nothing follows).  PH = ring phase: the slot of
// this GEMM's k-step 0 (the sequence's k-steps are numbered through: slot = (PH + kb) % CH).
struct Tr2NoHook { __device__ __forceinline__ void operator()(int) const {} };
// `side(kb)`: other work of the wave issued beside k-step kb's MFMAs, inside the same scheduling region (the feed-forward's GELU of the NEXT
// hidden quarter rides on FF2's k-steps: VALU instructions issue while the matrix cores work on the eight MFMAs)
// SWAP: the activation fragment is the MFMA's first operand and the weight fragment its second -- the same products summed over k in the same
// order (bit-identical values), but the accumulator tile comes out TRANSPOSED: a lane then holds four consecutive ROWS of one column (the
// chained V projection stages its tile as V^T with 8-byte writes instead of gathering 2-byte elements, k_tail_rows2)
template <int NKS, int NNKS, int PH, class SIDE = Tr2NoHook, bool SWAP = false>
__device__ __forceinline__ void tr2_gemm(Tr2Ring& R, const s16x8* w0, const s16x8* w1, const s16x8* nw0, const s16x8* nw1,
                                         const char* panel, int a_off, unsigned lane16, f32x4 (&acc)[2][4], SIDE side = SIDE()) {
    // software pipeline, fenced per k-step (left to itself hipcc hoists the fully unrolled loop's LDS reads and weight loads far ahead:
    // 256 VGPRs + 370 spilled): the NEXT k-step's four activation fragments and this slot's refill are requested, then this k-step's
    // eight MFMAs run
    bf16x8 af[2][4];
#pragma unroll
    for (int rt = 0; rt < 4; rt++) af[0][rt] = *reinterpret_cast<const bf16x8*>(panel + (size_t)rt * 1024 + a_off);
#ifdef TR2_DIAG_MFMA32
    // TIMING-ONLY diagnostic (round 6, the round-5 review's item 4a; results are numerically meaningless): the k-step's eight
    // v_mfma_f32_16x16x32_bf16 (each holds the SIMD's vector issue 8 of its 16 cycles) replaced by four v_mfma_f32_32x32x16_bf16 (8 of 32)
    // on the SAME operand and accumulator registers -- same matrix-core time, same LDS / weight traffic, half the issue cycles taken from the
    // GELU that rides beside them.  What the feed-forward phase would gain from 32 x 32 tiles, before any layout is rewritten.
    f32x16 t32a, t32b;
#pragma unroll
    for (int i = 0; i < 16; i++) { t32a[i] = acc[0][i >> 2][i & 3]; t32b[i] = acc[1][i >> 2][i & 3]; }
#endif
#pragma unroll
    for (int kb = 0; kb < NKS; kb++) {
        const int sl = (PH + kb) % TR2_CH;
        const bf16x8 f0 = __builtin_bit_cast(bf16x8, R.r0[sl]), f1 = __builtin_bit_cast(bf16x8, R.r1[sl]);
        if (kb + 1 < NKS) {
#pragma unroll
            for (int rt = 0; rt < 4; rt++) af[(kb + 1) & 1][rt] = *reinterpret_cast<const bf16x8*>(panel + (size_t)((kb + 1) * 4 + rt) * 1024 + a_off);
        }
        __builtin_amdgcn_sched_barrier(0);
#ifdef TR2_DIAG_MFMA32
        t32a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0, af[kb & 1][0], t32a, 0, 0, 0);
        t32b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1, af[kb & 1][1], t32b, 0, 0, 0);
        t32a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0, af[kb & 1][2], t32a, 0, 0, 0);
        t32b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1, af[kb & 1][3], t32b, 0, 0, 0);
        side(kb);
        asm volatile("" : "+v"(t32a), "+v"(t32b));
        if (kb + TR2_CH < NKS) { R.r0[sl] = tr2_ld(w0, kb + TR2_CH, lane16); R.r1[sl] = tr2_ld(w1, kb + TR2_CH, lane16); }
        else if (NNKS > 0 && kb + TR2_CH - NKS < NNKS) { R.r0[sl] = tr2_ld(nw0, kb + TR2_CH - NKS, lane16); R.r1[sl] = tr2_ld(nw1, kb + TR2_CH - NKS, lane16); }
        __builtin_amdgcn_sched_barrier(0);
        continue;
#endif
#pragma unroll
        for (int rt = 0; rt < 4; rt++) {
#ifndef TR2_DIAG_NOMFMA
            acc[0][rt] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[kb & 1][rt], f0, acc[0][rt], 0, 0, 0)
                              : __builtin_amdgcn_mfma_f32_16x16x32_bf16(f0, af[kb & 1][rt], acc[0][rt], 0, 0, 0);
            acc[1][rt] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[kb & 1][rt], f1, acc[1][rt], 0, 0, 0)
                              : __builtin_amdgcn_mfma_f32_16x16x32_bf16(f1, af[kb & 1][rt], acc[1][rt], 0, 0, 0);
#else
            acc[0][rt] += __builtin_bit_cast(f32x4, f0) + __builtin_bit_cast(f32x4, af[kb & 1][rt]);
            acc[1][rt] += __builtin_bit_cast(f32x4, f1);
#endif
        }
        side(kb);
        // (both column tiles' accumulator chains pass through one opaque statement per k-step: hipcc otherwise runs tile 0's chain over all
        // k-steps first and keeps every activation fragment for tile 1's pass -- in scratch)
        asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[0][3]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[1][2]), "+v"(acc[1][3]));
        // (the slot is free once its fragments sit in the MFMAs' operands)
        if (kb + TR2_CH < NKS) { R.r0[sl] = tr2_ld(w0, kb + TR2_CH, lane16); R.r1[sl] = tr2_ld(w1, kb + TR2_CH, lane16); }
        else if (NNKS > 0 && kb + TR2_CH - NKS < NNKS) { R.r0[sl] = tr2_ld(nw0, kb + TR2_CH - NKS, lane16); R.r1[sl] = tr2_ld(nw1, kb + TR2_CH - NKS, lane16); }
        __builtin_amdgcn_sched_barrier(0);
    }
#ifdef TR2_DIAG_MFMA32
#pragma unroll
    for (int i = 0; i < 16; i++) { acc[0][i >> 2][i & 3] = t32a[i]; acc[1][i >> 2][i & 3] = t32b[i]; }
#endif
}
template <bool QKV>
__global__ __launch_bounds__(512) void k_tail_rows2(TailArgs a) {
    constexpr int RT = 4, LDC = 260, KS0 = 16, KS1 = 8, KS2 = 32;
    constexpr size_t P_BYTES = 68 * 1024, X_OFF = P_BYTES, PAR_OFF = X_OFF + 32 * 1024;     // P: attention panel 64 K | C tile 65 K | hidden quarters 2 x 32 K | QKV staging 2 x 33 K
    constexpr int SLD = 264;                                                              // bf16 per staging row (256 + 8)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* xs = smem + X_OFF;
    float* par = reinterpret_cast<float*>(smem + PAR_OFF);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.y * 64, n = lane * 4;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    SK_STAMP_DECL;
    SK_STAMP(0);
    {   // attention panel: piece (kb, rt) = 16 rows x 32 k; wave w fetches k-steps w and w + 8 of every row tile
        const int srow = lane >> 2, schunk = (lane & 3) ^ ((lane >> 5) << 1);
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
                const int kb = wave + 8 * h;
                const uint16_t* src = a.att + (long)(m0 + rt * 16 + srow) * a.lda + kb * 32 + schunk * 8;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(smem + (size_t)(kb * RT + rt) * 1024), 16, 0, 0);
            }
    }
    if (wave < 6) {                                          // per-column vectors: bo, g3, b3, b2, gn, bn
        const float* src = wave == 0 ? a.bo : wave == 1 ? a.g3 : wave == 2 ? a.b3 : wave == 3 ? a.b2 : wave == 4 ? a.gn : a.bn;
        if (src)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + lane * 4),
                                             (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(par) + wave * 1024), 16, 0, 0);
    }
    f32x4 xrow[8];                                            // this wave's rows of the residual stream: rows wave + 8 i
#pragma unroll
    for (int i = 0; i < 8; i++) xrow[i] = *reinterpret_cast<const f32x4*>(a.xf + (size_t)(m0 + wave + 8 * i) * 256 + n);
    __builtin_amdgcn_s_barrier();
    // packed weights: fragment (16-column tile t, k-step kb) of a [N][K] matrix = 1 KiB at ((t * K / 32 + kb) * 64 + lane) * 16 B
    // (wave-uniform pointers: the loads take the scalar base + lane offset form; per-lane pointers for all 30 weight tiles cost 60 VGPRs)
    const s16x8* wo0 = reinterpret_cast<const s16x8*>(a.Wo) + ((size_t)(2 * wave) * KS0) * 64;
    const s16x8* wo1 = wo0 + (size_t)KS0 * 64;
    auto w1p = [&](int hq, int ct) { return reinterpret_cast<const s16x8*>(a.W1) + ((size_t)(hq * 16 + 2 * wave + ct) * KS1) * 64; };
    auto w2p = [&](int hq, int ct) { return reinterpret_cast<const s16x8*>(a.W2) + ((size_t)(2 * wave + ct) * KS2 + hq * 8) * 64; };
    auto wqp = [&](int j, int ct) { return reinterpret_cast<const s16x8*>(a.Wqkv) + ((size_t)(j * 16 + 2 * wave + ct) * KS1) * 64; };
    Tr2Ring R;
    const unsigned lane16 = (unsigned)lane * 16u;
    tr2_prime<KS0>(R, wo0, wo1, lane16);
    int ep_start = 0, ep_len = a.M_valid;
    if (a.seq.tile_seq) {
        typedef int i32x4_t __attribute__((ext_vector_type(4)));
        typedef const __attribute__((address_space(4))) i32x4_t k_i32x4;
        const i32x4_t ti = *reinterpret_cast<k_i32x4*>(reinterpret_cast<uintptr_t>(a.seq.tile_info + (m0 >> 6)));
        ep_start = ti.y; ep_len = ti.z;
    }
    vmcnt_wait<2 * TR2_CH>();                                // everything older than the ring's loads: the panel, the vectors, the residual rows
    __builtin_amdgcn_s_barrier();
    SK_STAMP(1);                                             // panel, vectors and residual rows arrived
    const int a_off = subtile_off(lane & 15, lane >> 4);
    const f32x4* pv = reinterpret_cast<const f32x4*>(par) + lane;
    float* C = reinterpret_cast<float*>(smem);
    auto zero = [&](f32x4 (&acc)[2][4]) {
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int rt = 0; rt < 4; rt++) acc[c][rt] = z4;
    };
    auto to_c = [&](const f32x4 (&acc)[2][4]) {              // accumulators -> C tile [64][LDC]
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int rt = 0; rt < 4; rt++)
                *reinterpret_cast<f32x4*>(&C[(rt * 16 + (lane & 15)) * LDC + wave * 32 + c * 16 + 4 * (lane >> 4)]) = acc[c][rt];
    };
    // ---- GEMM 0: O projection (ring phase 0; followed by FF1 quarter 0)
    f32x4 acc[2][4], acc2[2][4];
    zero(acc);
    tr2_gemm<KS0, KS1, 0>(R, wo0, wo1, w1p(0, 0), w1p(0, 1), smem, a_off, lane16, acc);
    SK_STAMP(2);                                             // O projection's MFMAs issued
    TR2_BARRIER;                                         // attention panel fully read: its LDS becomes the C tile
    to_c(acc);
    TR2_BARRIER;
#pragma unroll
    for (int i = 0; i < 8; i++) {   // x += o + bo ; y = LN3(x) -> x panel (bf16, operand layout of FF1)
        const int r = wave + 8 * i;
        const bool valid = (m0 + r - ep_start) < ep_len;
        f32x4 v = *reinterpret_cast<const f32x4*>(&C[r * LDC + n]) + pv[0] + xrow[i];
        if (!valid) v = z4;
        xrow[i] = v;
        const float mean = wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.f / 256);
        const f32x4 d = v - mean;
        const float var = wave_sum(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.f / 256);
        f32x4 y = d * rsqrtf(var + a.eps3) * pv[64] + pv[128];
        if (!valid) y = z4;
        char* dst = xs + (size_t)((n >> 5) * RT + (r >> 4)) * 1024 + subtile_off(r & 15, (n & 31) >> 3) + (n & 7) * 2;
        *reinterpret_cast<uint2*>(dst) = make_uint2(pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3]));
    }
    TR2_BARRIER;                                         // x panel complete; the C tile is consumed: its LDS becomes the hidden quarters
    // ---- feed-forward, one hidden quarter at a time; FF2 accumulates over the quarters (in order: the sums are k_tail_rows<4>'s).  The
    // GELU of a quarter is the block's largest VALU item (32 values per thread: ~7 000 cycles per quarter beside 4 400 of MFMAs when it runs
    // on its own): quarter q's GELU rides on the k-steps of FF2(q - 1) -- one accumulator tile per k-step -- so only quarter 0's is exposed.
    // GEMM order for the weight ring: FF1(0), FF1(1), FF2(0), FF1(2), FF2(1), FF1(3), FF2(2), FF2(3)[, QKV tiles]; every one 8 k-steps, phase 0.
    SK_STAMP(3);                                             // norm3 rows done, x panel complete
    zero(acc2);
    f32x4 b1v[2];
    auto gelu_tile = [&](int hq, int c, int rt) {            // acc[c][rt] (+ b1) -> GELU -> hidden quarter hq's panel (operand layout of FF2)
        char* hs = smem + (size_t)(hq & 1) * 32 * 1024;
        const int cq = wave * 32 + c * 16 + 4 * (lane >> 4);
        f32x4 v = acc[c][rt] + b1v[c];
#ifndef TR2_DIAG_NOGELU                                     // (diagnostic builds: where does the feed-forward phase spend its time?)
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] = act_apply(v[e], ACT_GELU, 0.f);
#endif
        char* d = hs + (size_t)((cq >> 5) * RT + rt) * 1024 + subtile_off(lane & 15, (cq & 31) >> 3) + (cq & 7) * 2;
        *reinterpret_cast<uint2*>(d) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
    };
    auto load_b1 = [&](int hq) {
#pragma unroll
        for (int c = 0; c < 2; c++) b1v[c] = *reinterpret_cast<const f32x4*>(a.b1 + hq * 256 + wave * 32 + c * 16 + 4 * (lane >> 4));
    };
    load_b1(0);
    zero(acc);
    tr2_gemm<KS1, KS1, 0>(R, w1p(0, 0), w1p(0, 1), w1p(1, 0), w1p(1, 1), xs, a_off, lane16, acc);
#pragma unroll
    for (int t = 0; t < 8; t++) gelu_tile(0, t >> 2, t & 3);
    TR2_BARRIER;                                             // hidden quarter 0 complete
#define TR2_STEP(HQ, NW0, NW1, NNKS)   /* FF1(HQ), then FF2(HQ - 1) with GELU(HQ) beside it; next in the ring: NW */                   \
    {                                                                                                                               \
        load_b1(HQ);                                                                                                                 \
        zero(acc);                                                                                                                   \
        tr2_gemm<KS1, KS1, 0>(R, w1p(HQ, 0), w1p(HQ, 1), w2p((HQ) - 1, 0), w2p((HQ) - 1, 1), xs, a_off, lane16, acc);                  \
        tr2_gemm<KS1, NNKS, 0>(R, w2p((HQ) - 1, 0), w2p((HQ) - 1, 1), NW0, NW1, smem + (size_t)(((HQ) - 1) & 1) * 32 * 1024, a_off, lane16, acc2, \
                               [&](int kb) { gelu_tile(HQ, kb >> 2, kb & 3); });                                                      \
        TR2_BARRIER;                                         /* hidden quarter HQ complete, quarter HQ - 1 consumed */              \
    }
    TR2_STEP(1, w1p(2, 0), w1p(2, 1), KS1)
    TR2_STEP(2, w1p(3, 0), w1p(3, 1), KS1)
    TR2_STEP(3, w2p(3, 0), w2p(3, 1), KS1)
#undef TR2_STEP
    if (QKV) tr2_gemm<KS1, KS1, 0>(R, w2p(3, 0), w2p(3, 1), wqp(0, 0), wqp(0, 1), smem + 32 * 1024, a_off, lane16, acc2);
    else tr2_gemm<KS1, 0, 0>(R, w2p(3, 0), w2p(3, 1), (const s16x8*)nullptr, (const s16x8*)nullptr, smem + 32 * 1024, a_off, lane16, acc2);
    SK_STAMP(4);                                             // feed-forward quarters done
    TR2_BARRIER;                                         // the last hidden quarter is consumed: the LDS becomes the C tile again
    to_c(acc2);
    TR2_BARRIER;
#pragma unroll
    for (int i = 0; i < 8; i++) {   // x += ff + b2 ; stores
        const int r = wave + 8 * i, m = m0 + r;
        const bool valid = (m - ep_start) < ep_len;
        f32x4 v = *reinterpret_cast<const f32x4*>(&C[r * LDC + n]) + pv[192] + xrow[i];
        if (!valid) v = z4;
        if (a.gn) {
            *reinterpret_cast<f32x4*>(a.xf + (size_t)m * 256 + n) = v;
            const float mean = wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.f / 256);
            const f32x4 d = v - mean;
            const float var = wave_sum(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.f / 256);
            f32x4 y = d * rsqrtf(var + a.epsn) * pv[256] + pv[320];
            if (!valid) y = z4;
            const uint2 yb = make_uint2(pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3]));
            if (QKV) *reinterpret_cast<uint2*>(xs + (size_t)((n >> 5) * RT + (r >> 4)) * 1024 + subtile_off(r & 15, (n & 31) >> 3) + (n & 7) * 2) = yb;
            else *reinterpret_cast<uint2*>(a.out_ln + (size_t)m * a.ldo_ln + n) = yb;
        } else {
            *reinterpret_cast<uint2*>(a.out_x + (size_t)m * a.ldo_x + n) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
        }
    }
    if (!QKV) {
        SK_STAMP(5);
        SK_STAMP_FLUSH_RING(((unsigned long long)0x7A11 << 32) | 256u, ((unsigned long long)64 << 48) | ((unsigned long long)32 << 32) | (unsigned)gridDim.y);
        return;
    }
    TR2_BARRIER;                                         // y panel complete (and the C tile consumed: its LDS becomes the staging tiles)
    // ---- the next block's QKV projection: six tiles of 256 features; a tile is staged as bf16 [64][SLD] (two buffers) and stored by the
    // whole block in full rows (Q / K: 512 B per row) or full columns (V^T: 128 B per feature)
#define TR2_QKV(J, NW0, NW1, NNKS)                                    /* Q / K features 256 J ..: staged [64 rows][SLD], stored in rows of 512 B */ \
    {                                                                                                                               \
        uint16_t* S = reinterpret_cast<uint16_t*>(smem + (size_t)((J) & 1) * 34 * 1024);                                              \
        zero(acc);                                                                                                                   \
        tr2_gemm<KS1, NNKS, 0>(R, wqp(J, 0), wqp(J, 1), NW0, NW1, xs, a_off, lane16, acc);                                                    \
        _Pragma("unroll") for (int c = 0; c < 2; c++)                                                                                \
            _Pragma("unroll") for (int rt = 0; rt < 4; rt++) {                                                                       \
                const int r = rt * 16 + (lane & 15);                                                                                 \
                f32x4 v = acc[c][rt];                                                                                                \
                if ((m0 + r - ep_start) >= ep_len) v = z4;                                                                           \
                *reinterpret_cast<uint2*>(S + (size_t)r * SLD + wave * 32 + c * 16 + 4 * (lane >> 4)) =                               \
                    make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));                                                    \
            }                                                                                                                       \
        TR2_BARRIER;                                                                                                             \
        _Pragma("unroll") for (int it = 0; it < 4; it++) {                                              /* 32 threads x 16 B per row */ \
            const int e = tid + 512 * it, r = e >> 5, ch = e & 31;                                                                   \
            *reinterpret_cast<uint4*>(a.qk + (size_t)(m0 + r) * 1024 + (J) * 256 + ch * 8) = *reinterpret_cast<const uint4*>(S + (size_t)r * SLD + ch * 8); \
        }                                                                                                                           \
    }
    // V features: the accumulators come out transposed (tr2_gemm<.., SWAP>: a lane holds four consecutive rows of one feature), so the tile is
    // staged AS V^T [256 features][TLD] with 8-byte writes and leaves in 16-byte pieces (8 rows of a feature: 8 threads cover the 128 B of a
    // feature's 64 rows).  Staged [64][SLD] like Q / K, every 16-byte piece was eight 2-byte LDS reads a row apart -- all on one bank: the
    // two V tiles cost 12 us of a 207 us launch more than the four Q / K tiles' way of storing.
#define TR2_V(J, NW0, NW1, NNKS)                                                                                                     \
    {                                                                                                                               \
        constexpr int TLD = 68;                              /* bf16 per staged V^T row: 64 + 4 (256 x 68 x 2 B = the 34 KB staging buffer) */ \
        uint16_t* S = reinterpret_cast<uint16_t*>(smem + (size_t)((J) & 1) * 34 * 1024);                                              \
        zero(acc);                                                                                                                   \
        tr2_gemm<KS1, NNKS, 0, Tr2NoHook, true>(R, wqp(J, 0), wqp(J, 1), NW0, NW1, xs, a_off, lane16, acc);                          \
        _Pragma("unroll") for (int c = 0; c < 2; c++)                                                                                \
            _Pragma("unroll") for (int rt = 0; rt < 4; rt++) {                                                                       \
                const int r0 = rt * 16 + 4 * (lane >> 4);        /* this lane's rows r0 .. r0 + 3 of feature f */                    \
                const int f = wave * 32 + c * 16 + (lane & 15);                                                                      \
                f32x4 v = acc[c][rt];                                                                                                \
                _Pragma("unroll") for (int e = 0; e < 4; e++) if ((m0 + r0 + e - ep_start) >= ep_len) v[e] = 0.f;                     \
                *reinterpret_cast<uint2*>(S + (size_t)f * TLD + r0) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));  \
            }                                                                                                                       \
        TR2_BARRIER;                                                                                                             \
        _Pragma("unroll") for (int it = 0; it < 4; it++) {                                                                           \
            const int e = tid + 512 * it, f = e >> 3, g = e & 7;                                                                     \
            const uint2 lo = *reinterpret_cast<const uint2*>(S + (size_t)f * TLD + 8 * g), hi = *reinterpret_cast<const uint2*>(S + (size_t)f * TLD + 8 * g + 4); \
            *reinterpret_cast<uint4*>(a.vt + (size_t)(((J) - 4) * 256 + f) * a.vt_ld + m0 + 8 * g) = make_uint4(lo.x, lo.y, hi.x, hi.y); \
        }                                                                                                                           \
    }
    TR2_QKV(0, wqp(1, 0), wqp(1, 1), KS1)
    TR2_QKV(1, wqp(2, 0), wqp(2, 1), KS1)
    TR2_QKV(2, wqp(3, 0), wqp(3, 1), KS1)
    TR2_QKV(3, wqp(4, 0), wqp(4, 1), KS1)
    TR2_V(4, wqp(5, 0), wqp(5, 1), KS1)
    TR2_V(5, (const s16x8*)nullptr, (const s16x8*)nullptr, 0)
#undef TR2_V
#undef TR2_QKV
    SK_STAMP(5);
    SK_STAMP_FLUSH_RING(((unsigned long long)0x7A11 << 32) | 1536u, ((unsigned long long)64 << 48) | ((unsigned long long)32 << 32) | (unsigned)gridDim.y);
}
constexpr size_t tail_rows2_smem() { return (size_t)(68 + 32 + 6) * 1024; }

template <int RT>
constexpr size_t tail_rows_smem() {
    constexpr size_t c = (size_t)RT * 16 * 260 * 4, att = (size_t)16 * RT * 1024;
    return ((c > att ? c : att) + 1023) / 1024 * 1024 + (size_t)(8 + 8) * RT * 1024 + 6 * 1024;
}

template <int BM, int BN, bool SPLITA = false, int NSTAGE = 2>
constexpr size_t gemm_smem_bytes() {
    constexpr size_t stages = NSTAGE * (size_t)(((SPLITA ? 2 : 1) * BM + BN) / 16 * 2) * 1024;
    constexpr size_t ctile = (size_t)BM * (BN + 4) * 4;
    return stages > ctile ? stages : ctile;
}

