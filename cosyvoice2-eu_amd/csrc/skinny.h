// Skinny GEMM core for weight-streaming decode (rows <= 32): out[r][n] = sum_k W[n][k] * f(x)[r][k].
//
// Roofline: HBM.  Every weight byte is read exactly once per launch with 16 B/lane non-temporal loads from the
// packed layout (include/cv2_amd.h): one wave-load = one contiguous 1 KiB block = one MFMA A operand.
// The activation operand is split x = hi + lo (two bf16) and multiplied twice, so the products keep ~16
// mantissa bits of x while the MFMA (otherwise idle in this HBM-bound kernel) does the work; accumulation fp32.
//
// Block = NWR row tiles (16 output features each) x NWK K-slices (one wave each); gridDim.y splits K further.
// A wave puts ALL of its weight fragments (<= MAXKS x 1 KiB) in flight first.  While they travel from HBM the
// block folds the residual stream (x = base + partial sums left by the previous kernel, fixed order), computes
// the per-row RMS, normalises, splits and writes the B operands to LDS -- one L2 round trip, hidden under the
// HBM latency of the weight stream.
#pragma once
#include "common.h"

#define SK_ROWS_CAP 32    // row capacity of partial buffers

#define SK_MAXNP 4        // partial buffers a consumer can fold

typedef __attribute__((ext_vector_type(8))) float f32x8;

#define SK_MAXSPLIT 16    // key splits of the decode attention a consumer can combine

struct SkinnyX {
    const float* base;    // [rows][K]
    const float* parts;   // [np][SK_ROWS_CAP][K] partial sums added to base in index order (may be null)
    int np;               // <= SK_MAXNP
    const float* norm_w;  // RMSNorm weight [K] or null (requires gridDim.y == 1)
    float eps;
    float* x_out;         // optional [rows][K]: receives base + sum(parts) (pre-norm); written by blockIdx.x == 0
    // split-key attention combine (flash-decoding): x[r][k] = sum_s w_s o_s[r][k] / sum_s w_s l_s, w_s = exp(m_s - max m),
    // o_s = parts[s][r][k] unnormalised, (m_s, l_s) = att_ml[((s * SK_ROWS_CAP + r) * (K / 64) + k / 64) * 2 + {0, 1}]
    const float* att_ml;
    const int* att_cnt;   // != null selects this mode (base unused, np = split slots): [SK_ROWS_CAP] non-empty splits per row
    // PRE kernels: the operand already folded / normalised / split by k_prep, in LDS B-operand order
    // [K/32][NB=2][hi, lo][1 KiB]; the prologue is a straight copy of the block's K slice
    const uint16_t* pre;
};

__device__ __forceinline__ void split8(const f32x8 v, bf16x8& hi, bf16x8& lo) {
    hi = __builtin_convertvector(v, bf16x8);                       // v_cvt_pk_bf16_f32, RNE
    const f32x8 back = __builtin_convertvector(hi, f32x8);
    lo = __builtin_convertvector(v - back, bf16x8);
}

// Operand loads come in two halves so that callers can put other loads between them: sk_issue_x only issues (no value is
// consumed, so no wait is emitted), sk_finish_x folds what arrived.
template <bool ATT>
struct SkRaw;
template <>
struct SkRaw<false> { f32x8 v; f32x8 p[SK_MAXNP]; };
template <>
struct SkRaw<true> { float mv[SK_MAXSPLIT], lv[SK_MAXSPLIT]; f32x8 ov[SK_MAXSPLIT]; int ns; };

__device__ __forceinline__ void sk_issue_x(const SkinnyX& X, int r, int K, int k, SkRaw<true>& o) {
    const int nq = K >> 6, hd = k >> 6;
    // every split slot is loaded unconditionally, together with the split count: nothing waits on the count before issuing
    // (slots beyond it hold stale values and are only masked out in sk_finish_x).  Addresses are a uniform base (scalar
    // registers) plus ONE 32-bit per-thread offset shared by all slots: the issue phase is bound by instruction count.
    o.ns = X.att_cnt[r];
    const unsigned off_ml = (unsigned)(r * nq + hd) * 2u, off_o = (unsigned)(r * K + k);
#pragma unroll
    for (int s = 0; s < SK_MAXSPLIT; s++) {
        if (s < X.np) {           // np = number of split slots the engine was created with (uniform): older than the weight loads,
            const float* ml = X.att_ml + (size_t)s * SK_ROWS_CAP * nq * 2;      // so a run-time count here costs no wait precision
            const float2 mlv = *reinterpret_cast<const float2*>(ml + off_ml);
            o.mv[s] = mlv.x; o.lv[s] = mlv.y;
            o.ov[s] = *reinterpret_cast<const f32x8*>(X.parts + (size_t)s * SK_ROWS_CAP * K + off_o);
        }
    }
}
__device__ __forceinline__ f32x8 sk_finish_x(const SkinnyX& X, const SkRaw<true>& o) {
    float M = -INFINITY;
#pragma unroll
    for (int s = 0; s < SK_MAXSPLIT; s++) if (s < o.ns) M = fmaxf(M, o.mv[s]);
    f32x8 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float den = 0.f;
#pragma unroll
    for (int s = 0; s < SK_MAXSPLIT; s++) {
        if (s < o.ns) {
            const float w = __expf(o.mv[s] - M);
            den += w * o.lv[s];
            acc += w * o.ov[s];
        }
    }
    return acc * (1.f / den);
}
// Half-range variant for the one-row O-projection: two thread groups fold SK_MAXSPLIT / 2 split slots each (half the loads and
// half the exp / FMA work per thread) and merge their (max, denominator, numerator) through LDS.
#define SK_HALFSPLIT (SK_MAXSPLIT / 2)
struct SkRawHalf { float mv[SK_HALFSPLIT], lv[SK_HALFSPLIT]; f32x8 ov[SK_HALFSPLIT]; int ns; };
__device__ __forceinline__ void sk_issue_att_half(const SkinnyX& X, int r, int K, int k, int s_lo, SkRawHalf& o) {
    const int nq = K >> 6, hd = k >> 6;
    o.ns = X.att_cnt[r];
    const unsigned off_ml = (unsigned)(r * nq + hd) * 2u, off_o = (unsigned)(r * K + k);
#pragma unroll
    for (int i = 0; i < SK_HALFSPLIT; i++) {
        const int s = s_lo + i;
        if (s < X.np) {
            const float2 mlv = *reinterpret_cast<const float2*>(X.att_ml + (size_t)s * SK_ROWS_CAP * nq * 2 + off_ml);
            o.mv[i] = mlv.x; o.lv[i] = mlv.y;
            o.ov[i] = *reinterpret_cast<const f32x8*>(X.parts + (size_t)s * SK_ROWS_CAP * K + off_o);
        }
    }
}
__device__ __forceinline__ void sk_partial_att_half(const SkRawHalf& o, int s_lo, float& M, float& den, f32x8& acc) {
    M = -INFINITY;
#pragma unroll
    for (int i = 0; i < SK_HALFSPLIT; i++) if (s_lo + i < o.ns) M = fmaxf(M, o.mv[i]);
    acc = (f32x8){0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    den = 0.f;
#pragma unroll
    for (int i = 0; i < SK_HALFSPLIT; i++) {
        if (s_lo + i < o.ns) {
            const float w = __expf(o.mv[i] - M);
            den += w * o.lv[i];
            acc += w * o.ov[i];
        }
    }
}
__device__ __forceinline__ void sk_issue_x(const SkinnyX& X, int r, int K, int k, SkRaw<false>& o) {
    const unsigned off = (unsigned)(r * K + k);
    o.v = *reinterpret_cast<const f32x8*>(X.base + off);
    if (X.np > 0) {
#pragma unroll
        for (int i = 0; i < SK_MAXNP; i++) {             // all partial loads in flight together; fixed summation order
            const int ii = i < X.np ? i : X.np - 1;
            o.p[i] = *reinterpret_cast<const f32x8*>(X.parts + (size_t)ii * SK_ROWS_CAP * K + off);
        }
    }
}
__device__ __forceinline__ f32x8 sk_finish_x(const SkinnyX& X, const SkRaw<false>& o) {
    f32x8 v = o.v;
    if (X.np > 0) {
#pragma unroll
        for (int i = 0; i < SK_MAXNP; i++)
            if (i < X.np) v += o.p[i];
    }
    return v;
}
template <bool ATT>
__device__ __forceinline__ f32x8 sk_load_x(const SkinnyX& X, int r, int K, int k) {
    SkRaw<ATT> raw;
    sk_issue_x(X, r, K, k, raw);
    return sk_finish_x(X, raw);
}
struct SkNoHook { __device__ __forceinline__ void operator()() const {} };

template <int NB>
__device__ __host__ constexpr int sk_xstage_bytes(int nks) { return nks * NB * 2 * 1024; }

// Leaves the reduced tile in LDS: res[NWR*16 features][NB*16+1 rows]; returns its address.
// `tile` = this wave's 16-feature row tile of W.
// `issued` runs right after the last weight load has been issued and before anything is consumed: the place for a caller's
// dependent loads (k_qkv: position -> RoPE table) that must not delay the stream.
// KEEP (PRE kernels that run several feature tiles per block): the reduction buffers sit behind the x stage also for 32 rows, so the
// staged operand survives the call; `staged` = the operand of an earlier call is still there (no DMA, no wait for it).
template <int NB, int NWR, int NWK, int MAXKS, bool ATT = false, bool PRE = false, class HOOK = SkNoHook, bool KEEP = false>
__device__ __forceinline__ float* skinny_core(const uint16_t* __restrict__ W, int tile, int KS, int rows, int K,
                                               const SkinnyX& X, char* smem, HOOK issued = HOOK(), bool staged = false) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    // wave-uniform values are made provably uniform (readfirstlane): the weight addresses then live in scalar registers and the
    // loads use the saddr + 32-bit lane offset form instead of per-lane 64-bit address arithmetic
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    tile = __builtin_amdgcn_readfirstlane(tile);
    const int wr = wave % NWR;
    const int wk = wave / NWR;
    constexpr int nthreads = 64 * NWR * NWK;
    // 32-bit on purpose (KS * gridDim.y is tiny): a 64-bit division is ~150 scalar instructions in front of the first load
    const int ks0 = gridDim.y == 1 ? 0 : (int)(((unsigned)KS * blockIdx.y) / gridDim.y);
    const int ks1 = gridDim.y == 1 ? KS : (int)(((unsigned)KS * (blockIdx.y + 1)) / gridDim.y);
    const int nks = ks1 - ks0;
    const int w0 = ks0 + (nks * wk) / NWK;
    const int w1 = ks0 + (nks * (wk + 1)) / NWK;

    // 0. (PRE) the operand prepared by k_prep goes global -> LDS by DMA, issued BEFORE the weight stream so that the two are in
    //    flight together and no VGPR / ds_write work is spent on it: this block's K slice [ks0, ks1) of the hi / lo planes
    if (PRE && !staged) {
        const char* src = reinterpret_cast<const char*>(X.pre) + (size_t)ks0 * NB * 2 * 1024 + lane * 16;
        for (int pc = wave; pc < nks * NB * 2; pc += NWR * NWK)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)pc * 1024),
                                             (__attribute__((address_space(3))) void*)(smem + pc * 1024), 16, 0, 0);
    }
    SK_STAMP_DECL;
    SK_STAMP(0);
    float rs_def = 1.f;                                          // deferred RMS scale of the single row (batch-1 fast path)
    // 1. this thread's first operand item, THEN every weight fragment of the wave.  Loads return in issue order per wave: with the
    //    (L2-resident, tiny) operand issued first it arrives after one L2 round trip and the fold / RMS / split below runs while the
    //    weight stream is still in flight; issued after the weights it would only arrive once the whole stream has landed.
    const int k8n = nks * 4;
    const int nitems = rows * k8n;
    // item -> row without an integer division (~30 instructions): (2 it + 1) / (2 k8n) is never within 8e-4 of an integer
    const float inv2k = 0.5f * __builtin_amdgcn_rcpf((float)k8n);
    auto row_of = [rows, inv2k](int it) { return rows == 1 ? 0 : (int)((float)(2 * it + 1) * inv2k); };
    f32x8 v0 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    SkRaw<ATT> raw0;
    // one-row O-projection (ATT, 256 threads, <= 128 items): threads 0..127 fold the first half of the split slots, threads
    // 128..255 the second half; the other shapes fold every slot in the item's own thread
    const bool half_fold = ATT && !PRE && nthreads == 256 && rows == 1 && nitems <= 128;
    SkRawHalf rawh;
    const int hgrp = tid >> 7, htid = tid & 127;
    if (ATT && half_fold) {
        if (htid < nitems) sk_issue_att_half(X, 0, K, ks0 * 32 + htid * 8, hgrp * SK_HALFSPLIT, rawh);
    }
    const bool has0 = !PRE && !half_fold && tid < nitems;
    const int r0 = row_of(tid), c0 = ks0 * 32 + (tid - r0 * k8n) * 8;
    f32x8 g0;                                                     // RMSNorm weight of the item (unused without norm)
    if (has0) {
        sk_issue_x(X, r0, K, c0, raw0);
        // unconditional (a valid dummy address without norm): a conditional load would make the in-flight count unknown
        g0 = *reinterpret_cast<const f32x8*>(X.norm_w ? X.norm_w + c0 : (ATT ? X.parts : X.base) + c0);
    }
    const char* wbase = reinterpret_cast<const char*>(W) + ((size_t)tile * KS + w0) * 1024;
    const unsigned wlane = lane * 16;
    s16x8 abuf[MAXKS];
    // exactly MAXKS loads per wave, unconditionally (slots beyond the wave's share re-read its last fragment): with a
    // compile-time count hipcc can wait for the OLDER operand loads with a counted vmcnt and leave the weight stream in flight;
    // a data-dependent number of loads forces vmcnt(0) and serialises the prologue behind the whole stream
    const int nw = w1 - w0;
#pragma unroll
    for (int i = 0; i < MAXKS; i++)
        abuf[i] = __builtin_nontemporal_load(reinterpret_cast<const s16x8*>(wbase + (size_t)(i < nw ? i : (nw > 0 ? nw - 1 : 0)) * 1024 + wlane));
    // the scheduler must not pull the operand fold (and its waits) above the weight loads, nor sink the loads below it
    __builtin_amdgcn_sched_barrier(0);
    issued();
    if (has0) v0 = sk_finish_x(X, raw0);
    if (ATT && half_fold) {
        float hM = -INFINITY, hden = 0.f;
        f32x8 hacc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (htid < nitems) sk_partial_att_half(rawh, hgrp * SK_HALFSPLIT, hM, hden, hacc);
        // exchange through the (still unused) reduction area behind the x stage: [item][10] floats
        float* xch = reinterpret_cast<float*>(smem + sk_xstage_bytes<NB>(nks) + 32 * sizeof(float));
        if (hgrp == 1 && htid < nitems) {
            float* d = xch + htid * 10;
            d[0] = hM; d[1] = hden;
#pragma unroll
            for (int e = 0; e < 8; e++) d[2 + e] = hacc[e];
        }
        __syncthreads();
        if (hgrp == 0 && htid < nitems) {
            const float* d = xch + htid * 10;
            const float M1 = d[0], M = fmaxf(hM, M1);
            const float w0 = __expf(hM - M), w1 = __expf(M1 - M);         // a group without live slots has max = -inf: weight 0
            const float den = hden * w0 + d[1] * w1;
#pragma unroll
            for (int e = 0; e < 8; e++) v0[e] = (hacc[e] * w0 + d[2 + e] * w1) * (1.f / den);
        }
        // (no second barrier: the area becomes the reduction buffer of step 4 only after the staging barrier below)
    }

  if (PRE) {
    if (!staged) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
  } else {
    // 2. fold / normalise / split the block's x slice [ks0*32, ks1*32) into LDS (B-operand order)
    //    item = (row r, group of 8 columns); the first item of every thread stays in registers.
    //    Up to 16 rows (NB == 1) the RMS statistic is DEFERRED: W (g . x) rs = rs (W (g . x)), so g . x is staged without waiting
    //    for the row's sum of squares, which travels through LDS under the staging barrier and scales the reduced tile in step 4
    //    (one barrier and one dependent LDS round trip less between the operand's arrival and the MFMAs).
    constexpr bool DEFER = NB == 1;
    float* isq = reinterpret_cast<float*>(DEFER ? smem + sk_xstage_bytes<NB>(nks) + 32 * sizeof(float) + NWK * NWR * NB * 1024 +
                                                      NWR * 16 * (NB * 16 + 1) * sizeof(float)
                                                : smem);                          // [nitems]; NB > 1: aliases the x stage
    float* rstd = reinterpret_cast<float*>(smem + sk_xstage_bytes<NB>(nks));      // [32]
    SK_STAMP(1);                                                 // loads issued
    const bool one_row = rows == 1 && nitems <= 128;             // batch-1 decode: the row's items sit in waves 0 and 1
    float rs_one = 0.f;
    if (X.norm_w && one_row) {
        // sum of squares straight from the registers: DPP wave sums, two partials through LDS
        float sq = ((v0[0] * v0[0] + v0[1] * v0[1]) + (v0[2] * v0[2] + v0[3] * v0[3])) + ((v0[4] * v0[4] + v0[5] * v0[5]) + (v0[6] * v0[6] + v0[7] * v0[7]));
        if (wave < 2) { sq = wave_sum(sq); if (lane == 0) rstd[wave] = sq; }
        if (!DEFER) {
            __syncthreads();
            rs_one = rsqrtf((rstd[0] + rstd[1]) / (float)K + X.eps);
        }
        SK_STAMP(2);                                             // operand arrived
    } else if (X.norm_w && !DEFER) {                             // (DEFER: the sums of squares are taken in the staging pass below: each item is folded once)
        for (int it = tid; it < nitems; it += nthreads) {
            f32x8 v = v0;
            if (it != tid) { const int r = row_of(it), k8 = it - r * k8n; v = sk_load_x<ATT>(X, r, K, ks0 * 32 + k8 * 8); }
            isq[it] = ((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3])) + ((v[4] * v[4] + v[5] * v[5]) + (v[6] * v[6] + v[7] * v[7]));
        }
        if (!DEFER) {
            __syncthreads();
            for (int r = wave; r < rows; r += NWR * NWK) {
                float s = 0.f;
                for (int i = lane; i < k8n; i += 64) s += isq[r * k8n + i];
                s = wave_sum(s);
                if (lane == 0) rstd[r] = rsqrtf(s / (float)K + X.eps);
            }
            __syncthreads();
        }
    }
    for (int it = tid; it < nitems; it += nthreads) {
        const int r = row_of(it), k8 = it - r * k8n;
        const int k = ks0 * 32 + k8 * 8;
        f32x8 v = v0, g = g0;
        if (it != tid) { v = sk_load_x<ATT>(X, r, K, k); if (X.norm_w) g = *reinterpret_cast<const f32x8*>(X.norm_w + k); }
        if (X.x_out && blockIdx.x == 0) *reinterpret_cast<f32x8*>(X.x_out + (unsigned)(r * K + k)) = v;
        if (DEFER && X.norm_w && !one_row)
            isq[it] = ((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3])) + ((v[4] * v[4] + v[5] * v[5]) + (v[6] * v[6] + v[7] * v[7]));
        if (X.norm_w) v = DEFER ? g * v : g * (v * (one_row ? rs_one : rstd[r]));
        bf16x8 hi, lo;
        split8(v, hi, lo);
        const int s = k8 >> 2, hq = k8 & 3, t = r >> 4;
        bf16x8* dst = reinterpret_cast<bf16x8*>(smem + ((size_t)(s * NB + t) * 2) * 1024) + hq * 16 + (r & 15);
        dst[0] = hi;
        dst[64] = lo;
    }
    __syncthreads();
    if (DEFER && X.norm_w) {                                     // row statistics -> rstd[row]; consumed after the barrier of step 4
        if (one_row) {
            rs_def = rsqrtf((rstd[0] + rstd[1]) / (float)K + X.eps);
        } else {
            for (int r = wave; r < rows; r += NWR * NWK) {
                float s = 0.f;
                for (int i = lane; i < k8n; i += 64) s += isq[r * k8n + i];
                s = wave_sum(s);
                if (lane == 0) rstd[16 + r] = rsqrtf(s / (float)K + X.eps);       // [16..31]: [0..1] may still be read as partials
            }
        }
    }
  }

    SK_STAMP(3);                                                 // B operand staged in LDS
    // 3. MFMAs: A = weights (registers), B = x hi / lo (LDS)
    f32x4 acc[NB];
#pragma unroll
    for (int t = 0; t < NB; t++) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < MAXKS; i++) {
        if (w0 + i < w1) {
            const int s = w0 + i - ks0;
            const bf16x8 a = __builtin_bit_cast(bf16x8, abuf[i]);
#pragma unroll
            for (int t = 0; t < NB; t++) {
                const bf16x8* xb = reinterpret_cast<const bf16x8*>(smem + ((size_t)(s * NB + t) * 2) * 1024) + lane;
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, xb[0], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, xb[64], acc[t], 0, 0, 0);
            }
        }
    }
    SK_STAMP(4);                                                 // weights arrived, MFMAs issued
    // 4. reduce the NWK K-slices through LDS, leave the tile in res[feature][row].  The reduction buffers sit BEHIND the x
    //    stage (not on top of it) for up to 16 rows, so no barrier is needed between the MFMAs and the partial-sum writes.
    if (NB > 1 && !KEEP) __syncthreads();                         // 32-row stage is too large to keep: reuse it (x stage fully read)
    char* rbase = (NB > 1 && !KEEP) ? smem : smem + sk_xstage_bytes<NB>(nks) + 32 * sizeof(float);
    f32x4* red = reinterpret_cast<f32x4*>(rbase);                                   // [NWK][NWR][NB][64]
    float* res = reinterpret_cast<float*>(rbase + NWK * NWR * NB * 1024);           // [NWR*16][NB*16+1]
#pragma unroll
    for (int t = 0; t < NB; t++) red[((wk * NWR + wr) * NB + t) * 64 + lane] = acc[t];
    __syncthreads();
    if (wk == 0) {
#pragma unroll
        for (int t = 0; t < NB; t++) {
            f32x4 v = red[((0 * NWR + wr) * NB + t) * 64 + lane];
#pragma unroll
            for (int q = 1; q < NWK; q++) v += red[((q * NWR + wr) * NB + t) * 64 + lane];
            const int b = 16 * t + (lane & 15);
            const int n0 = wr * 16 + 4 * (lane >> 4);
            if (NB == 1 && !PRE && X.norm_w) {                    // deferred RMS scale of row b (rows >= `rows` are never read)
                const float* rstd_ = reinterpret_cast<const float*>(smem + sk_xstage_bytes<NB>(nks));
                v *= (rows == 1 && nitems <= 128) ? rs_def : rstd_[16 + (lane & 15)];
            }
#pragma unroll
            for (int r = 0; r < 4; r++) res[(n0 + r) * (NB * 16 + 1) + b] = v[r];
        }
    }
    __syncthreads();
    SK_STAMP(5);                                                 // tile reduced
    SK_STAMP_FLUSH;
    return res;
}

// Operand preparation for many-row launches (rows > 16): fold + RMSNorm weight + hi/lo split ONCE per row instead of once per
// block, written in the LDS B-operand order the PRE kernels copy.  One block per row; K % 8 == 0, K <= 8 * 256.
// RMSNorm in the DEFERRED form of the <= 16-row kernels (skinny_core, step 2): the planes hold g . x, the row's sum of squares goes to
// sq_out[row] and the consumer scales its OUTPUTS by rsqrt(sq / K + eps) -- W (g . x) rs, the arithmetic every other form of the decode
// step uses.  (Until round 5 this kernel staged g . (x rs): the normalised value was rounded to its hi / lo planes, a 2^-17 difference
// against the other forms where everything else differs by fp32 round-off.)  The sum of squares runs in skinny_core's order: per item
// ((v0^2 + v1^2) + (v2^2 + v3^2)) + ((v4^2 + v5^2) + (v6^2 + v7^2)), lane l of one wave adds items l, l + 64, .., then wave_sum.
template <bool ATT>
__global__ __launch_bounds__(256) void k_prep(SkinnyX X, int K, uint16_t* pre, float* sq_out) {
    __shared__ float isq[256];
    const int r = blockIdx.x, tid = threadIdx.x;
    const int nitem = K / 8;
    for (int it = tid; it < nitem; it += 256) {
        f32x8 v = sk_load_x<ATT>(X, r, K, it * 8);
        if (X.x_out) *reinterpret_cast<f32x8*>(X.x_out + (size_t)r * K + it * 8) = v;
        if (X.norm_w) {
            isq[it] = ((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3])) + ((v[4] * v[4] + v[5] * v[5]) + (v[6] * v[6] + v[7] * v[7]));
            v = *reinterpret_cast<const f32x8*>(X.norm_w + it * 8) * v;
        }
        bf16x8 hi, lo;
        split8(v, hi, lo);
        const int s = it >> 2, hq = it & 3, t = r >> 4;
        bf16x8* dst = reinterpret_cast<bf16x8*>(pre + ((size_t)(s * 2 + t) * 2) * 512) + hq * 16 + (r & 15);
        dst[0] = hi;
        dst[64] = lo;
    }
    if (X.norm_w) {
        __syncthreads();
        if (tid < 64) {
            float s = 0.f;
            for (int i = tid; i < nitem; i += 64) s += isq[i];
            s = wave_sum(s);
            if (tid == 0) sq_out[r] = s;
        }
    }
}

template <int NB, int NWR, int NWK>
static inline size_t skinny_smem_bytes_keep(int nks_block) {      // KEEP layout: x stage, then the reduction buffers
    return (size_t)nks_block * NB * 2 * 1024 + 32 * sizeof(float) + (size_t)NWK * NWR * NB * 1024 + (size_t)NWR * 16 * (NB * 16 + 1) * sizeof(float);
}
template <int NB, int NWR, int NWK>
static inline size_t skinny_smem_bytes(int nks_block) {
    const size_t xs = (size_t)nks_block * NB * 2 * 1024 + 32 * sizeof(float);
    const size_t rr = (size_t)NWK * NWR * NB * 1024 + (size_t)NWR * 16 * (NB * 16 + 1) * sizeof(float);
    // NB == 1: reduction buffers behind the x stage, then the per-item sums of squares of the deferred RMS (16 rows x 4 nks items)
    return NB > 1 ? (xs > rr ? xs : rr) : xs + rr + 1024 + (size_t)16 * nks_block * 4 * sizeof(float);
}
