// EXPERIMENT, NOT BUILT INTO libcv2amd.so: never run on hardware.  Kept beside tools/micro/edge.hip, whose measurements (profiles/r2_edge_handoff_chain.txt)
// showed that the hand-off chain alone costs 22 us per layer at these volumes, i.e. no better than the five launches it would replace.
// One decode step of the Qwen2-0.5B speech-token LM for ONE sequence as a single persistent launch (replaces the 121 dependent
// launches of run_layers<1> at rows == 1; reference: HFBackbone.forward_one_step, cosyvoice/llm/llm.py:107-117, one token).
//
// Why: at batch 1 a layer is 29 MB of weights behind five all-to-all dependencies; as separate launches every dependency costs a
// kernel boundary + a cold first load + the ramp of a new weight stream (24.6 us per layer measured, 15.6 % of the HBM roof).
// Here every CU keeps its slice of the NEXT layer's weights in flight in registers while it waits for its input, so a dependency
// costs one hand-off and the weights are already there (cdna_hip_programming.md §5.6; MI355X_MICROARCH.md "prefetch-credit").
//
// Structure: grid = 256 workgroups (one per CU, all resident), 256 threads.  Fixed roles per layer:
//   CU   0.. 35  Q : one (head, half) unit of the QKV projection (2 row tiles x 28 fragments) + 2 down-projection units
//   CU  36.. 91  O : one 16-row tile of the O projection, builds x_mid = x + o                + 2-3 down-projection units
//   CU  92..243  G : two (gate, up) tile pairs, SwiGLU
//   CU 244..255  A : decode attention of one (kv group, 128-key split); K / V tiles of the next layer prefetched into registers
//   after the last layer every CU takes 1-2 row tiles of llm_decoder (logits).
// A down-projection unit = (16-row tile, quarter of K): its 16 partial sums go to slot `quarter`; consumers fold the four slots.
// Every weight fragment (1 KiB MFMA A operand, packed layout of include/cv2_amd.h) is read exactly once per step with non-temporal
// 16 B/lane loads, K split over the four waves of the workgroup, x split hi + lo in two B columns of ONE MFMA.
//
// Hand-offs (Guideline 16, form R2): every value that crosses CUs is an 8-byte granule {tag = launch epoch, fp32 value} written by
// one agent-scope relaxed atomic store (sc1) and read by agent-scope relaxed atomic loads until the tag matches; buffers are per
// layer, so a granule is written once per launch and a stale one carries the previous epoch.  No fences, no flags.  The epoch lives
// in device memory and is advanced by k_sample between two launches (a kernel argument would be frozen under graph replay).
// Every spin is bounded (wall clock): on a timeout the workgroup raises the abort word, every other workgroup sees it in its own
// spin, the launch drains, and k_sample turns it into CV2_ST_ERR = 3.
#pragma once
#include "common.h"

typedef unsigned long long u64;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define PS_H 896
#define PS_KSH 28            // k-steps (32 columns) of the hidden size
#define PS_NQ 14
#define PS_NKV 2
#define PS_I 4864
#define PS_KSI 152
#define PS_NC 4              // K quarters of the down projection
#define PS_DKS 38            // k-steps per quarter
#define PS_NSPL 6            // key splits per kv group
#define PS_KSPL 128          // keys per split and round
#define PS_G 256             // workgroups
#define PS_QKVN 1152         // q 896 | k_new 128 | v_new 128
#define PS_APN 924           // per split: 14 heads x 64 unnormalised outputs | 14 maxima | 14 sums
// granule offsets inside one layer's block
#define PG_XM 0
#define PG_DP (PG_XM + PS_H)
#define PG_QKV (PG_DP + PS_NC * PS_H)
#define PG_AP (PG_QKV + PS_QKVN)
#define PG_HB (PG_AP + PS_NSPL * PS_APN)
#define PG_LAYER (PG_HB + PS_I)
#define PS_TIMEOUT_TICKS 20000000ull      // 200 ms of s_memrealtime (100 MHz)

struct StepArgs {
    const cv2_llm_layer* layers;          // DEVICE copy of the layer table
    int n_layers, vocab_pad;
    const float* final_norm; const uint16_t* wdec; const float* bdec;
    const float* cosT; const float* sinT;
    float eps;
    const int* state;                     // slot 0
    const float* xin;                     // [hidden] input embedding of this step (k_sample's x_next)
    float* kc; float* vc; size_t cache_l; int max_pos;
    float* logits;
    u64* gran;                            // [n_layers][PG_LAYER]
    const unsigned* epoch; unsigned* abort_flag;
};

__device__ __forceinline__ u64 pg_load(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void pg_store(u64* p, unsigned epoch, float v) {
    __hip_atomic_store(p, ((u64)epoch << 32) | (u64)__builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct PsCtx {
    unsigned epoch;
    u64 t0;
    unsigned* abort_flag;
    int* fail;                // LDS
    int tid, lane, wave;
};

// spin helper: true = give up (timeout or another workgroup aborted)
__device__ __forceinline__ bool ps_giveup(const PsCtx& c, unsigned spins) {
    if ((spins & 127u) != 127u) return false;
    if (__builtin_amdgcn_s_memrealtime() - c.t0 > PS_TIMEOUT_TICKS || __hip_atomic_load(c.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
        if (c.lane == 0) { __hip_atomic_store(c.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); *c.fail = 1; }
        return true;
    }
    return false;
}

// Wave 0 waits until ONE granule of the region carries the epoch (a cheap sentinel: the other three waves sleep at the barrier
// instead of sweeping, MI355X_MICROARCH.md "polling-cost"), then the workgroup sweeps.  Returns false on abort.
__device__ __forceinline__ bool ps_wait_sentinel(const PsCtx& c, const u64* g) {
    if (c.wave == 0) {
        for (unsigned spins = 0;; spins++) {
            const u64 v = pg_load(g);
            if ((unsigned)(v >> 32) == c.epoch) break;
            if (ps_giveup(c, spins)) break;
            __builtin_amdgcn_s_sleep(2);
        }
    }
    __syncthreads();
    return *c.fail == 0;
}

// NPT granules per thread (index tid + 256 j, valid while < n): re-read until every tag of the wave matches.
template <int NPT>
__device__ __forceinline__ void ps_sweep(const PsCtx& c, const u64* g, int n, float (&v)[NPT]) {
    for (unsigned spins = 0;; spins++) {
        bool ok = true;
        u64 raw[NPT];
#pragma unroll
        for (int j = 0; j < NPT; j++) { const int i = c.tid + 256 * j; raw[j] = pg_load(g + (i < n ? i : n - 1)); }
#pragma unroll
        for (int j = 0; j < NPT; j++) { ok &= (unsigned)(raw[j] >> 32) == c.epoch; v[j] = __builtin_bit_cast(float, (unsigned)raw[j]); }
        if (__all(ok)) return;
        if (ps_giveup(c, spins)) return;
    }
}

// ---- LDS carve (bytes) ----------------------------------------------------------------------------------------------
#define PL_XHI 0                          // bf16 [1216]
#define PL_XLO 2560
#define PL_RED 5120                       // float [4 items][4 waves][16]
#define PL_SQ (PL_RED + 4 * 4 * 16 * 4)   // float [4] per-wave sums of squares
#define PL_FAIL (PL_SQ + 16)
#define PL_MISC (PL_FAIL + 16)            // role scratch (attention: q 7x64, k_new 64, v_new 64, scores 7x128, ml 7x2, po 16x7x64)
#define PL_BYTES (PL_MISC + (448 + 128 + 7 * 128 + 16 + 16 * 448) * 4)

// stage g[k] * x[k] (or x[k]) as hi / lo bf16 and leave this wave's sum of squares in LDS
__device__ __forceinline__ void ps_stage(char* smem, int k, float x, float gx) {
    uint16_t hi, lo;
    split_bf16(gx, hi, lo);
    reinterpret_cast<uint16_t*>(smem + PL_XHI)[k] = hi;
    reinterpret_cast<uint16_t*>(smem + PL_XLO)[k] = lo;
}

// one row tile x NF k-steps of this wave: A = weight fragments (registers), B column 0 = x hi, column 1 = x lo
template <int NF>
__device__ __forceinline__ f32x4 ps_mfma(const s16x8 (&w)[NF], int n, const char* smem, int ks0, int lane) {
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
    const char* xb = smem + ((lane & 15) == 1 ? PL_XLO : PL_XHI) + (ks0 * 32 + 8 * (lane >> 4)) * 2;
#pragma unroll
    for (int i = 0; i < NF; i++) {
        if (i < n) {
            const bf16x8 b = *reinterpret_cast<const bf16x8*>(xb + i * 64);
            const bf16x8 a = __builtin_bit_cast(bf16x8, w[i]);
            if (i & 1) a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, a1, 0, 0, 0);
            else a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, a0, 0, 0, 0);
        }
    }
    f32x4 r = a0 + a1;
#pragma unroll
    for (int e = 0; e < 4; e++) r[e] += dpp_mov_f32<0xB1, 0xf>(0.f, r[e]);      // column 0 (hi) + column 1 (lo): quad_perm [1,0,3,2]
    return r;                                                                     // lanes 0, 16, 32, 48: rows 4 (lane >> 4) + e
}
__device__ __forceinline__ void ps_put_partial(char* smem, int item, int wave, int lane, const f32x4 r) {
    if ((lane & 15) == 0) *reinterpret_cast<f32x4*>(smem + PL_RED + ((item * 4 + wave) * 16 + 4 * (lane >> 4)) * 4) = r;
}
__device__ __forceinline__ float ps_get(const char* smem, int item, int row) {
    const float* p = reinterpret_cast<const float*>(smem + PL_RED) + item * 64 + row;
    return (p[0] + p[16]) + (p[32] + p[48]);
}
template <int NF>
__device__ __forceinline__ void ps_load_w(s16x8 (&w)[NF], const uint16_t* W, size_t frag0, int n, int lane) {
    const char* base = reinterpret_cast<const char*>(W) + frag0 * 1024 + lane * 16;
#pragma unroll
    for (int i = 0; i < NF; i++) w[i] = __builtin_nontemporal_load(reinterpret_cast<const s16x8*>(base + (size_t)(i < n ? i : (n > 0 ? n - 1 : 0)) * 1024));
}

// x of a layer = x_mid of the previous one + its four down-projection slots (fixed order), or the step's input embedding
template <int NPT>
__device__ __forceinline__ void ps_gather_x(const PsCtx& c, const StepArgs& a, int layer, float (&x)[NPT], int e0, int n) {
    // elements e0 + tid + 256 j, j < NPT, valid while tid + 256 j < n
    if (layer == 0) {
#pragma unroll
        for (int j = 0; j < NPT; j++) { const int i = c.tid + 256 * j; x[j] = a.xin[e0 + (i < n ? i : n - 1)]; }
        return;
    }
    const u64* lg = a.gran + (size_t)(layer - 1) * PG_LAYER;
    for (unsigned spins = 0;; spins++) {
        bool ok = true;
        u64 raw[NPT][1 + PS_NC];
#pragma unroll
        for (int j = 0; j < NPT; j++) {
            const int i = c.tid + 256 * j, e = e0 + (i < n ? i : n - 1);
            raw[j][0] = pg_load(lg + PG_XM + e);
#pragma unroll
            for (int q = 0; q < PS_NC; q++) raw[j][1 + q] = pg_load(lg + PG_DP + q * PS_H + e);
        }
#pragma unroll
        for (int j = 0; j < NPT; j++) {
            float s = __builtin_bit_cast(float, (unsigned)raw[j][0]);
            ok &= (unsigned)(raw[j][0] >> 32) == c.epoch;
#pragma unroll
            for (int q = 0; q < PS_NC; q++) { ok &= (unsigned)(raw[j][1 + q] >> 32) == c.epoch; s += __builtin_bit_cast(float, (unsigned)raw[j][1 + q]); }
            x[j] = s;
        }
        if (__all(ok)) return;
        if (ps_giveup(c, spins)) return;
    }
}

// RMSNorm operand of the hidden size into LDS (g . x as hi / lo; the row statistic is applied to the outputs): x[j] = element tid + 256 j
__device__ __forceinline__ void ps_stage_norm(const PsCtx& c, char* smem, const float (&x)[4], const float* gw) {
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int k = c.tid + 256 * j;
        if (k < PS_H) { sq += x[j] * x[j]; ps_stage(smem, k, x[j], gw[k] * x[j]); }
    }
    sq = wave_sum(sq);
    if (c.lane == 0) reinterpret_cast<float*>(smem + PL_SQ)[c.wave] = sq;
}
__device__ __forceinline__ float ps_rstd(const char* smem, float eps) {
    const float* s = reinterpret_cast<const float*>(smem + PL_SQ);
    return rsqrtf(((s[0] + s[1]) + (s[2] + s[3])) / (float)PS_H + eps);
}

__global__ __launch_bounds__(256, 1) void k_step(StepArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    PsCtx c;
    c.tid = threadIdx.x; c.lane = c.tid & 63; c.wave = __builtin_amdgcn_readfirstlane(c.tid >> 6);
    c.t0 = __builtin_amdgcn_s_memrealtime();
    c.abort_flag = a.abort_flag;
    c.fail = reinterpret_cast<int*>(smem + PL_FAIL);
    c.epoch = *a.epoch;
    if (c.tid == 0) *c.fail = 0;
    const int b = blockIdx.x;
    const int pos = a.state[CV2_ST_POS];
    const int L = pos + 1;                                         // keys this step attends
    const int wave = c.wave, lane = c.lane, tid = c.tid;
    __syncthreads();
    const int NL = a.n_layers;

    if (b < 92) {
        // ============================================================ roles Q (b < 36) and O (36 <= b < 92), both with down units
        const bool isQ = b < 36;
        // down units of this CU: Q role: 2b, 2b + 1; O role: 72 + j, 72 + j + 56, 72 + j + 112 (the last only for j < 40)
        const int j = b - 36;
        const int nd = isQ ? 2 : (j < 40 ? 3 : 2);
        int du[3];
        du[0] = isQ ? 2 * b : 72 + j; du[1] = isQ ? 2 * b + 1 : 72 + j + 56; du[2] = isQ ? 0 : 72 + j + 112;
        // unit u = (tile u / 4, quarter u % 4); wave w takes k-steps [w0, w0 + nk) of the quarter: 10, 10, 9, 9
        const int dk0 = wave < 2 ? 10 * wave : 20 + 9 * (wave - 2), dkn = wave < 2 ? 10 : 9;
        s16x8 wa0[7], wa1[7];  // Q: tiles (head, half) and (head, half + 2), 7 k-steps per wave each; O: tile j in wa0
        s16x8 wd[3][10];
        const int head = b >> 1, half = b & 1;
        auto load_a = [&](int l) {
            const cv2_llm_layer& Lw = a.layers[l];
            if (isQ) {
                ps_load_w<7>(wa0, Lw.wqkv, (size_t)(head * 4 + half) * PS_KSH + 7 * wave, 7, lane);
                ps_load_w<7>(wa1, Lw.wqkv, (size_t)(head * 4 + half + 2) * PS_KSH + 7 * wave, 7, lane);
            } else {
                ps_load_w<7>(wa0, Lw.wo, (size_t)j * PS_KSH + 7 * wave, 7, lane);
            }
        };
        auto load_d = [&](int l) {
            const cv2_llm_layer& Lw = a.layers[l];
#pragma unroll
            for (int u = 0; u < 3; u++)
                if (u < nd) ps_load_w<10>(wd[u], Lw.wdown, (size_t)(du[u] >> 2) * PS_KSI + (du[u] & 3) * PS_DKS + dk0, dkn, lane);
        };
        load_a(0);
        load_d(0);
        // epilogue operands of the Q role (bias, RoPE of this step's position): thread t < 32 owns feature f of the unit
        const int f = half * 16 + ((tid >> 4) & 1) * 32 + (tid & 15);
        float cs = 1.f, sn = 0.f;
        if (isQ && tid < 32) { cs = a.cosT[pos * 32 + (f & 31)]; sn = a.sinT[pos * 32 + (f & 31)]; }
        for (int l = 0; l < NL; l++) {
            const cv2_llm_layer& Lw = a.layers[l];
            u64* lg = a.gran + (size_t)l * PG_LAYER;
            if (isQ) {
                // ---- QKV: x -> RMSNorm -> 32 features of one head -> bias, RoPE -> q / k_new / v_new granules (+ KV cache)
                if (l > 0 && !ps_wait_sentinel(c, a.gran + (size_t)(l - 1) * PG_LAYER + PG_DP + 3 * PS_H + PS_H - 1)) return;
                float x[4];
                ps_gather_x<4>(c, a, l, x, 0, PS_H);
                ps_stage_norm(c, smem, x, Lw.ln1);
                __syncthreads();
                if (*c.fail) return;
                const f32x4 r0 = ps_mfma<7>(wa0, 7, smem, 7 * wave, lane);
                const f32x4 r1 = ps_mfma<7>(wa1, 7, smem, 7 * wave, lane);
                ps_put_partial(smem, 0, wave, lane, r0);
                ps_put_partial(smem, 1, wave, lane, r1);
                __syncthreads();
                if (tid < 32) {
                    const float rs = ps_rstd(smem, a.eps);
                    const int w = (tid >> 4) & 1, i16 = tid & 15;
                    float v = ps_get(smem, w, i16) * rs + Lw.bqkv[head * 64 + f];
                    if (head < PS_NQ + PS_NKV) {                   // rotate-half RoPE on q and k heads
                        const float vp = ps_get(smem, 1 - w, i16) * rs + Lw.bqkv[head * 64 + (f ^ 32)];
                        v = (f < 32) ? (v * cs - vp * sn) : (v * cs + vp * sn);
                    }
                    pg_store(lg + PG_QKV + head * 64 + f, c.epoch, v);
                    if (head >= PS_NQ) {
                        float* cache = head < PS_NQ + PS_NKV ? a.kc : a.vc;
                        const int gk = head < PS_NQ + PS_NKV ? head - PS_NQ : head - PS_NQ - PS_NKV;
                        cache[(size_t)l * a.cache_l + ((size_t)gk * a.max_pos + pos) * 64 + f] = v;
                    }
                }
                if (l + 1 < NL) load_a(l + 1);
            } else {
                // ---- O projection of tile j: combine the key splits, 16 outputs, + residual -> x_mid granules
                const int nlive = min(PS_NSPL, (L + PS_KSPL - 1) / PS_KSPL);
                if (!ps_wait_sentinel(c, lg + PG_AP + (size_t)(nlive - 1) * PS_APN + PS_APN - 1)) return;
                {
                    // element e = tid + 256 jj (< 896), head e >> 6: o_s[e], m_s[head], l_s[head] of every live split
                    float ov[4][PS_NSPL], mv[4][PS_NSPL], lv[4][PS_NSPL];
                    for (unsigned spins = 0;; spins++) {
                        bool ok = true;
                        u64 ro[4][PS_NSPL], rm[4][PS_NSPL], rl[4][PS_NSPL];
#pragma unroll
                        for (int jj = 0; jj < 4; jj++) {
                            const int e = min(tid + 256 * jj, PS_H - 1), hd = e >> 6;
#pragma unroll
                            for (int s = 0; s < PS_NSPL; s++) {
                                if (s < nlive) {
                                    const u64* ap = lg + PG_AP + (size_t)s * PS_APN;
                                    ro[jj][s] = pg_load(ap + e); rm[jj][s] = pg_load(ap + 896 + hd); rl[jj][s] = pg_load(ap + 910 + hd);
                                }
                            }
                        }
#pragma unroll
                        for (int jj = 0; jj < 4; jj++)
#pragma unroll
                            for (int s = 0; s < PS_NSPL; s++)
                                if (s < nlive) {
                                    ok &= (unsigned)(ro[jj][s] >> 32) == c.epoch && (unsigned)(rm[jj][s] >> 32) == c.epoch && (unsigned)(rl[jj][s] >> 32) == c.epoch;
                                    ov[jj][s] = __builtin_bit_cast(float, (unsigned)ro[jj][s]);
                                    mv[jj][s] = __builtin_bit_cast(float, (unsigned)rm[jj][s]);
                                    lv[jj][s] = __builtin_bit_cast(float, (unsigned)rl[jj][s]);
                                }
                        if (__all(ok)) break;
                        if (ps_giveup(c, spins)) break;
                    }
#pragma unroll
                    for (int jj = 0; jj < 4; jj++) {
                        const int e = tid + 256 * jj;
                        if (e < PS_H) {
                            float M = -INFINITY;
#pragma unroll
                            for (int s = 0; s < PS_NSPL; s++) if (s < nlive) M = fmaxf(M, mv[jj][s]);
                            float num = 0.f, den = 0.f;
#pragma unroll
                            for (int s = 0; s < PS_NSPL; s++)
                                if (s < nlive) { const float w = __expf(mv[jj][s] - M); den += w * lv[jj][s]; num += w * ov[jj][s]; }
                            const float v = num * (1.f / den);
                            ps_stage(smem, e, v, v);
                        }
                    }
                }
                // residual of this tile: x[16 j .. + 16) (published one layer earlier: no wait to speak of)
                float xr[1];
                ps_gather_x<1>(c, a, l, xr, 16 * j, 16);
                __syncthreads();
                if (*c.fail) return;
                const f32x4 r0 = ps_mfma<7>(wa0, 7, smem, 7 * wave, lane);
                ps_put_partial(smem, 0, wave, lane, r0);
                __syncthreads();
                if (tid < 16) pg_store(lg + PG_XM + 16 * j + tid, c.epoch, xr[0] + ps_get(smem, 0, tid));
                if (l + 1 < NL) load_a(l + 1);
            }
            // ---- down-projection units: h quarter -> 16 partial sums each
            {
                // all units of this CU may sit in different quarters: stage per unit
                __syncthreads();                                   // RED / x stage of the phase above are free again
#pragma unroll
                for (int u = 0; u < 3; u++) {
                    if (u < nd) {
                        const int tile = du[u] >> 2, qt = du[u] & 3;
                        const u64* hg = lg + PG_HB + qt * (PS_DKS * 32);
                        if (u == 0 && !ps_wait_sentinel(c, lg + PG_HB + PS_I - 1)) return;
                        float hv[5];
                        ps_sweep<5>(c, hg, PS_DKS * 32, hv);
#pragma unroll
                        for (int jj = 0; jj < 5; jj++) { const int k = tid + 256 * jj; if (k < PS_DKS * 32) ps_stage(smem, k, hv[jj], hv[jj]); }
                        __syncthreads();
                        if (*c.fail) return;
                        const f32x4 r = ps_mfma<10>(wd[u], dkn, smem, dk0, lane);
                        ps_put_partial(smem, u, wave, lane, r);
                        __syncthreads();
                        if (tid < 16) pg_store(lg + PG_DP + qt * PS_H + tile * 16 + tid, c.epoch, ps_get(smem, u, tid));
                    }
                }
                if (l + 1 < NL) load_d(l + 1);
            }
        }
    } else if (b < 244) {
        // ============================================================ role G: pairs 2 g, 2 g + 1 (gate tile, up tile each)
        const int g2 = (b - 92) * 2;
        s16x8 wg[4][7];
        auto load_g = [&](int l) {
            const uint16_t* W = a.layers[l].wgu;
#pragma unroll
            for (int t = 0; t < 4; t++) ps_load_w<7>(wg[t], W, (size_t)(2 * g2 + t) * PS_KSH + 7 * wave, 7, lane);
        };
        load_g(0);
        for (int l = 0; l < NL; l++) {
            u64* lg = a.gran + (size_t)l * PG_LAYER;
            if (!ps_wait_sentinel(c, lg + PG_XM + PS_H - 1)) return;
            float x[4];
            ps_sweep<4>(c, lg + PG_XM, PS_H, x);
            ps_stage_norm(c, smem, x, a.layers[l].ln2);
            __syncthreads();
            if (*c.fail) return;
#pragma unroll
            for (int t = 0; t < 4; t++) ps_put_partial(smem, t, wave, lane, ps_mfma<7>(wg[t], 7, smem, 7 * wave, lane));
            __syncthreads();
            if (tid < 32) {
                const float rs = ps_rstd(smem, a.eps);
                const int p = tid >> 4, i = tid & 15;
                const float gt = ps_get(smem, 2 * p, i) * rs, up = ps_get(smem, 2 * p + 1, i) * rs;
                pg_store(lg + PG_HB + (g2 + p) * 16 + i, c.epoch, (gt / (1.f + __expf(-gt))) * up);
            }
            if (l + 1 < NL) load_g(l + 1);
            __syncthreads();
        }
    } else {
        // ============================================================ role A: kv group gk, key split sp
        const int au = b - 244, gk = au / PS_NSPL, sp = au % PS_NSPL;
        float* qs = reinterpret_cast<float*>(smem + PL_MISC);          // [7][64]
        float* kn = qs + 448;                                           // [64]
        float* vn = kn + 64;                                            // [64]
        float* ps = vn + 64;                                            // [7][128]
        float* ml = ps + 7 * 128;                                       // [7][2]
        float* po = ml + 16;                                            // [16][448]
        const int key_t = tid >> 1, hf = tid & 1;                       // scores: thread = (key, 32-dim half)
        const int d4 = tid & 15, kq = tid >> 4;                         // PV: thread = (4 dims, 8 keys)
        const int rounds = (L + PS_NSPL * PS_KSPL - 1) / (PS_NSPL * PS_KSPL);
        f32x4 kk[8], vv[8];
        auto load_kv = [&](int l, int rd) {
            const int j0 = (rd * PS_NSPL + sp) * PS_KSPL;
            const float* K = a.kc + (size_t)l * a.cache_l + (size_t)gk * a.max_pos * 64;
            const float* V = a.vc + (size_t)l * a.cache_l + (size_t)gk * a.max_pos * 64;
            const int kr = min(j0 + key_t, a.max_pos - 1);
#pragma unroll
            for (int i = 0; i < 8; i++) kk[i] = *reinterpret_cast<const f32x4*>(K + (size_t)kr * 64 + hf * 32 + 4 * i);
#pragma unroll
            for (int k = 0; k < 8; k++) vv[k] = *reinterpret_cast<const f32x4*>(V + (size_t)min(j0 + kq * 8 + k, a.max_pos - 1) * 64 + d4 * 4);
        };
        const bool live = sp * PS_KSPL < L;                            // this split sees at least one key (round 0)
        if (live) load_kv(0, 0);
        for (int l = 0; l < NL; l++) {
            u64* lg = a.gran + (size_t)l * PG_LAYER;
            if (!live) continue;
            if (!ps_wait_sentinel(c, lg + PG_QKV + (PS_NQ + PS_NKV + gk) * 64 + 63)) return;
            {
                // q of the 7 heads of the group, k_new, v_new: 576 granules
                float v[3];
                for (unsigned spins = 0;; spins++) {
                    bool ok = true;
                    u64 raw[3];
#pragma unroll
                    for (int jj = 0; jj < 3; jj++) {
                        const int i = min(tid + 256 * jj, 575);
                        const int src = i < 448 ? gk * 448 + i : (i < 512 ? (PS_NQ + gk) * 64 + (i - 448) : (PS_NQ + PS_NKV + gk) * 64 + (i - 512));
                        raw[jj] = pg_load(lg + PG_QKV + src);
                    }
#pragma unroll
                    for (int jj = 0; jj < 3; jj++) { ok &= (unsigned)(raw[jj] >> 32) == c.epoch; v[jj] = __builtin_bit_cast(float, (unsigned)raw[jj]); }
                    if (__all(ok)) break;
                    if (ps_giveup(c, spins)) break;
                }
#pragma unroll
                for (int jj = 0; jj < 3; jj++) { const int i = tid + 256 * jj; if (i < 576) qs[i] = v[jj]; }      // qs | kn | vn are contiguous
            }
            __syncthreads();
            if (*c.fail) return;
            float run_m[2] = {-INFINITY, -INFINITY}, run_l[2] = {0.f, 0.f};     // wave w owns heads w and w + 4 (lane-uniform copies)
            f32x4 o[7];
#pragma unroll
            for (int h = 0; h < 7; h++) o[h] = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int rd = 0; rd < rounds; rd++) {
                const int j0 = (rd * PS_NSPL + sp) * PS_KSPL;
                if (j0 >= L) break;
                if (rd > 0) load_kv(l, rd);
                // the new key / value of this step are not in the cache as this launch sees it: take them from the hand-off
                if (j0 + key_t == pos) {
#pragma unroll
                    for (int i = 0; i < 8; i++) kk[i] = *reinterpret_cast<const f32x4*>(kn + hf * 32 + 4 * i);
                }
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const int jk = j0 + kq * 8 + k;
                    if (jk == pos) vv[k] = *reinterpret_cast<const f32x4*>(vn + d4 * 4);
                    else if (jk >= L) vv[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int h = 0; h < 7; h++) {
                    float acc = 0.f;
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        const f32x4 qv = *reinterpret_cast<const f32x4*>(qs + h * 64 + hf * 32 + 4 * i);
                        acc += kk[i][0] * qv[0] + kk[i][1] * qv[1] + kk[i][2] * qv[2] + kk[i][3] * qv[3];
                    }
                    acc += dpp_mov_f32<0xB1, 0xf>(0.f, acc);          // the two halves of a key sit in neighbouring lanes
                    if (hf == 0) ps[h * PS_KSPL + key_t] = j0 + key_t < L ? acc * 0.125f : -INFINITY;
                }
                __syncthreads();
                float tscale[2] = {1.f, 1.f};
#pragma unroll
                for (int hh = 0; hh < 2; hh++) {
                    const int h = wave + 4 * hh;
                    if (h < 7) {
                        const float s0 = ps[h * PS_KSPL + lane], s1 = ps[h * PS_KSPL + 64 + lane];
                        const float mt = wave_max(fmaxf(s0, s1));
                        const float mn = fmaxf(run_m[hh], mt);
                        const float p0 = __expf(s0 - mn), p1 = __expf(s1 - mn);
                        ps[h * PS_KSPL + lane] = p0; ps[h * PS_KSPL + 64 + lane] = p1;
                        const float lt = wave_sum(p0 + p1);
                        tscale[hh] = __expf(run_m[hh] - mn);           // 0 on the first round (run_m = -inf)
                        run_l[hh] = run_l[hh] * tscale[hh] + lt;
                        run_m[hh] = mn;
                        if (lane == 0) { ml[h * 2] = tscale[hh]; }
                    }
                }
                __syncthreads();
#pragma unroll
                for (int h = 0; h < 7; h++) {
                    o[h] *= ml[h * 2];
#pragma unroll
                    for (int k4 = 0; k4 < 2; k4++) {
                        const f32x4 pa = *reinterpret_cast<const f32x4*>(ps + h * PS_KSPL + kq * 8 + 4 * k4);
#pragma unroll
                        for (int e = 0; e < 4; e++) o[h] += pa[e] * vv[4 * k4 + e];
                    }
                }
                __syncthreads();                                    // ps / ml are rewritten by the next round
            }
            // prefetch the next layer's tiles, then merge the 16 key groups and publish
            if (l + 1 < NL) load_kv(l + 1, 0);
#pragma unroll
            for (int h = 0; h < 7; h++) *reinterpret_cast<f32x4*>(po + kq * 448 + h * 64 + d4 * 4) = o[h];
#pragma unroll
            for (int hh = 0; hh < 2; hh++) {
                const int h = wave + 4 * hh;
                if (h < 7 && lane == 0) { ml[h * 2] = run_m[hh]; ml[h * 2 + 1] = run_l[hh]; }
            }
            __syncthreads();
            u64* ap = lg + PG_AP + (size_t)sp * PS_APN;
            for (int e = tid; e < 448; e += 256) {
                float s = 0.f;
#pragma unroll
                for (int q = 0; q < 16; q++) s += po[q * 448 + e];
                pg_store(ap + gk * 448 + e, c.epoch, s);
            }
            if (tid < 14) pg_store(ap + 896 + (tid & 1) * 14 + gk * 7 + (tid >> 1), c.epoch, ml[tid]);      // m -> 896 + head, l -> 910 + head
            __syncthreads();
        }
    }

    // ================================================================ head: final norm -> llm_decoder rows of tiles b, b + 256
    {
        const int nt = a.vocab_pad / 16;
        const int t0 = b, t1 = b + PS_G;
        s16x8 wh[2][7];
        if (t0 < nt) ps_load_w<7>(wh[0], a.wdec, (size_t)t0 * PS_KSH + 7 * wave, 7, lane);
        if (t1 < nt) ps_load_w<7>(wh[1], a.wdec, (size_t)t1 * PS_KSH + 7 * wave, 7, lane);
        if (t0 >= nt) return;
        __syncthreads();
        if (!ps_wait_sentinel(c, a.gran + (size_t)(NL - 1) * PG_LAYER + PG_DP + 3 * PS_H + PS_H - 1)) return;
        float x[4];
        ps_gather_x<4>(c, a, NL, x, 0, PS_H);
        ps_stage_norm(c, smem, x, a.final_norm);
        __syncthreads();
        if (*c.fail) return;
        ps_put_partial(smem, 0, wave, lane, ps_mfma<7>(wh[0], 7, smem, 7 * wave, lane));
        if (t1 < nt) ps_put_partial(smem, 1, wave, lane, ps_mfma<7>(wh[1], 7, smem, 7 * wave, lane));
        __syncthreads();
        if (tid < 32) {
            const int p = tid >> 4, i = tid & 15, t = p ? t1 : t0;
            if (t < nt) a.logits[t * 16 + i] = ps_get(smem, p, i) * ps_rstd(smem, a.eps) + a.bdec[t * 16 + i];
        }
    }
}
