// One-row decode chain: O projection -> gate/up -> down projection of a layer in ONE launch (llm.hip: k_chain).
//
// Why: at one row the five weight-streaming kernels of a layer are latency-bound (1.6 us dispatch + ~1 us first dependent
// load + ~1 us of weight stream each).  Weights do not depend on activations: when the three GEMVs share a launch, every block
// requests its weight fragments at kernel entry, so the gate/up (17.4 MB) and down (8.7 MB) streams land while the O projection
// is still working, and what remains behind each dependency is the hand-off plus ~30 MFMAs.
//
// Hand-offs (cdna_hip_programming.md Guideline 16, form R2): the data is the flag.  A value travels as one naturally aligned
// 8-byte granule {tag = epoch, fp32 bits}, written by ONE write-through (sc1) store and polled with sc1 loads; no fence, no
// separate flag.  epoch = a device counter k_sample advances once per decode step, every layer has granule buffers of its own,
// so a granule of an earlier step never matches and nothing is re-initialised between steps (the buffers are zeroed once at
// create, epoch starts at 1).
//
// Forward progress: producers never wait inside the launch and have the LOWEST block indices of their consumers (O < gate/up <
// down); blocks are dispatched in index order (per XCD too), so a polling consumer can never keep its producer off the chip.
// Every poll is bounded (0.5 s) and reports through the slot's CV2_ST_ERR.
#pragma once
#include "skinny.h"
#ifndef R1_T_OPERAND
#define R1_T_OPERAND do { } while (0)
#endif

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;

__device__ __forceinline__ void gran_store(u64* p, unsigned epoch, float v) {
    __hip_atomic_store((gu64*)p, ((u64)epoch << 32) | (u64)__builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#define GRAN_TIMEOUT_TICKS 50000000ull     // s_memrealtime runs at 100 MHz: 0.5 s

// One wave gathers 8 consecutive granules per active lane; returns false when the tags did not all match within the time limit.
__device__ __forceinline__ bool gran_gather8(const u64* g, unsigned epoch, bool active, f32x8& v) {
    const gu64* p = (const gu64*)g;
    const u64 t0 = __builtin_amdgcn_s_memrealtime();
    for (unsigned spin = 0;; spin++) {
        bool ok = true;
        if (active) {
            u64 x[8];
#pragma unroll
            for (int k = 0; k < 8; k++) x[k] = __hip_atomic_load(p + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int k = 0; k < 8; k++) { ok &= (unsigned)(x[k] >> 32) == epoch; v[k] = __builtin_bit_cast(float, (unsigned)x[k]); }
        }
        if (__all(ok)) return true;
        if ((spin & 15) == 15 && __builtin_amdgcn_s_memrealtime() - t0 > GRAN_TIMEOUT_TICKS) return false;
        __builtin_amdgcn_s_sleep(2);
    }
}

// One wave waits for ONE granule (every lane loads the same word: one request); cheap enough to poll while other blocks stream.
__device__ __forceinline__ bool gran_wait1(const u64* g, unsigned epoch) {
    const gu64* p = (const gu64*)g;
    const u64 t0 = __builtin_amdgcn_s_memrealtime();
    for (unsigned spin = 0;; spin++) {
        const u64 x = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__builtin_amdgcn_readfirstlane((unsigned)(x >> 32)) == epoch) return true;
        if ((spin & 15) == 15 && __builtin_amdgcn_s_memrealtime() - t0 > GRAN_TIMEOUT_TICKS) return false;
        __builtin_amdgcn_s_sleep(8);
    }
}

// ---- operand providers of row1_core: before_weights() / issue() run before the weight loads, finish() after (all threads call them)
struct OpGran {              // the vector arrives as granules from another block of this launch
    const u64* gran; unsigned epoch; int* err;
    const u64* sentinel;     // the granule expected LAST (the highest-indexed producer's): wave 0 polls it alone, then everyone sweeps
    const u64* gate;         // != null: hold the weight requests back until this granule (of an EARLIER hand-off) has arrived, so that
                             // this block's stream does not compete with the phases in front of it
    bool delay;              // hold the weight requests back by ~0.5 us (the phase in front of this one requests first)
    bool dbg;
    __device__ __forceinline__ void before_weights(int wave) {
        if (gate) {
            if (wave == 0 && !gran_wait1(gate, epoch) && (threadIdx.x & 63) == 0) *err = 3;
            __syncthreads();
        } else if (delay) {
            __builtin_amdgcn_s_sleep(19);
        }
    }
    __device__ __forceinline__ void issue(int, int, bool) {}
    __device__ __forceinline__ f32x8 finish(int k, int wave, int nitems, bool active, char*) {
        f32x8 v = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (wave == 0 && !gran_wait1(sentinel, epoch) && (threadIdx.x & 63) == 0) *err = 3;
        __syncthreads();
        if (wave * 64 < nitems) {                                  // wave-uniform: waves without an item do not poll
            if (!gran_gather8(gran + k, epoch, active, v) && (threadIdx.x & 63) == 0) *err = 3;
        }
        return v;
    }
};
struct OpAtt {               // the vector is the split-key attention output: combined while loading (skinny.h, half-fold form)
    SkinnyX X; int K; bool dbg;
    SkRawHalf raw; int hgrp, htid;
    __device__ __forceinline__ void before_weights(int) {}
    __device__ __forceinline__ void issue(int tid, int nitems, bool) {
        hgrp = tid >> 7; htid = tid & 127;
        if (htid < nitems) sk_issue_att_half(X, 0, K, htid * 8, hgrp * SK_HALFSPLIT, raw);
    }
    __device__ __forceinline__ f32x8 finish(int, int, int nitems, bool, char* xch_) {
        float hM = -INFINITY, hden = 0.f;
        f32x8 hacc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, v = hacc;
        if (htid < nitems) sk_partial_att_half(raw, hgrp * SK_HALFSPLIT, hM, hden, hacc);
        float* xch = reinterpret_cast<float*>(xch_);
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
            const float w0 = __expf(hM - M), w1 = __expf(M1 - M);
            const float den = hden * w0 + d[1] * w1;
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = (hacc[e] * w0 + d[2 + e] * w1) * (1.f / den);
        }
        return v;
    }
};

// LDS of one block: operand stage (hi / lo planes, 64 B per 32-wide k-step and plane: only column 0 of the MFMA's B operand is
// real at one row, every lane of a 16-lane quarter reads the same 16 bytes), exchange area of OpAtt, reduction slots.
#define R1_STAGE_BYTES(nks) ((nks) * 128)
#define R1_XCH_BYTES (128 * 10 * 4)
template <int NWR, int NWK>
__device__ __host__ constexpr int r1_smem_bytes(int nks) { return R1_STAGE_BYTES(nks) + R1_XCH_BYTES + 16 + NWK * NWR * 4 * 16; }

// out[f] (f < NWR * 16) = sum_k W[tile0 * 16 + f][k] * x[k] over the block's k-steps [ks0, ks1), x = op's vector [* RMSNorm].
// Returns the feature's value in thread f (threads >= NWR * 16: unspecified).  256 threads = NWR x NWK waves.
template <int NWR, int NWK, int MAXKS, bool NORM, class OP>
__device__ __forceinline__ float row1_core(const uint16_t* __restrict__ W, int tile0, int KS, int K, int ks0, int ks1, OP& op,
                                           const float* norm_w, float eps, float* x_out, char* smem) {
    static_assert(NWR * NWK == 4, "row1_core: 256 threads");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave % NWR, wk = wave / NWR;
    const int nks = ks1 - ks0;
    const int w0 = ks0 + (nks * wk) / NWK, w1 = ks0 + (nks * (wk + 1)) / NWK;
    const int nitems = nks * 4;                                   // groups of 8 operand values; item i belongs to thread i
    const bool active = tid < nitems;
    const int k = ks0 * 32 + tid * 8;
    char* stage = smem;
    char* xch = smem + R1_STAGE_BYTES(nks);
    float* sqs = reinterpret_cast<float*>(xch + R1_XCH_BYTES);     // [4] per-wave sums of squares
    f32x4* red = reinterpret_cast<f32x4*>(xch + R1_XCH_BYTES + 16);   // [NWK][NWR][4 quarters]
    op.before_weights(wave);
    op.issue(tid, nitems, active);
    f32x8 g0 = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
    if (NORM && active) g0 = *reinterpret_cast<const f32x8*>(norm_w + k);
    const char* wbase = reinterpret_cast<const char*>(W) + ((size_t)(tile0 + wr) * KS + w0) * 1024;
    const unsigned wlane = lane * 16;
    s16x8 abuf[MAXKS];
    const int nw = w1 - w0;
#pragma unroll
    for (int i = 0; i < MAXKS; i++)
        abuf[i] = __builtin_nontemporal_load(reinterpret_cast<const s16x8*>(wbase + (size_t)(i < nw ? i : (nw > 0 ? nw - 1 : 0)) * 1024 + wlane));
    __builtin_amdgcn_sched_barrier(0);
    f32x8 v = op.finish(k, wave, nitems, active, xch);
    R1_T_OPERAND;
    if (NORM) {
        float sq = 0.f;
        if (active) sq = ((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3])) + ((v[4] * v[4] + v[5] * v[5]) + (v[6] * v[6] + v[7] * v[7]));
        sq = wave_sum(sq);
        if (lane == 0) sqs[wave] = sq;
    }
    if (active) {
        if (x_out) *reinterpret_cast<f32x8*>(x_out + k) = v;
        if (NORM) v = g0 * v;
        bf16x8 hi, lo;
        split8(v, hi, lo);
        bf16x8* dst = reinterpret_cast<bf16x8*>(stage + (size_t)(tid >> 2) * 128) + (tid & 3);
        dst[0] = hi;
        dst[4] = lo;
    }
    __syncthreads();
    float rs = 1.f;
    if (NORM) rs = rsqrtf(((sqs[0] + sqs[1]) + (sqs[2] + sqs[3])) / (float)K + eps);    // waves without items contributed 0
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < MAXKS; i++) {
        if (w0 + i < w1) {
            const int s = w0 + i - ks0;
            const bf16x8 a = __builtin_bit_cast(bf16x8, abuf[i]);
            const bf16x8* xb = reinterpret_cast<const bf16x8*>(stage + (size_t)s * 128) + (lane >> 4);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, xb[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, xb[4], acc, 0, 0, 0);
        }
    }
    // every column of the tile holds the same vector: the lanes of column 0 carry it out
    if ((lane & 15) == 0) red[(wk * NWR + wr) * 4 + (lane >> 4)] = acc;
    __syncthreads();
    float out = 0.f;
    if (tid < NWR * 16) {
        const int wr_ = tid >> 4, q = (tid >> 2) & 3, r = tid & 3;
        const float* rf = reinterpret_cast<const float*>(red);
        out = rf[((0 * NWR + wr_) * 4 + q) * 4 + r];
#pragma unroll
        for (int j = 1; j < NWK; j++) out += rf[((j * NWR + wr_) * 4 + q) * 4 + r];
        out *= rs;
    }
    return out;
}
