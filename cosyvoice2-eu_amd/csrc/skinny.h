// Skinny GEMM core for weight-streaming decode (rows <= 32): out[r][n] = sum_k W[n][k] * x[r][k].
//
// Roofline: HBM.  Every weight byte is read exactly once per launch with 16 B/lane non-temporal loads from the
// packed layout (include/cv2_amd.h): one wave-load = one contiguous 1 KiB block = one MFMA A operand.
// The activation operand is split x = hi + lo (two bf16) and multiplied twice, so the products keep ~16
// mantissa bits of x while the MFMA (otherwise idle in this HBM-bound kernel) does the work; accumulation fp32.
//
// Block = NWR row tiles (16 output features each) x NWK K-slices (one wave each); gridDim.y splits K further.
// Prologue (overlapped with the first weight loads in flight): the block's x slice is summed from its partial
// buffers, optionally RMS-normalised, split and written to LDS in MFMA B-operand order.
#pragma once
#include "common.h"

#define SK_ROWS_CAP 32    // rows capacity of partial buffers
#define SK_U 8            // weight fragments in flight per wave

struct SkinnyX {
    const float* base;    // [rows][K]
    const float* parts;   // [np][SK_ROWS_CAP][K] partial sums added to base (may be null)
    int np;
    const float* norm_w;  // RMSNorm weight [K] or null (requires gridDim.y == 1)
    float eps;
    float* x_out;         // optional [rows][K]: receives base + sum(parts) (pre-norm); written by blockIdx.x == 0
};

template <int NB>
__device__ __forceinline__ int sk_xstage_bytes(int nks) { return nks * NB * 2 * 1024; }

// returns this lane's reduced accumulators in LDS tile res[NWR*16][NB*16+1] (row = feature within block, col = row r)
template <int NB, int NWR, int NWK>
__device__ __forceinline__ float* skinny_core(const uint16_t* __restrict__ W, int KS, int rows, int K, const SkinnyX& X,
                                               char* smem) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave % NWR;
    const int wk = wave / NWR;
    const int nthreads = 64 * NWR * NWK;
    const int tile = blockIdx.x * NWR + wr;
    const int ks0 = (int)(((long)KS * blockIdx.y) / gridDim.y);
    const int ks1 = (int)(((long)KS * (blockIdx.y + 1)) / gridDim.y);
    const int nks = ks1 - ks0;
    const int w0 = ks0 + (nks * wk) / NWK;
    const int w1 = ks0 + (nks * (wk + 1)) / NWK;

    // 1. first SK_U weight fragments in flight before anything else
    const s16x8* wp = reinterpret_cast<const s16x8*>(W) + ((size_t)tile * KS + w0) * 64 + lane;
    s16x8 abuf[SK_U];
#pragma unroll
    for (int i = 0; i < SK_U; i++) {
        if (w0 + i < w1) abuf[i] = __builtin_nontemporal_load(wp + (size_t)i * 64);
    }

    // 2. stage x slice [ks0*32, ks1*32) -> LDS (hi/lo bf16, B-operand order)
    float* rstd = reinterpret_cast<float*>(smem + sk_xstage_bytes<NB>(nks));   // [32]
    if (X.norm_w) {
        for (int r = wave; r < rows; r += NWR * NWK) {
            float ss = 0.f;
            for (int k = lane; k < K; k += 64) {
                float v = X.base[(size_t)r * K + k];
                for (int p = 0; p < X.np; p++) v += X.parts[((size_t)p * SK_ROWS_CAP + r) * K + k];
                ss += v * v;
            }
            ss = wave_sum(ss);
            if (lane == 0) rstd[r] = rsqrtf(ss / (float)K + X.eps);
        }
        __syncthreads();
    }
    {
        const int k8n = nks * 4;    // groups of 8 k per row
        for (int it = tid; it < rows * k8n; it += nthreads) {
            const int r = it / k8n;
            const int k8 = it - r * k8n;
            const int k = ks0 * 32 + k8 * 8;
            const float4* src = reinterpret_cast<const float4*>(X.base + (size_t)r * K + k);
            float4 a = src[0], b = src[1];
            float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            for (int p = 0; p < X.np; p++) {
                const float4* ps = reinterpret_cast<const float4*>(X.parts + ((size_t)p * SK_ROWS_CAP + r) * K + k);
                float4 c = ps[0], d = ps[1];
                v[0] += c.x; v[1] += c.y; v[2] += c.z; v[3] += c.w;
                v[4] += d.x; v[5] += d.y; v[6] += d.z; v[7] += d.w;
            }
            if (X.x_out && blockIdx.x == 0) {
                float4* dst = reinterpret_cast<float4*>(X.x_out + (size_t)r * K + k);
                dst[0] = make_float4(v[0], v[1], v[2], v[3]);
                dst[1] = make_float4(v[4], v[5], v[6], v[7]);
            }
            if (X.norm_w) {
                const float rs = rstd[r];
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] = X.norm_w[k + j] * (v[j] * rs);
            }
            s16x8 hi, lo;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                uint16_t h, l;
                split_bf16(v[j], h, l);
                hi[j] = (short)h;
                lo[j] = (short)l;
            }
            const int s = k8 >> 2, hq = k8 & 3;
            const int t = r >> 4;
            const int ln = hq * 16 + (r & 15);
            s16x8* dst = reinterpret_cast<s16x8*>(smem + ((size_t)(s * NB + t) * 2) * 1024) + ln;
            dst[0] = hi;
            dst[64] = lo;
        }
    }
    __syncthreads();

    // 3. stream the weights
    f32x4 acc[NB];
#pragma unroll
    for (int t = 0; t < NB; t++) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int ks = w0; ks < w1; ks += SK_U) {
        s16x8 nbuf[SK_U];
#pragma unroll
        for (int i = 0; i < SK_U; i++) {
            if (ks + SK_U + i < w1) nbuf[i] = __builtin_nontemporal_load(wp + (size_t)(ks - w0 + SK_U + i) * 64);
        }
#pragma unroll
        for (int i = 0; i < SK_U; i++) {
            if (ks + i < w1) {
                const int s = ks + i - ks0;
                const bf16x8 a = __builtin_bit_cast(bf16x8, abuf[i]);
#pragma unroll
                for (int t = 0; t < NB; t++) {
                    const s16x8* xb = reinterpret_cast<const s16x8*>(smem + ((size_t)(s * NB + t) * 2) * 1024) + lane;
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(bf16x8, xb[0]), acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(bf16x8, xb[64]), acc[t], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < SK_U; i++) abuf[i] = nbuf[i];
    }
    __syncthreads();   // everyone is done with the x stage; reuse it for the cross-wave reduction

    // 4. reduce the NWK K-slices, leave the tile in res[feature][row]
    f32x4* red = reinterpret_cast<f32x4*>(smem);                                    // [NWK][NWR][NB][64]
    float* res = reinterpret_cast<float*>(smem + NWK * NWR * NB * 1024);            // [NWR*16][NB*16+1]
#pragma unroll
    for (int t = 0; t < NB; t++) red[((wk * NWR + wr) * NB + t) * 64 + lane] = acc[t];
    __syncthreads();
    if (wk == 0) {
#pragma unroll
        for (int t = 0; t < NB; t++) {
            f32x4 v = red[((0 * NWR + wr) * NB + t) * 64 + lane];
            for (int q = 1; q < NWK; q++) {
                f32x4 u = red[((q * NWR + wr) * NB + t) * 64 + lane];
                v += u;
            }
            const int b = 16 * t + (lane & 15);
            const int n0 = wr * 16 + 4 * (lane >> 4);
#pragma unroll
            for (int r = 0; r < 4; r++) res[(n0 + r) * (NB * 16 + 1) + b] = v[r];
        }
    }
    __syncthreads();
    return res;
}

template <int NB, int NWR, int NWK>
static inline size_t skinny_smem_bytes(int nks_block) {
    size_t xs = (size_t)nks_block * NB * 2 * 1024 + 32 * sizeof(float);
    size_t rr = (size_t)NWK * NWR * NB * 1024 + (size_t)NWR * 16 * (NB * 16 + 1) * sizeof(float);
    return xs > rr ? xs : rr;
}
