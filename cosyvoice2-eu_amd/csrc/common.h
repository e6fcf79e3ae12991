// Shared device/host helpers for the cv2amd HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include <stdio.h>
#include <string>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// ---- error plumbing: every C-ABI entry returns 0 / negative and leaves a message here ----
extern thread_local std::string g_cv2_err;
int cv2_fail(const char* fmt, ...);

#define CV2_HIP(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return cv2_fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

#define CV2_CHECK(cond, ...)                \
    do {                                    \
        if (!(cond)) return cv2_fail(__VA_ARGS__); \
    } while (0)

#define CV2_LAUNCH_CHECK() CV2_HIP(hipGetLastError())

// ---- bf16 bit helpers (round-to-nearest-even; inputs are finite activations/weights) ----
__device__ __forceinline__ uint16_t f2bf(float f) {
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float bf2f(uint16_t h) {
    return __builtin_bit_cast(float, ((uint32_t)h) << 16);
}
// x = hi + lo with hi, lo bf16: ~16 mantissa bits of x survive two bf16 MFMAs
__device__ __forceinline__ void split_bf16(float x, uint16_t& hi, uint16_t& lo) {
    hi = f2bf(x);
    lo = f2bf(x - bf2f(hi));
}

// Wave-wide reductions through DPP row operations: one VALU instruction per step instead of an LDS-routed ds_bpermute per
// __shfl_xor (~6 x 100+ cycles on the critical path of every norm / softmax / top-k).  row_shr 1/2/4/8 leave each 16-lane row's
// reduction in its lane 15 (Hillis-Steele scan; lanes without a source contribute the identity), row_bcast 15 / 31 carry it
// across the rows into lane 63; the result is read back wave-uniform.
template <int CTRL, int RMASK>
__device__ __forceinline__ float dpp_mov_f32(float old, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, RMASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_mov_f32<0x111, 0xf>(0.f, v);
    v += dpp_mov_f32<0x112, 0xf>(0.f, v);
    v += dpp_mov_f32<0x114, 0xf>(0.f, v);
    v += dpp_mov_f32<0x118, 0xf>(0.f, v);
    v += dpp_mov_f32<0x142, 0xa>(0.f, v);
    v += dpp_mov_f32<0x143, 0xc>(0.f, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_mov_f32<0x111, 0xf>(v, v));
    v = fmaxf(v, dpp_mov_f32<0x112, 0xf>(v, v));
    v = fmaxf(v, dpp_mov_f32<0x114, 0xf>(v, v));
    v = fmaxf(v, dpp_mov_f32<0x118, 0xf>(v, v));
    v = fmaxf(v, dpp_mov_f32<0x142, 0xa>(v, v));
    v = fmaxf(v, dpp_mov_f32<0x143, 0xc>(v, v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Diagnostic build only (-DCV2_STAMPS, tools/dbg_stamps.py): wave 0 of block 0 records s_memtime at the phase boundaries of
// the stamped kernels into g_stamps[slot][8] (one copy per translation unit); no output value depends on them and the product build contains none of this.
#ifdef CV2_STAMPS
__device__ unsigned long long g_stamps[64][8];
__device__ int g_stamp_slot;
#define SK_STAMP_DECL unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define SK_STAMP(i) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); st_[i] = t_; } while (0)
#define SK_ACC(i, expr) do { unsigned long long t0_, t1_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0_) :: "memory"); expr; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1_) :: "memory"); st_[i] += t1_ - t0_; } while (0)
#define SK_STAMP_FLUSH do { if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && g_stamp_slot < 64) for (int i_ = 0; i_ < 8; i_++) g_stamps[g_stamp_slot][i_] = st_[i_]; } while (0)
#else
#define SK_STAMP_DECL
#define SK_STAMP(i) do { } while (0)
#define SK_ACC(i, expr) do { expr; } while (0)
#define SK_STAMP_FLUSH do { } while (0)
#endif
#ifdef CV2_STAMPS
__device__ int g_stamp_ring;
// ring variant for kernels launched from many places (GEMM): block (0,0,0) takes the next slot itself; entries 6 / 7 carry tags
#define SK_STAMP_FLUSH_RING(tag6, tag7) do { if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) { \
    const int sl_ = atomicAdd(&g_stamp_ring, 1) % 56; for (int i_ = 0; i_ < 6; i_++) g_stamps[sl_][i_] = st_[i_]; \
    g_stamps[sl_][6] = (tag6); g_stamps[sl_][7] = (tag7); } } while (0)
#else
#define SK_STAMP_FLUSH_RING(tag6, tag7) do { } while (0)
#endif
