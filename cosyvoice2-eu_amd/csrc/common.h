// Shared device/host helpers for the cv2amd HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
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

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
