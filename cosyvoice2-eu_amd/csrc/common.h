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

// x = h0 + h1 + h2 EXACTLY, each the top 16 bits (sign, exponent, 7 mantissa bits) of what the previous ones left: truncation instead
// of rounding makes the three-plane form lossless for a 24-bit mantissa and costs two ANDs and two subtractions per element
// (round-to-nearest planes cost ~20 VALU operations per element, and this staging is VALU-bound: it runs once per block and chunk)
__device__ __forceinline__ void split3t(float v, uint32_t& h0, uint32_t& h1, uint32_t& h2) {
    h0 = __builtin_bit_cast(uint32_t, v) & 0xFFFF0000u;
    const float r1 = v - __builtin_bit_cast(float, h0);
    h1 = __builtin_bit_cast(uint32_t, r1) & 0xFFFF0000u;
    const float r2 = r1 - __builtin_bit_cast(float, h1);
    h2 = __builtin_bit_cast(uint32_t, r2);               // <= 8 significant bits left: its low half is zero
}
// two planes' high halves -> one dword (element a low, element b high)
__device__ __forceinline__ uint32_t pack_hi(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }
// the same for 8 / 4 consecutive values, packed as MFMA operand fragments (element e in half e & 1 of dword e >> 1)
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
__device__ __forceinline__ void planes8(const f32x4 a, const f32x4 b, bf16x8& p0, bf16x8& p1, bf16x8& p2) {
    uint32_t h0[8], h1[8], h2[8];
#pragma unroll
    for (int e = 0; e < 4; e++) { split3t(a[e], h0[e], h1[e], h2[e]); split3t(b[e], h0[4 + e], h1[4 + e], h2[4 + e]); }
    p0 = __builtin_bit_cast(bf16x8, (u32x4_t){pack_hi(h0[0], h0[1]), pack_hi(h0[2], h0[3]), pack_hi(h0[4], h0[5]), pack_hi(h0[6], h0[7])});
    p1 = __builtin_bit_cast(bf16x8, (u32x4_t){pack_hi(h1[0], h1[1]), pack_hi(h1[2], h1[3]), pack_hi(h1[4], h1[5]), pack_hi(h1[6], h1[7])});
    p2 = __builtin_bit_cast(bf16x8, (u32x4_t){pack_hi(h2[0], h2[1]), pack_hi(h2[2], h2[3]), pack_hi(h2[4], h2[5]), pack_hi(h2[6], h2[7])});
}
__device__ __forceinline__ void planes4(const f32x4 v, s16x4& p0, s16x4& p1, s16x4& p2) {
    uint32_t h0[4], h1[4], h2[4];
#pragma unroll
    for (int e = 0; e < 4; e++) split3t(v[e], h0[e], h1[e], h2[e]);
    p0 = __builtin_bit_cast(s16x4, (u32x2_t){pack_hi(h0[0], h0[1]), pack_hi(h0[2], h0[3])});
    p1 = __builtin_bit_cast(s16x4, (u32x2_t){pack_hi(h1[0], h1[1]), pack_hi(h1[2], h1[3])});
    p2 = __builtin_bit_cast(s16x4, (u32x2_t){pack_hi(h2[0], h2[1]), pack_hi(h2[2], h2[3])});
}
// S^T = K Q^T and O = P V at fp32 accuracy on the bf16 matrix cores: both operands as three exact bf16 planes, the six products of
// weight >= 2^-16 (what is dropped is below an fp32 rounding of the term; same scheme as k_conv6, hift.hip)
__device__ __forceinline__ f32x4 mm6_32(f32x4 acc, const bf16x8 (&a)[3], const bf16x8 (&b)[3]) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mm6_16(f32x4 acc, const s16x4 (&a)[3], const s16x4 (&b)[3]) {
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a[2], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a[0], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a[1], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a[0], b[1], acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a[0], b[0], acc, 0, 0, 0);
}
// max / sum over the four lanes l, l ^ 16, l ^ 32, l ^ 48 (same column of the four 16-lane rows), result in all of them: two
// v_permlane*_swap (VALU) instead of two LDS-routed shuffles
// (the swap exchanges halves of TWO registers: the second operand is an opaque copy, or the compiler hands the instruction one register
// twice and nothing moves)
// (clang lowers __builtin_bit_cast(float, vec[i]) on a vector ELEMENT lvalue as element 0: convert through a by-value helper)
__device__ __forceinline__ float bits2f(unsigned u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ void swap32(float v, float& a, float& b) {
    unsigned u = __builtin_bit_cast(unsigned, v), w = u;
    asm volatile("" : "+v"(w));
    const auto r = __builtin_amdgcn_permlane32_swap(u, w, false, false);
    const unsigned r0 = r[0], r1 = r[1];
    a = bits2f(r0); b = bits2f(r1);
}
__device__ __forceinline__ void swap16(float v, float& a, float& b) {
    unsigned u = __builtin_bit_cast(unsigned, v), w = u;
    asm volatile("" : "+v"(w));
    const auto r = __builtin_amdgcn_permlane16_swap(u, w, false, false);
    const unsigned r0 = r[0], r1 = r[1];
    a = bits2f(r0); b = bits2f(r1);
}
__device__ __forceinline__ float rows4_max(float v) {
    float a, b;
    swap32(v, a, b); v = fmaxf(a, b);
    swap16(v, a, b); return fmaxf(a, b);
}
__device__ __forceinline__ float rows4_sum(float v) {
    float a, b;
    swap32(v, a, b); v = a + b;
    swap16(v, a, b); return a + b;
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

// XCD-aware order of a 2-D grid (cdna_hip_programming.md T1, bijective form): consecutive block ids are dealt round the 8 XCDs; blocks
// with equal id % 8 take one contiguous share of the (x, y) list with y fastest, so the gridDim.y blocks that read the same x tile share
// one L2.  A speed choice only.
__device__ __forceinline__ void xcd_tile_yfast(int& bx, int& by) {
    const int gy = gridDim.y, T = gridDim.x * gy;
    bx = blockIdx.x; by = blockIdx.y;
    if (gridDim.z != 1 || gy == 1 || T < 16) return;
    const int L = bx + by * gridDim.x, x = L & 7, q = T >> 3, r = T & 7;
    const int Lp = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (L >> 3);
    bx = Lp / gy; by = Lp - bx * gy;
}

// The same with the XCDs split nch ways over the y tiles and 8 / nch ways over the x tiles (nch = 1, 2, 4, 8; gridDim.y % nch == 0 and
// gridDim.x % (8 / nch) == 0: the launcher pads x, blocks past x_real leave).  XCD (c, f) serves y tiles [c gy / nch, (c + 1) gy / nch) of the x
// tiles [f gx / nfr, (f + 1) gx / nfr): what is indexed by y (a convolution's weights) is fetched by 8 / nch L2s, what is indexed by x (its
// input rows) by nch of them.  Returns false for a padding block.
__device__ __forceinline__ bool xcd_tile_split(int& bx, int& by, int nch, int x_real) {
    const int gx = gridDim.x, gy = gridDim.y;
    bx = blockIdx.x; by = blockIdx.y;
    if (gridDim.z != 1 || nch <= 1) { xcd_tile_yfast(bx, by); return bx < x_real; }
    const int L = bx + by * gx, x = L & 7, n = L >> 3;
    const int nfr = 8 / nch, hy = gy / nch, qn = gx / nfr;
    const int c = x % nch, f = x / nch;
    const int dx = n / hy;
    bx = f * qn + dx; by = c * hy + (n - dx * hy);
    return bx < x_real;
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
#define SK_TICK(v) unsigned long long v; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory")
#define SK_ADD(i, d) do { st_[i] += (d); } while (0)
#else
#define SK_TICK(v) do { } while (0)
#define SK_ADD(i, d) do { } while (0)
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
