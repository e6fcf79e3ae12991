// Stage 3 on MI355X: HiFT vocoder, mel -> waveform (replaces cosyvoice/hifigan/generator.py:570-582 and below:
// decode :520-552, _stft/_istft :504-518, ResBlock :94-101, SineGen2 :256-339, SourceModuleHnNSF2 :375-389,
// ConvRNNF0Predictor f0_predictor.py:55-58, Snake transformer/activation.py:73-84).
//
// Layout: every activation is fp32, TIME-MAJOR [L][C].  All arithmetic is fp32 or fp32-equivalent (the reference computes this
// stage in fp32 and the waveform is exp()/sin() of the last conv, so plain bf16 products would not hold a tight tolerance): k_conv
// multiplies on the fp32 matrix cores, k_conv6 forms every product from three bf16 planes per operand (see there).
//
// k_conv: Conv1d as an LDS line-buffered implicit GEMM on the fp32 matrix cores (v_mfma_f32_32x32x2_f32, bit-exact
// fp32 FMA chains at the vector-FMA peak rate, leaving the VALU to the activation and address work):
//   * block = 128 output frames x 64 output channels; wave (fw, cw) owns 64 frames x 32 channels = two MFMA tiles that share
//     every WEIGHT fragment (one 256 B L2 read per two MFMAs; sharing the x fragment instead, as a first version did, made the
//     weight reads saturate the CU's 64 B/clk vector-memory path exactly at the MFMA rate);
//   * per 64-input-channel chunk the block loads the frames [t0 - pad, t0 + 128 + (k-1)*dil - pad) ONCE into an LDS
//     line buffer, applying the pre-activation (Snake / leaky-ReLU) and the zero padding while loading, and every tap
//     then reads its shifted window from LDS (row stride 65 floats: conflict-free ds_read_b32 across frames);
//   * weights are pre-packed per (tap, channel pair, 32-channel output tile) in MFMA B-operand order, one coalesced
//     256 B read per MFMA straight from L2 (they are shared by every block);
//   * epilogue: bias, residual, ELU, and the MRF accumulate modes (sum / (sum + v) / 3) so the three ResBlocks of a
//     stage and the source branch never take an extra pass over HBM.
// ConvTranspose1d(stride u) is the same kernel with u*C_out output channels over 3 input taps (polyphase), whose
// [frame][phase][channel] output IS the time-major upsampled signal.
// Roofline: fp32 MFMA (157 TFLOP/s); HBM traffic per conv is one read + one write of the activation.
#include "common.h"
#include <map>
#include "../../include/cv2_amd.h"
#include <math.h>
#include <atomic>
#include <vector>

enum { PRE_NONE = 0, PRE_SNAKE = 1, PRE_LRELU = 2 };
enum { POST_NONE = 0, POST_ELU = 1 };
enum { ACC_STORE = 0, ACC_ADD = 1, ACC_ADD_DIV3 = 2 };

struct ConvArgs {
    const float* x; int L_in, Cin;          // input frames, channels (row stride Cin)
    const float* wp; const float* bias;     // packed weights [taps][CinP/2][CoutP/32][64], bias [CoutP]
    const uint16_t* w3;                     // k_conv6: three bf16 planes of the weights, [taps][CinP/16][CoutP/32][3][64 lanes][8]
    int CinP, CoutP, Cout_store;            // padded sizes (CinP % 64 == 0, CoutP % 64 == 0)
    int taps, dil, pad_left;
    int pre; const float* alpha; float slope;
    int L_out;
    float* out; int ldo; long out_off;      // out[(t) * ldo + out_off + co]
    const float* res; int ldres;
    int post, acc;
    long zs;                                // batched chunks (cv2_hift_inference_batch): lane blockIdx.z works zs floats further into the workspace
    // flat windows (k_conv6 only; the source_downs, generator.py:468-479): ld_in > 0 -> input row t is the Cin consecutive floats starting at
    // x[t * ld_in + flat_off] of a flat signal of flat_n floats (zero outside it): a strided Conv1d(18 -> C, k, stride s, pad p) over
    // [F][18] rows IS that 1-tap convolution with Cin = 18 k, ld_in = 18 s, flat_off = -18 p (consecutive windows overlap)
    int ld_in; long flat_off, flat_n;
    int xcd_ch;                             // XCDs split this many ways over the output-channel tiles (xcd_tile_split; 0 / 1: frame tiles only)
};

#define CV_BT 128
#define CV_CK 64
#define CV_LD 65

__device__ __forceinline__ float pre_apply(float v, int pre, float al, float slope) {
    // v_sin_f32 (hardware range reduction, abs error ~1e-6 for the |x * alpha| < 100 seen here) instead of the ~40-instruction sinf
    if (pre == PRE_SNAKE) { const float s = __sinf(v * al); return v + (1.0f / (al + 1e-9f)) * (s * s); }
    if (pre == PRE_LRELU) return v > 0.f ? v : v * slope;
    return v;
}

// the same with 1 / (alpha + 1e-9) computed by the caller (the same division, once per channel: bit-identical)
__device__ __forceinline__ float pre_apply_inv(float v, int pre, float al, float inv_al, float slope) {
    if (pre == PRE_SNAKE) { const float s = __sinf(v * al); return v + inv_al * (s * s); }
    if (pre == PRE_LRELU) return v > 0.f ? v : v * slope;
    return v;
}

// FT = frame tiles of 32 per wave (2: blocks of 128 frames = CV_BT; 1: of 64 frames, for grids that leave CUs idle; see k_conv6)
template <int FT>
__global__ __launch_bounds__(256) void k_conv(ConvArgs a) {
    constexpr int BT = 64 * FT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    { const size_t zo = (size_t)blockIdx.z * a.zs; a.x += zo; a.out += zo; if (a.res) a.res += zo; }
    float* xs = reinterpret_cast<float*>(smem);                     // [rows][CV_LD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bx_, by_;
    // frame tiles of one XCD share input rows in its L2, channel tiles share weights: the launcher picks the split (conv_launch)
    if (!xcd_tile_split(bx_, by_, a.xcd_ch, (a.L_out + BT - 1) / BT)) return;
    const int t0 = bx_ * BT, co0 = by_ * 64;
    const int span = (a.taps - 1) * a.dil;
    const int rows = BT + span;
    const int ntile = a.CoutP / 32;
    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; i++) { acc0[i] = 0.f; acc1[i] = 0.f; }
    const int li = lane & 31, lk = lane >> 5;
    const int fw = wave >> 1, cw = wave & 1;                        // frame half (64 frames), 32-channel output tile
    for (int c0 = 0; c0 < a.CinP; c0 += CV_CK) {
        __syncthreads();
        // line buffer: frames t0 - pad_left + r, channels c0 .. c0+63 ; 16 lanes x float4 per frame.  Four items per thread are
        // requested before the first is used: one item at a time, every item paid a full memory round trip before its activation
        for (int it0 = tid; it0 < rows * 16; it0 += 256 * 4) {
            f32x4 q[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int it = it0 + 256 * u;
                const int r = it >> 4, c = c0 + (it & 15) * 4, t = t0 - a.pad_left + r;
                q[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (it < rows * 16 && t >= 0 && t < a.L_in) {
                    const float* src = a.x + (size_t)t * a.Cin + c;
                    if (c + 3 < a.Cin) q[u] = *reinterpret_cast<const f32x4*>(src);
                    else for (int e = 0; e < 4; e++) if (c + e < a.Cin) q[u][e] = src[e];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int it = it0 + 256 * u;
                if (it >= rows * 16) break;
                const int r = it >> 4, c4 = (it & 15) * 4, c = c0 + c4, t = t0 - a.pad_left + r;
                f32x4 v = q[u];
                if (a.pre != PRE_NONE && t >= 0 && t < a.L_in)
                    for (int e = 0; e < 4; e++)
                        if (c + e < a.Cin) v[e] = pre_apply(v[e], a.pre, a.pre == PRE_SNAKE ? a.alpha[c + e] : 0.f, a.slope);
                float* d = xs + r * CV_LD + c4;
                d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
            }
        }
        __syncthreads();
        // weights: groups of 8 k-steps, the NEXT group's 16 fragments are loaded (L2) while the current group's 16 MFMAs run
        constexpr int GK = 8, NG = CV_CK / 2 / GK;               // 4 groups per tap
        const size_t wstride = (size_t)ntile * 64;
        const float* wb = a.wp + (((size_t)0 * (a.CinP / 2) + c0 / 2) * ntile + co0 / 32 + cw) * 64 + lane;
        const size_t tapstride = (size_t)(a.CinP / 2) * ntile * 64;
        // two register sets (A, B) alternate without copies: while the 16 MFMAs of one group run, the other set's 8 loads fly
        float bA[GK], bB[GK];
#define CV_LOAD(SET, GI)                                                                                            \
        {                                                                                                           \
            const float* wn_ = wb + ((GI) / NG) * tapstride + (size_t)((GI) % NG) * GK * wstride;                     \
            _Pragma("unroll") for (int u = 0; u < GK; u++) SET[u] = wn_[u * wstride];                                 \
        }
#define CV_MMA(SET, GI)                                                                                             \
        {                                                                                                           \
            const float* xr_ = xs + (fw * (32 * FT) + li + ((GI) / NG) * a.dil) * CV_LD + lk + 2 * ((GI) % NG) * GK;         \
            _Pragma("unroll") for (int u = 0; u < GK; u++) {                                                          \
                const float av0 = xr_[2 * u], av1 = FT == 2 ? xr_[32 * CV_LD + 2 * u] : 0.f;                          \
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av0, SET[u], acc0, 0, 0, 0);                              \
                if (FT == 2) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av1, SET[u], acc1, 0, 0, 0);                 \
            }                                                                                                       \
        }
        const int ngroups = a.taps * NG;                         // even (NG = 4)
        CV_LOAD(bA, 0)
        for (int gi = 0; gi < ngroups; gi += 2) {
            CV_LOAD(bB, gi + 1)
            __builtin_amdgcn_sched_barrier(0);                   // keep the issue order: hipcc otherwise sinks the loads to their use
            CV_MMA(bA, gi)
            __builtin_amdgcn_sched_barrier(0);
            if (gi + 2 < ngroups) CV_LOAD(bA, gi + 2)
            __builtin_amdgcn_sched_barrier(0);
            CV_MMA(bB, gi + 1)
            __builtin_amdgcn_sched_barrier(0);
        }
#undef CV_LOAD
#undef CV_MMA
    }
    // epilogue: lane holds channel co (li) of its output tile and 16 frames of each of its two frame tiles
    // Every residual value and every accumulate-mode output value of the lane (16 frames per tile) is requested BEFORE any is used, from
    // clamped addresses instead of behind a per-frame branch: with `if (t >= L_out) continue` in front of each element hipcc kept the loads
    // where they were -- 32 dependent global round trips per lane (phase stamps: 33-46 000 cycles of epilogue around 8-38 000 of MFMAs).
    {
        const int co = co0 + cw * 32 + li;
        const bool cok = co < a.Cout_store;
        const int coc = cok ? co : 0;
        const float b = a.bias ? a.bias[coc] : 0.f;
        float rv[FT][16], ov[FT][16];
#pragma unroll
        for (int tile = 0; tile < FT; tile++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int t = min(t0 + fw * (32 * FT) + tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk, a.L_out - 1);
                rv[tile][r] = a.res ? a.res[(size_t)t * a.ldres + coc] : 0.f;
                ov[tile][r] = a.acc != ACC_STORE ? a.out[(size_t)t * a.ldo + a.out_off + coc] : 0.f;
            }
#pragma unroll
        for (int tile = 0; tile < FT; tile++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int t = t0 + fw * (32 * FT) + tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                float v = (tile == 0 ? acc0[r] : acc1[r]) + b;
                if (a.res) v += rv[tile][r];
                if (a.post == POST_ELU) v = v > 0.f ? v : expm1f(v);
                if (a.acc == ACC_ADD) v = ov[tile][r] + v;
                else if (a.acc == ACC_ADD_DIV3) v = (ov[tile][r] + v) / 3.0f;
                if (cok && t < a.L_out) a.out[(size_t)t * a.ldo + a.out_off + co] = v;
            }
    }
}

// k_conv6: the same convolution on the bf16 matrix cores at fp32 accuracy.  Both operands are split into three bf16 planes
// (x = x0 + x1 + x2, each the leading bf16 of what the previous ones left: 24 mantissa bits) and the six products of weight >= 2^-16
// (x0 w0, x0 w1, x1 w0, x0 w2, x1 w1, x2 w0) are accumulated in the MFMA's fp32 accumulators; what is dropped (x1 w2, x2 w1, x2 w2)
// is below 2^-23 of the term, i.e. the size of an fp32 rounding.  Six 32x32x16 bf16 MFMAs (32 cycles each) replace the eight
// 32x32x2 fp32 MFMAs (64 cycles each) of a 16-channel step: 192 against 512 matrix-core cycles per tile.  Same block shape,
// staging, epilogue and argument struct as k_conv; the line buffer holds the three planes of the pre-activated input.
// NP = 2 (round 6; the default, CV2_HIFT_PLANES=3 / cv2_hift_debug_precision(0) bring the three planes back): TWO planes per operand -- x = x0 + x1 with x1 the round-to-nearest bf16 of
// what x0 left (|x - x0 - x1| <= 2^-17 |x|), w0 + w1 of the same three-plane weight buffer -- and the three products x1 w0, x0 w1, x0 w0:
// what is dropped (x1 w1, the operands' third planes) is <= ~2^-15.4 of a term against 2^-23 with three planes; half the matrix-core
// work, two thirds of the weight-fragment traffic and of the line buffer.
__device__ __forceinline__ void split2r(float v, uint32_t& h0, uint32_t& h1) {
    h0 = __builtin_bit_cast(uint32_t, v) & 0xFFFF0000u;
    const uint32_t r = __builtin_bit_cast(uint32_t, v - __builtin_bit_cast(float, h0));
    h1 = (r + 0x7FFFu + ((r >> 16) & 1u)) & 0xFFFF0000u;
}
#define C6_G 4                                        // 16-channel steps per weight register set (4 = one tap of the 64-channel chunk)
#define C6_LD 72                                      // bf16 elements per line-buffer row (64 + 8: 144-B stride, conflict-free ds_read_b128)
// (split3t / pack_hi: common.h)
// FT = frame tiles of 32 per wave: 2 -> blocks of 128 frames (CV_BT), 1 -> blocks of 64 frames for convolutions whose 128-frame grid leaves
// half the CUs idle (one utterance: the 256-channel stage is 32 x 4 = 128 blocks)
// TAPS > 0: the tap count at compile time (3 / 7 / 11 cover every layer but the 1-tap flat windows; 0 = run-time a.taps).  With it the loop
// over the taps' weight register sets is straight-line code and the waits hipcc places are counted ones: around the run-time loop it put
// s_waitcnt vmcnt(0) in front of every set's MFMAs -- behind the request for the OTHER set, i.e. no weight fragment was ever in flight
// beside the matrix cores (the same effect and the same remedy as k_gemm_panel's compile-time K).
template <int FT, int TAPS = 0, int NP = 3>
__global__ __launch_bounds__(256) void k_conv6(ConvArgs a) {
    constexpr int BT = 64 * FT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    { const size_t zo = (size_t)blockIdx.z * a.zs; a.x += zo; a.out += zo; if (a.res) a.res += zo; }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bx_, by_;
    // frame tiles of one XCD share input rows in its L2, channel tiles share weights: the launcher picks the split (conv_launch)
    if (!xcd_tile_split(bx_, by_, a.xcd_ch, (a.L_out + BT - 1) / BT)) return;
    const int t0 = bx_ * BT, co0 = by_ * 64;
    const int span = (a.taps - 1) * a.dil;
    const int rows = BT + span;
    uint16_t* xp[3];
    xp[0] = reinterpret_cast<uint16_t*>(smem); xp[1] = xp[0] + (size_t)rows * C6_LD; xp[2] = xp[1] + (size_t)rows * C6_LD;
    const int ntile = a.CoutP / 32;
    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; i++) { acc0[i] = 0.f; acc1[i] = 0.f; }
    const int li = lane & 31, lk = lane >> 5;
    const int fw = wave >> 1, cw = wave & 1;                        // frame half (32 FT frames), 32-channel output tile
    const size_t kbstride = (size_t)ntile * 3 * 512;                // elements between consecutive 16-channel blocks of one tap
    SK_STAMP_DECL;
    SK_STAMP(0);
    for (int c0 = 0; c0 < a.CinP; c0 += CV_CK) {
        __syncthreads();
        // (a thread's items share their column group -- 1024 is a multiple of 16 --: the Snake alphas of its four channels are fetched once per
        // chunk, not once per element of every item)
        f32x4 al4 = {0.f, 0.f, 0.f, 0.f}, ia4 = {0.f, 0.f, 0.f, 0.f};      // alpha and 1 / (alpha + 1e-9) (activation.py:84): one division per channel, not per element
        if (a.pre == PRE_SNAKE) {
            const int ca = c0 + (tid & 15) * 4;
            for (int e = 0; e < 4; e++) if (ca + e < a.Cin) { al4[e] = a.alpha[ca + e]; ia4[e] = 1.0f / (al4[e] + 1e-9f); }
        }
        // A thread's items are rows r_b, r_b + 16, ..: of its column group: one base pointer and constant increments (the address arithmetic per
        // item -- 64-bit products, bounds, the item -> (row, column) split -- was 6 000 of a single-chunk block's 17 000 staging cycles).  Six
        // items are requested before the first is used: two round trips for the longest line buffer (178 rows = 11.1 items per thread).
        const int r_b = tid >> 4, c4 = (tid & 15) * 4, c = c0 + c4;
        const int t_b = t0 - a.pad_left + r_b;
        const bool cfull = c + 3 < a.Cin;
        const float* src_b = a.ld_in ? a.x : a.x + (long)t_b * a.Cin + c;             // (dereferenced only where the row is inside the signal)
        const long rstep = 16l * a.Cin;
        const long flat_b = (long)t_b * a.ld_in + a.flat_off + c, flat_step = 16l * a.ld_in;
        uint16_t* xo = xp[0] + (size_t)r_b * C6_LD + c4;
        const size_t plane = (size_t)rows * C6_LD;
        const int n_it = (rows - r_b + 15) >> 4;                  // items of this thread
        for (int u0 = 0; u0 < n_it; u0 += 6) {
            f32x4 q[6];
            SK_TICK(ta_);
#pragma unroll
            for (int uu = 0; uu < 6; uu++) {
                const int u = u0 + uu, t = t_b + 16 * u;
                q[uu] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (u >= n_it) continue;
                if (a.ld_in) {                                      // flat windows (block-uniform branch): rows overlap, 8-byte alignment at best
                    const long base = flat_b + u * flat_step;
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const long ix = base + e;
                        if (c + e < a.Cin && ix >= 0 && ix < a.flat_n) q[uu][e] = a.x[ix];
                    }
                } else if (t >= 0 && t < a.L_in) {
                    const float* src = src_b + u * rstep;
                    if (cfull) q[uu] = *reinterpret_cast<const f32x4*>(src);
                    else for (int e = 0; e < 4; e++) if (c + e < a.Cin) q[uu][e] = src[e];
                }
            }
            SK_TICK(tb_);
#pragma unroll
            for (int uu = 0; uu < 6; uu++) {
                const int u = u0 + uu, t = t_b + 16 * u;
                if (u >= n_it) break;
                f32x4 v = q[uu];
                if (a.pre != PRE_NONE && t >= 0 && t < a.L_in) {
                    if (cfull) {
#pragma unroll
                        for (int e = 0; e < 4; e++) v[e] = pre_apply_inv(v[e], a.pre, al4[e], ia4[e], a.slope);
                    } else
                        for (int e = 0; e < 4; e++)
                            if (c + e < a.Cin) v[e] = pre_apply_inv(v[e], a.pre, al4[e], ia4[e], a.slope);
                }
                uint32_t h0[4], h1[4], h2[4];
#pragma unroll
                for (int e = 0; e < 4; e++) { if (NP == 3) split3t(v[e], h0[e], h1[e], h2[e]); else split2r(v[e], h0[e], h1[e]); }
                uint16_t* o = xo + (size_t)u * 16 * C6_LD;
                *reinterpret_cast<uint2*>(o) = make_uint2(pack_hi(h0[0], h0[1]), pack_hi(h0[2], h0[3]));
                *reinterpret_cast<uint2*>(o + plane) = make_uint2(pack_hi(h1[0], h1[1]), pack_hi(h1[2], h1[3]));
                if (NP == 3) *reinterpret_cast<uint2*>(o + 2 * plane) = make_uint2(pack_hi(h2[0], h2[1]), pack_hi(h2[2], h2[3]));
            }
            SK_TICK(tc_);
            SK_ADD(4, tb_ - ta_);                                    // (diagnostic builds) issue of the batch's loads
            SK_ADD(5, tc_ - tb_);                                    // wait for them + activation + planes + LDS writes
        }
        __syncthreads();
        SK_STAMP(1);                                                 // (last chunk's) line buffer staged
        // one 16-channel step = weight fragments of the three planes (3 x 16 B per lane, global / L2) + the two frame tiles' activation
        // fragments of the three planes (LDS) + 12 MFMAs; the next step's weights are requested before the current step's MFMAs
        const uint16_t* wb = a.w3 + ((size_t)(c0 / 16) * ntile + co0 / 32 + cw) * 3 * 512 + (size_t)lane * 8;
        const size_t tapstride = (size_t)(a.CinP / 16) * kbstride;
        // register sets of C6_G steps (3 C6_G fragments): while one set's 12 C6_G MFMAs run, the other set's loads are in flight
        // (one step = 384 matrix-core cycles is less than an L2 round trip: with one step per set the loop ran at the weights' latency)
        bf16x8 wA[NP * C6_G], wB[NP * C6_G];
#define C6_LOAD(SET, GI)                                                                                            \
        _Pragma("unroll") for (int g_ = 0; g_ < C6_G; g_++) {                                                         \
            const int si_ = (GI) * C6_G + g_;                                                                         \
            const uint16_t* wn_ = wb + (si_ >> 2) * tapstride + (size_t)(si_ & 3) * kbstride;                         \
            _Pragma("unroll") for (int p = 0; p < NP; p++) SET[g_ * NP + p] = *reinterpret_cast<const bf16x8*>(wn_ + p * 512); \
        }
#define C6_MMA(SET, GI)                                                                                             \
        _Pragma("unroll") for (int g_ = 0; g_ < C6_G; g_++) {                                                         \
            const int si_ = (GI) * C6_G + g_;                                                                         \
            const size_t xo_ = (size_t)(fw * (32 * FT) + li + (si_ >> 2) * a.dil) * C6_LD + (si_ & 3) * 16 + lk * 8;         \
            bf16x8 x0[3], x1[3];                                                                                      \
            _Pragma("unroll") for (int p = 0; p < NP; p++) {                                                          \
                x0[p] = *reinterpret_cast<const bf16x8*>(xp[p] + xo_);                                                \
                if (FT == 2) x1[p] = *reinterpret_cast<const bf16x8*>(xp[p] + xo_ + 32 * C6_LD);                      \
            }                                                                                                       \
            const bf16x8 w0_ = SET[g_ * NP], w1_ = SET[g_ * NP + 1];                                                  \
            if (NP == 3) {                                                                                          \
            const bf16x8 w2_ = SET[g_ * NP + NP - 1];                                                                 \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x0[2], w0_, acc0, 0, 0, 0);                                \
            if (FT == 2) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1[2], w0_, acc1, 0, 0, 0);                                \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x0[1], w1_, acc0, 0, 0, 0);                                \
            if (FT == 2) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1[1], w1_, acc1, 0, 0, 0);                                \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x0[0], w2_, acc0, 0, 0, 0);                                \
            if (FT == 2) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1[0], w2_, acc1, 0, 0, 0);                                \
            }                                                                                                       \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x0[1], w0_, acc0, 0, 0, 0);                                \
            if (FT == 2) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1[1], w0_, acc1, 0, 0, 0);                                \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x0[0], w1_, acc0, 0, 0, 0);                                \
            if (FT == 2) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1[0], w1_, acc1, 0, 0, 0);                                \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x0[0], w0_, acc0, 0, 0, 0);                                \
            if (FT == 2) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1[0], w0_, acc1, 0, 0, 0);                                \
        }
        static_assert(4 % C6_G == 0, "a register set must not straddle taps");
        const int ngroups = (TAPS ? TAPS : a.taps) * 4 / C6_G;
        C6_LOAD(wA, 0)
        constexpr int UNR = TAPS ? 8 : 1;                           // (straight-line code when the tap count is a compile-time constant)
#pragma unroll UNR
        for (int gi = 0; gi < ngroups; gi += 2) {
            if (gi + 1 < ngroups) C6_LOAD(wB, gi + 1)
            __builtin_amdgcn_sched_barrier(0);
            C6_MMA(wA, gi)
            __builtin_amdgcn_sched_barrier(0);
            if (gi + 2 < ngroups) C6_LOAD(wA, gi + 2)
            __builtin_amdgcn_sched_barrier(0);
            if (gi + 1 < ngroups) C6_MMA(wB, gi + 1)
            __builtin_amdgcn_sched_barrier(0);
        }
#undef C6_LOAD
#undef C6_MMA
        SK_STAMP(2);                                                 // (last chunk's) MFMAs issued
    }
    // epilogue: identical to k_conv
    // Every residual value and every accumulate-mode output value of the lane (16 frames per tile) is requested BEFORE any is used, from
    // clamped addresses instead of behind a per-frame branch: with `if (t >= L_out) continue` in front of each element hipcc kept the loads
    // where they were -- 32 dependent global round trips per lane (phase stamps: 33-46 000 cycles of epilogue around 8-38 000 of MFMAs).
    // Frame of (tile, r) = tb + 32 tile + (r & 3) + 8 (r >> 2): row pointers by constant increments of the leading dimensions.
    {
        const int co = co0 + cw * 32 + li;
        const bool cok = co < a.Cout_store;
        const int coc = cok ? co : 0;
        const float b = a.bias ? a.bias[coc] : 0.f;
        const int tb = t0 + fw * (32 * FT) + 4 * lk;
        float* const ob = a.out + a.out_off + coc;
        const float* const rb = a.res ? a.res + coc : nullptr;
        const int last = a.L_out - 1;
        float rv[FT][16], ov[FT][16];
#pragma unroll
        for (int tile = 0; tile < FT; tile++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int t = min(tb + tile * 32 + (r & 3) + 8 * (r >> 2), last);
                rv[tile][r] = rb ? rb[(long)t * a.ldres] : 0.f;
                ov[tile][r] = a.acc != ACC_STORE ? ob[(long)t * a.ldo] : 0.f;
            }
#pragma unroll
        for (int tile = 0; tile < FT; tile++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int t = tb + tile * 32 + (r & 3) + 8 * (r >> 2);
                float v = (tile == 0 ? acc0[r] : acc1[r]) + b;
                if (a.res) v += rv[tile][r];
                if (a.post == POST_ELU) v = v > 0.f ? v : expm1f(v);
                if (a.acc == ACC_ADD) v = ov[tile][r] + v;
                else if (a.acc == ACC_ADD_DIV3) v = (ov[tile][r] + v) / 3.0f;
                if (cok && t <= last) ob[(long)t * a.ldo] = v;
            }
    }
    SK_STAMP(3);
    SK_STAMP_FLUSH_RING(((unsigned long long)a.taps << 48) | ((unsigned long long)a.dil << 32) | (unsigned)a.CinP,
                        ((unsigned long long)(64 * FT) << 48) | ((unsigned long long)a.CoutP << 32) | (unsigned)(gridDim.x * gridDim.y * gridDim.z));
}

// k_respair (round 5): one (dilated convolution, convolution) pair of a ResBlock (generator.py:94-101) in ONE launch:
//     out = x + conv2(snake(conv1(snake(x)) + b1)) + b2          (conv1: k taps, dilation d; conv2: k taps, dilation 1; C -> C -> C)
// for the 64- and 128-channel stages (C = 64 CT).  A block owns 128 - (k - 1) output frames and ALL channels: it stages the input rows of
// the 128 conv1 frames it needs (its output frames + (k - 1) / 2 on either side) as k_conv6 does -- Snake, three bf16 planes, line buffer
// --, runs conv1 on the matrix cores, and instead of writing the intermediate to HBM leaves snake(conv1 + b1) as three planes in LDS
// (the buffer takes the place of the input's line buffer, which is dead by then; rows outside the signal are conv2's zero padding), runs
// conv2 from there and closes with k_conv6's epilogue (residual x from global memory, store or MRF accumulate).  Per output element
// every sum runs over the same terms in the same order as in the two k_conv6 launches (chunk, tap, 16-channel step): the waveform is
// bit-identical.  What it saves per pair: the intermediate's write and read (2 of 5 activation passes), one launch, one staging from
// global memory and one epilogue to it; what it costs: 128 / (128 - (k - 1)) more frame tiles (2-8 %).  The 256-channel stage keeps
// the two launches: its intermediate (3 planes x 138 rows x 264 columns = 218 KB) does not fit the 160 KB of LDS.
struct PairArgs {
    const float* x; int L;                                            // input / residual rows [L][C]
    const uint16_t* w1; const float* b1; const float* al1; int taps, dil;     // conv1: three-plane weights [taps][C/16][C/32][3][64][8], bias, Snake alpha of its input
    const uint16_t* w2; const float* b2; const float* al2;            // conv2 (dilation 1): the same
    float* out; int acc;                                              // out[t][C], ACC_STORE / ACC_ADD / ACC_ADD_DIV3
    long zs;
};
template <int CT, int TAPS = 0, int NP = 3>         // TAPS: as k_conv6's (compile-time tap count: counted waits around the weight sets); NP: planes
__global__ __launch_bounds__(256 * CT) void k_respair(PairArgs a) {
    constexpr int NT = 256 * CT, C = 64 * CT, LDB = C + 8, NTILE = 2 * CT, RP = NT / 16;      // RP = line-buffer rows per staging pass
    extern __shared__ __attribute__((aligned(16))) char smem[];
    { const size_t zo = (size_t)blockIdx.z * a.zs; a.x += zo; a.out += zo; }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lk = lane >> 5;
    const int fw = wave / NTILE, cw = wave % NTILE;                   // frame half (64 frames), 32-channel output tile
    const int p2 = (a.taps - 1) >> 1, span1 = (a.taps - 1) * a.dil, p1 = span1 >> 1;
    const int rows1 = 128 + span1, rowsB = 128 + a.taps - 1, BTo = 128 - (a.taps - 1);
    const int t0 = blockIdx.x * BTo;                                  // first output frame
    const int f0 = t0 - p2;                                           // first conv1 frame = row 0 of the intermediate
    uint16_t* xp[3];
    xp[0] = reinterpret_cast<uint16_t*>(smem); xp[1] = xp[0] + (size_t)rows1 * C6_LD; xp[2] = xp[1] + (size_t)rows1 * C6_LD;
    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; i++) { acc0[i] = 0.f; acc1[i] = 0.f; }
    constexpr size_t kbstride = (size_t)NTILE * 3 * 512;              // elements between consecutive 16-channel blocks of one tap
    constexpr size_t tapstride = (size_t)(C / 16) * kbstride;
    bf16x8 wA[NP * C6_G], wB[NP * C6_G];
    constexpr int RP_UNR = TAPS ? 8 : 1;                             // (straight-line weight-set loop when the tap count is a compile-time constant)
#define RP_LOAD(SET, GI)                                                                                            \
    _Pragma("unroll") for (int g_ = 0; g_ < C6_G; g_++) {                                                             \
        const int si_ = (GI) * C6_G + g_;                                                                             \
        const uint16_t* wn_ = wb + (si_ >> 2) * tapstride + (size_t)(si_ & 3) * kbstride;                             \
        _Pragma("unroll") for (int p = 0; p < NP; p++) SET[g_ * NP + p] = *reinterpret_cast<const bf16x8*>(wn_ + p * 512); \
    }
#define RP_MMA(SET, GI, BUF, LD, DIL, COL0)                                                                         \
    _Pragma("unroll") for (int g_ = 0; g_ < C6_G; g_++) {                                                             \
        const int si_ = (GI) * C6_G + g_;                                                                             \
        const size_t xo_ = (size_t)(fw * 64 + li + (si_ >> 2) * (DIL)) * (LD) + (COL0) + (si_ & 3) * 16 + lk * 8;     \
        bf16x8 x0[3], x1[3];                                                                                          \
        _Pragma("unroll") for (int p = 0; p < NP; p++) {                                                              \
            x0[p] = *reinterpret_cast<const bf16x8*>(BUF[p] + xo_);                                                   \
            x1[p] = *reinterpret_cast<const bf16x8*>(BUF[p] + xo_ + 32 * (LD));                                       \
        }                                                                                                           \
        const bf16x8 w0_ = SET[g_ * NP], w1_ = SET[g_ * NP + 1];                                                      \
        if (NP == 3) {                                                                                              \
        const bf16x8 w2_ = SET[g_ * NP + NP - 1];                                                                     \
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x0[2], w0_, acc0, 0, 0, 0);                                    \
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1[2], w0_, acc1, 0, 0, 0);                                    \
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x0[1], w1_, acc0, 0, 0, 0);                                    \
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1[1], w1_, acc1, 0, 0, 0);                                    \
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x0[0], w2_, acc0, 0, 0, 0);                                    \
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1[0], w2_, acc1, 0, 0, 0);                                    \
        }                                                                                                           \
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x0[1], w0_, acc0, 0, 0, 0);                                    \
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1[1], w0_, acc1, 0, 0, 0);                                    \
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x0[0], w1_, acc0, 0, 0, 0);                                    \
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1[0], w1_, acc1, 0, 0, 0);                                    \
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x0[0], w0_, acc0, 0, 0, 0);                                    \
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1[0], w0_, acc1, 0, 0, 0);                                    \
    }
#define RP_RUN(BUF, LD, DIL, COL0)                                                                                  \
    {                                                                                                               \
        const int ngroups = (TAPS ? TAPS : a.taps) * 4 / C6_G;                                                        \
        RP_LOAD(wA, 0)                                                                                                \
        _Pragma("unroll RP_UNR") for (int gi = 0; gi < ngroups; gi += 2) {                                            \
            if (gi + 1 < ngroups) RP_LOAD(wB, gi + 1)                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            RP_MMA(wA, gi, BUF, LD, DIL, COL0)                                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            if (gi + 2 < ngroups) RP_LOAD(wA, gi + 2)                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            if (gi + 1 < ngroups) RP_MMA(wB, gi + 1, BUF, LD, DIL, COL0)                                              \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
        }                                                                                                           \
    }
    // ---- conv1: per 64-channel chunk stage snake(x) as three planes (k_conv6's staging), then the chunk's MFMAs
    for (int c0 = 0; c0 < C; c0 += 64) {
        __syncthreads();
        f32x4 al4, ia4;
        const int r_b = tid >> 4, c4 = (tid & 15) * 4, c = c0 + c4;
#pragma unroll
        for (int e = 0; e < 4; e++) { al4[e] = a.al1[c + e]; ia4[e] = 1.0f / (al4[e] + 1e-9f); }
        const int t_b = f0 - p1 + r_b;
        const float* src_b = a.x + (long)t_b * C + c;                 // (dereferenced only where the row is inside the signal)
        uint16_t* xo = xp[0] + (size_t)r_b * C6_LD + c4;
        const size_t plane = (size_t)rows1 * C6_LD;
        const int n_it = (rows1 - r_b + RP - 1) / RP;                 // items of this thread: rows r_b, r_b + RP, ..
        for (int u0 = 0; u0 < n_it; u0 += 6) {
            f32x4 q[6];
#pragma unroll
            for (int uu = 0; uu < 6; uu++) {
                const int u = u0 + uu, t = t_b + RP * u;
                q[uu] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (u < n_it && t >= 0 && t < a.L) q[uu] = *reinterpret_cast<const f32x4*>(src_b + (long)u * RP * C);
            }
#pragma unroll
            for (int uu = 0; uu < 6; uu++) {
                const int u = u0 + uu, t = t_b + RP * u;
                if (u >= n_it) break;
                f32x4 v = q[uu];
                if (t >= 0 && t < a.L) {
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = pre_apply_inv(v[e], PRE_SNAKE, al4[e], ia4[e], 0.f);
                }
                uint32_t h0[4], h1[4], h2[4];
#pragma unroll
                for (int e = 0; e < 4; e++) { if (NP == 3) split3t(v[e], h0[e], h1[e], h2[e]); else split2r(v[e], h0[e], h1[e]); }
                uint16_t* o = xo + (size_t)u * RP * C6_LD;
                *reinterpret_cast<uint2*>(o) = make_uint2(pack_hi(h0[0], h0[1]), pack_hi(h0[2], h0[3]));
                *reinterpret_cast<uint2*>(o + plane) = make_uint2(pack_hi(h1[0], h1[1]), pack_hi(h1[2], h1[3]));
                if (NP == 3) *reinterpret_cast<uint2*>(o + 2 * plane) = make_uint2(pack_hi(h2[0], h2[1]), pack_hi(h2[2], h2[3]));
            }
        }
        __syncthreads();
        const uint16_t* wb = a.w1 + ((size_t)(c0 / 16) * NTILE + cw) * 3 * 512 + (size_t)lane * 8;
        RP_RUN(xp, C6_LD, a.dil, 0)
    }
    __syncthreads();                                                  // every wave is done with the input's line buffer
    // ---- the intermediate: snake(conv1 + b1) as three planes, rows = conv1 frames f0 .. f0 + 127, zero outside the signal
    uint16_t* bp[3];
    bp[0] = reinterpret_cast<uint16_t*>(smem); bp[1] = bp[0] + (size_t)rowsB * LDB; bp[2] = bp[1] + (size_t)rowsB * LDB;
    {
        const int co = cw * 32 + li;
        const float b = a.b1[co], al = a.al2[co], ia = 1.0f / (al + 1e-9f);
#pragma unroll
        for (int tile = 0; tile < 2; tile++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int j = fw * 64 + tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                const int f = f0 + j;
                uint32_t h0 = 0, h1 = 0, h2 = 0;
                if (f >= 0 && f < a.L) {
                    const float y = pre_apply_inv((tile == 0 ? acc0[r] : acc1[r]) + b, PRE_SNAKE, al, ia, 0.f);
                    if (NP == 3) split3t(y, h0, h1, h2); else split2r(y, h0, h1);
                }
                const size_t o = (size_t)j * LDB + co;
                bp[0][o] = (uint16_t)(h0 >> 16); bp[1][o] = (uint16_t)(h1 >> 16);
                if (NP == 3) bp[2][o] = (uint16_t)(h2 >> 16);
            }
    }
#pragma unroll
    for (int i = 0; i < 16; i++) { acc0[i] = 0.f; acc1[i] = 0.f; }
    __syncthreads();
    // ---- conv2 from the intermediate (rows 128 .. rowsB - 1 are never written: they only reach the frames j >= BTo that are dropped below)
    for (int c0 = 0; c0 < C; c0 += 64) {
        const uint16_t* wb = a.w2 + ((size_t)(c0 / 16) * NTILE + cw) * 3 * 512 + (size_t)lane * 8;
        RP_RUN(bp, LDB, 1, c0)
    }
#undef RP_LOAD
#undef RP_MMA
#undef RP_RUN
    // ---- epilogue (k_conv6's): + b2 + x, store / accumulate; frames j >= BTo belong to the next block
    {
        const int co = cw * 32 + li;
        const float b = a.b2[co];
        const int jb = fw * 64 + 4 * lk;
        float* const ob = a.out + co;
        const float* const rb = a.x + co;
        const int last = a.L - 1;
        float rv[2][16], ov[2][16];
#pragma unroll
        for (int tile = 0; tile < 2; tile++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int t = min(t0 + jb + tile * 32 + (r & 3) + 8 * (r >> 2), last);
                rv[tile][r] = rb[(long)t * C];
                ov[tile][r] = a.acc != ACC_STORE ? ob[(long)t * C] : 0.f;
            }
#pragma unroll
        for (int tile = 0; tile < 2; tile++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int j = jb + tile * 32 + (r & 3) + 8 * (r >> 2), t = t0 + j;
                float v = (tile == 0 ? acc0[r] : acc1[r]) + b;
                v += rv[tile][r];
                if (a.acc == ACC_ADD) v = ov[tile][r] + v;
                else if (a.acc == ACC_ADD_DIV3) v = (ov[tile][r] + v) / 3.0f;
                if (j < BTo && t <= last) ob[(long)t * C] = v;
            }
    }
}

// source_downs: Conv1d(18 -> C, k, stride, pad) over s_stft [F][18] (generator.py:468-479); tiny, direct
struct SdArgs { const float* x; int F; const float* w; const float* b; int C, k, stride, pad; float* out; int L_out; long zs; };
__global__ __launch_bounds__(256) void k_source_down(SdArgs a) {
    a.x += (size_t)blockIdx.z * a.zs; a.out += (size_t)blockIdx.z * a.zs;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)a.L_out * a.C) return;
    const int t = idx / a.C, co = idx % a.C;
    float acc = 0.f;
    for (int j = 0; j < a.k; j++) {
        const int f = t * a.stride - a.pad + j;
        if (f < 0 || f >= a.F) continue;
        const float* xr = a.x + (size_t)f * 18;
        const float* wr = a.w + (size_t)j * 18 * a.C + co;           // packed [k][18][C]: the threads of a wave (consecutive co) read consecutive words
#pragma unroll
        for (int c = 0; c < 18; c++) acc += wr[(size_t)c * a.C] * xr[c];
    }
    a.out[(size_t)t * a.C + co] = acc + a.b[co];
}

// mel [80][T] channel-major -> [T][80]
__global__ void k_mel_tm(const float* mel, float* out, int T, long zs) {
    mel += (size_t)blockIdx.z * zs; out += (size_t)blockIdx.z * zs;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= T * 80) return;
    const int t = idx / 80, c = idx % 80;
    out[idx] = mel[(size_t)c * T + t];
}

// f0 = |Linear(512 -> 1)| (f0_predictor.py:57-58): one wave per frame
__global__ __launch_bounds__(256) void k_f0_head(const float* x, const float* w, const float* b, float* f0, int T, long zs) {
    x += (size_t)blockIdx.z * zs; f0 += (size_t)blockIdx.z * zs;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (t >= T) return;
    float acc = 0.f;
    for (int c = lane; c < 512; c += 64) acc += x[(size_t)t * 512 + c] * w[c];
    acc = wave_sum(acc);
    if (lane == 0) f0[t] = fabsf(acc + b[0]);
}

// SineGen2._f02sine (generator.py:263-285), first half: per harmonic the frame-rate phase.
//   rad_h[i] = (f0[i] * h / 24000) % 1     (the 1/480 linear down-interpolation of the nearest-upsampled signal returns
//   exactly the frame value: both taps, samples 480 i + 239 and + 240, lie inside frame i, so rand_ini at sample 0 never matters)
//   phase = cumsum(rad) (float64 accumulate like torch's CPU cumsum) -> fp32 -> * 2 pi -> * 480
// One block per lane: the per-element work (multiply, divide, fmod; the final two multiplies) by all threads over tiles of PH_TILE frames,
// the float64 running sums by nine threads walking the tile's values in LDS -- the same operations in the same order per harmonic as the
// nine-thread loop this replaces (106 us at 500 frames: every iteration paid a global load, a division and an fmod before its one addition).
#define PH_TILE 512
__global__ __launch_bounds__(256) void k_phase(const float* f0, float* phase, int T, long zs) {
    __shared__ float rad[PH_TILE * 9];
    __shared__ float cs[PH_TILE * 9];
    f0 += (size_t)blockIdx.z * zs; phase += (size_t)blockIdx.z * zs;
    const int tid = threadIdx.x;
    double acc = 0.0;                       // threads 0 .. 8: harmonic tid + 1
    for (int t0 = 0; t0 < T; t0 += PH_TILE) {
        const int n = min(PH_TILE, T - t0);
        for (int e = tid; e < n * 9; e += 256) {
            const int i = e / 9, h = e - 9 * i;
            const float fn = __fmul_rn(f0[t0 + i], (float)(h + 1));
            const float q = __fdiv_rn(fn, 24000.0f);
            rad[e] = fmodf(q, 1.0f);
        }
        __syncthreads();
        if (tid < 9) {
            for (int i = 0; i < n; i++) {
                acc += (double)rad[i * 9 + tid];
                cs[i * 9 + tid] = (float)acc;
            }
        }
        __syncthreads();
        for (int e = tid; e < n * 9; e += 256) phase[(size_t)t0 * 9 + e] = __fmul_rn(__fmul_rn(cs[e], 6.283185307179586f), 480.0f);
        __syncthreads();
    }
}

__device__ __forceinline__ uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
__device__ __forceinline__ void philox4(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t o[4]) {
#pragma unroll
    for (int i = 0; i < 10; i++) {
        const uint32_t n0 = mulhi32(0xCD9E8D57u, c2) ^ c1 ^ k0, n1 = 0xCD9E8D57u * c2;
        const uint32_t n2 = mulhi32(0xD2511F53u, c0) ^ c3 ^ k1, n3 = 0xD2511F53u * c0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

// second half + SourceModuleHnNSF2.forward (generator.py:375-389): per sample
//   phase up-interpolation x480 (linear, align_corners=False, the exact fp32 expression order of torch's CPU kernel:
//   src = fma(1/480, n + 0.5, -0.5), w1 = src - floor, out = fma(1 - w1, p0, w1 * p1)), sin, * 0.1, uv gate,
//   + noise_amp * N(0,1), tanh(Linear 9 -> 1).  noise == null: N(0,1) from Philox (seed, sample, harmonic).
struct SrcArgs {
    const float* f0; const float* phase; int T;
    const float* noise;                  // [480 T][9] or null
    uint32_t seed_lo, seed_hi;
    const uint32_t* seed_dev;            // != null: the seed is read from here (graph replays: kernel arguments are frozen)
    const float* lw; const float* lb;    // m_source.l_linear
    const float* cache; int n_cache;     // cache_source overwrite (generator.py:579-580)
    float* s;                            // [480 T]
    long zs;
};
__global__ __launch_bounds__(256) void k_source(SrcArgs a) {
    {
        const size_t zo = (size_t)blockIdx.z * a.zs;
        a.f0 += zo; a.phase += zo; a.s += zo;
        if (a.cache) a.cache += zo;
        if (a.seed_dev) a.seed_dev = reinterpret_cast<const uint32_t*>(reinterpret_cast<const float*>(a.seed_dev) + zo);
    }
    const long n = (long)blockIdx.x * 256 + threadIdx.x;
    const long Ls = (long)a.T * 480;
    if (n >= Ls) return;
    if (n < a.n_cache) { a.s[n] = a.cache[n]; return; }
    float src = __fmaf_rn(1.0f / 480.0f, __fadd_rn((float)n, 0.5f), -0.5f);
    src = fmaxf(src, 0.f);
    const int i0 = (int)src, i1 = min(i0 + 1, a.T - 1);
    const float w1 = __fsub_rn(src, (float)i0), w0 = __fsub_rn(1.0f, w1);
    const float f0 = a.f0[n / 480];
    const float uv = f0 > 10.f ? 1.f : 0.f;
    const float namp = __fadd_rn(__fmul_rn(uv, 0.003f), __fdiv_rn(__fmul_rn(__fsub_rn(1.f, uv), 0.1f), 3.0f));
    float acc = 0.f;
    float z[9];
    if (a.noise) {
#pragma unroll
        for (int h = 0; h < 9; h++) z[h] = a.noise[n * 9 + h];
    } else {
        uint32_t r[4];
#pragma unroll
        for (int q = 0; q < 3; q++) {          // 3 x (2 Box-Muller pairs) -> 12 normals, 9 used
            philox4((uint32_t)n, (uint32_t)(n >> 32), (uint32_t)q, 0x48694654u, a.seed_dev ? a.seed_dev[0] : a.seed_lo, a.seed_dev ? a.seed_dev[1] : a.seed_hi, r);
            const float u0 = ((r[0] >> 8) + 0.5f) * (1.0f / 16777216.0f), u1 = (r[1] >> 8) * (1.0f / 16777216.0f);
            const float u2 = ((r[2] >> 8) + 0.5f) * (1.0f / 16777216.0f), u3 = (r[3] >> 8) * (1.0f / 16777216.0f);
            const float ra = sqrtf(-2.f * logf(u0)), rb = sqrtf(-2.f * logf(u2));
            const float n0 = ra * cosf(6.283185307f * u1), n1 = ra * sinf(6.283185307f * u1), n2 = rb * cosf(6.283185307f * u3);
            z[3 * q] = n0; z[3 * q + 1] = n1; z[3 * q + 2] = n2;
        }
    }
#pragma unroll
    for (int h = 0; h < 9; h++) {
        const float p0 = a.phase[i0 * 9 + h], p1 = a.phase[i1 * 9 + h];
        const float ph = __fmaf_rn(w0, p0, __fmul_rn(w1, p1));
        const float sw = __fadd_rn(__fmul_rn(__fmul_rn(sinf(ph), 0.1f), uv), __fmul_rn(namp, z[h]));
        acc += a.lw[h] * sw;
    }
    a.s[n] = tanhf(acc + a.lb[0]);
}

// STFT n_fft 16 / hop 4 / periodic hann / center reflect (generator.py:504-510): s [L] -> [L/4 + 1][18] (9 real, 9 imag)
__global__ __launch_bounds__(256) void k_stft(const float* s, int L, float* out, int F, long zs) {
    s += (size_t)blockIdx.z * zs; out += (size_t)blockIdx.z * zs;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)F * 9) return;
    const int f = idx / 9, k = idx % 9;
    float re = 0.f, im = 0.f;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        int p = f * 4 - 8 + j;
        p = p < 0 ? -p : (p >= L ? 2 * (L - 1) - p : p);
        const float w = 0.5f - 0.5f * cospif((float)j / 8.0f);
        const float v = s[p] * w;
        const int ph = (k * j) & 15;
        re += v * cospif((float)ph / 8.0f);
        im -= v * sinpif((float)ph / 8.0f);
    }
    out[(size_t)f * 18 + k] = re;
    out[(size_t)f * 18 + 9 + k] = im;
}

// conv_post output [F][18] -> magnitude / phase (generator.py:546-548) -> inverse rFFT(16) * window: frames [F][16]
__global__ __launch_bounds__(256) void k_istft_frames(const float* x, int F, float* fr, long zs) {
    x += (size_t)blockIdx.z * zs; fr += (size_t)blockIdx.z * zs;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)F * 16) return;
    const int f = idx / 16, n = idx % 16;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const float mag = fminf(expf(x[(size_t)f * 18 + k]), 100.f);
        const float ph = sinf(x[(size_t)f * 18 + 9 + k]);
        const float re = mag * cosf(ph), im = mag * sinf(ph);
        const int a = (k * n) & 15;
        const float c = cospif((float)a / 8.0f), sn = sinpif((float)a / 8.0f);
        if (k == 0 || k == 8) acc += re * c;                      // imaginary parts of DC / Nyquist are ignored by irfft
        else acc += 2.f * (re * c - im * sn);
    }
    const float w = 0.5f - 0.5f * cospif((float)n / 8.0f);
    fr[idx] = acc * (1.0f / 16.0f) * w;
}
// overlap-add / window envelope, trim n_fft/2, clamp (generator.py:517-518, :551)
__global__ __launch_bounds__(256) void k_istft_ola(const float* fr, int F, float* wav, int L, float limit, long zs) {
    fr += (size_t)blockIdx.z * zs; wav += (size_t)blockIdx.z * zs;
    const long n = (long)blockIdx.x * 256 + threadIdx.x;
    if (n >= L) return;
    const long p = n + 8;                                          // position in the un-trimmed signal
    float acc = 0.f, env = 0.f;
    for (int f = (int)(p / 4); f >= 0 && f * 4 + 16 > p; f--) {
        if (f >= F) continue;
        const int j = (int)(p - f * 4);
        const float w = 0.5f - 0.5f * cospif((float)j / 8.0f);
        acc += fr[(size_t)f * 16 + j];
        env += w * w;
    }
    const float v = acc / env;
    wav[n] = fminf(fmaxf(v, -limit), limit);
}
// ReflectionPad1d((1, 0)) after the last upsample (generator.py:529-530): row 0 := row 2 of the shifted signal
__global__ void k_reflect_row0(float* x, int C, long zs) {
    x += (size_t)blockIdx.z * zs;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < C) x[c] = x[2 * C + c];
}

// =========================================================================== host
// lanes of the launch being issued (cv2_hift_inference_batch: gridDim.z chunks, each zs floats further into the workspace); set by hift_run
static thread_local int g_hz_n = 1;
static thread_local long g_hz_zs = 0;
struct cv2_hift {
    cv2_hift_dims d;
    cv2_hift_weights w;
    // workspace
    float *melT, *f0a, *f0b, *f0, *phase, *s, *sstft, *xpre, *x, *xt, *ra, *sum, *sd, *post, *frames;
    // short calls (streaming chunks) are host-launch-bound: ~330 launches for a few ms of GPU work.  They replay a hipGraph captured per
    // (frames, cache length) over engine-owned staging buffers; the caller's tensors are copied in / out around the replay.
    float *g_mel, *g_cs, *g_wav, *g_src; uint32_t* g_seed;
    std::map<long, hipGraphExec_t> graphs;
    hipStream_t cap_stream = nullptr;   // graph capture stream of a one-lane engine (see cv2_hift_create)
    long lane_floats = 0;      // workspace floats of one lane (d.lanes lanes behind one another: lane z's buffers = lane 0's + z * lane_floats)
};
// The stream a graph is captured on: the engine's own, or one that lives only for the capture (cv2_hift_create says which and why).
struct CapStream {
    hipStream_t s = nullptr; bool own = false;
    explicit CapStream(hipStream_t have) : s(have) { if (!s) { own = hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess; if (!own) s = nullptr; } }
    ~CapStream() { if (own && s) (void)hipStreamDestroy(s); }
};
#ifndef HG_MAX_T
#define HG_MAX_T 160          // longest call that goes through a graph (frames)
#endif
#define HG_MAX_GRAPHS 24

struct HCarver {
    char* base; size_t off = 0;
    float* take(size_t n) { size_t o = off; off += (n * 4 + 255) & ~(size_t)255; return base ? reinterpret_cast<float*>(base + o) : nullptr; }
};
static size_t hift_carve(const cv2_hift_dims& d, cv2_hift* h, char* base) {
    HCarver c{base};
    const size_t T = d.max_frames, L3 = 120 * T + 1;
    cv2_hift tmp_{};
    cv2_hift& f = h ? *h : tmp_;
    f.melT = c.take(T * 80); f.f0a = c.take(T * 512); f.f0b = c.take(T * 512); f.f0 = c.take(T); f.phase = c.take(T * 9);
    f.s = c.take(480 * T); f.sstft = c.take(L3 * 18); f.xpre = c.take(T * 512);
    const size_t big = L3 * 64 + 64;                                 // every stage is C * L = 7680 T (+ the reflect row)
    f.x = c.take(big); f.xt = c.take(big); f.ra = c.take(big); f.sum = c.take(big); f.sd = c.take(big);
    f.post = c.take(L3 * 18); f.frames = c.take(L3 * 16);
    const size_t GT = T < HG_MAX_T ? T : HG_MAX_T;
    f.g_mel = c.take(GT * 80); f.g_cs = c.take(480 * GT); f.g_wav = c.take(480 * GT); f.g_src = c.take(480 * GT);
    f.g_seed = reinterpret_cast<uint32_t*>(c.take(64));
    return c.off;
}
extern "C" size_t cv2_hift_workspace_bytes(const cv2_hift_dims* d) { return hift_carve(*d, nullptr, nullptr) * (size_t)(d->lanes > 1 ? d->lanes : 1); }

extern "C" int cv2_hift_create(const cv2_hift_dims* d, const cv2_hift_weights* w, void* ws, size_t ws_bytes, cv2_hift** out) {
    CV2_CHECK(d && w && ws && out, "cv2_hift_create: null argument");
    CV2_CHECK(d->max_frames >= 4, "cv2_hift_create: max_frames too small");
    CV2_CHECK(ws_bytes >= cv2_hift_workspace_bytes(d), "cv2_hift_create: workspace too small");
    cv2_hift* h = new cv2_hift();
    h->d = *d; h->w = *w;
    h->lane_floats = (long)(hift_carve(*d, h, (char*)ws) / 4);
    // Capture streams and the process's hardware queues.  How HIP streams share the few hardware queues depends on how many streams the
    // process holds, and it decides whether work on different streams really runs side by side.  Measured (bench.py, one model; the LLM
    // engine holds one capture stream): with that stream alone, or one more, the HiFT pool's four streams serialise (HiFT of 32
    // utterances 144-149 ms instead of 124); with four or five more (a persistent capture stream per vocoder engine) the scheduler's
    // decode / flow / vocoder streams share queues (8 streaming calls: chunk gap 51-53 ms, 74-104 audio-s/s); with two or three more both
    // are at their best (124 ms; gap 49.5-50 ms, 115-117 audio-s/s).  So the first two one-lane engines of the process keep their capture
    // stream for life and every other engine borrows one per capture (CV2_HIFT_CAP_PERSIST = number kept, diagnostics).
    static std::atomic<int> n_kept{0};
    static const int keep = getenv("CV2_HIFT_CAP_PERSIST") ? atoi(getenv("CV2_HIFT_CAP_PERSIST")) : 2;
    if (d->lanes <= 1 && n_kept.fetch_add(1) < keep && hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking) != hipSuccess) h->cap_stream = nullptr;
    static std::atomic<bool> once{false};    // idempotent attribute call: a second thread repeats it rather than launch before it is in place
    if (!once) {
        CV2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        CV2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
#define CV2_BIG_LDS(K) CV2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(K), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
        CV2_BIG_LDS((k_conv6<2, 0>)); CV2_BIG_LDS((k_conv6<2, 3>)); CV2_BIG_LDS((k_conv6<2, 7>)); CV2_BIG_LDS((k_conv6<2, 11>));
        CV2_BIG_LDS((k_conv6<1, 0>)); CV2_BIG_LDS((k_conv6<1, 3>)); CV2_BIG_LDS((k_conv6<1, 7>)); CV2_BIG_LDS((k_conv6<1, 11>));
        CV2_BIG_LDS((k_respair<1, 3>)); CV2_BIG_LDS((k_respair<1, 7>)); CV2_BIG_LDS((k_respair<1, 11>)); CV2_BIG_LDS((k_respair<1, 0>));
        CV2_BIG_LDS((k_respair<2, 3>)); CV2_BIG_LDS((k_respair<2, 7>)); CV2_BIG_LDS((k_respair<2, 11>)); CV2_BIG_LDS((k_respair<2, 0>));
        CV2_BIG_LDS((k_conv6<2, 0, 2>)); CV2_BIG_LDS((k_conv6<2, 3, 2>)); CV2_BIG_LDS((k_conv6<2, 7, 2>)); CV2_BIG_LDS((k_conv6<2, 11, 2>));
        CV2_BIG_LDS((k_conv6<1, 0, 2>)); CV2_BIG_LDS((k_conv6<1, 3, 2>)); CV2_BIG_LDS((k_conv6<1, 7, 2>)); CV2_BIG_LDS((k_conv6<1, 11, 2>));
        CV2_BIG_LDS((k_respair<1, 3, 2>)); CV2_BIG_LDS((k_respair<1, 7, 2>)); CV2_BIG_LDS((k_respair<1, 11, 2>)); CV2_BIG_LDS((k_respair<1, 0, 2>));
        CV2_BIG_LDS((k_respair<2, 3, 2>)); CV2_BIG_LDS((k_respair<2, 7, 2>)); CV2_BIG_LDS((k_respair<2, 11, 2>)); CV2_BIG_LDS((k_respair<2, 0, 2>));
#undef CV2_BIG_LDS
        once = true;
    }
    *out = h;
    return 0;
}
extern "C" int cv2_hift_destroy(cv2_hift* h) {
    if (h && h->cap_stream) (void)hipStreamDestroy(h->cap_stream);
    if (!h) return 0;
    for (auto& g : h->graphs) (void)hipGraphExecDestroy(g.second);
    delete h;
    return 0;
}
#ifdef CV2_STAMPS
extern "C" int cv2_debug_stamps_hift(unsigned long long* out_host) {       // this translation unit's copy of the stamp ring
    CV2_HIP(hipDeviceSynchronize());
    CV2_HIP(hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 64 * 8));
    return 0;
}
#endif
// test hook: device pointers of intermediate buffers (0 melT, 1 f0, 2 s_stft, 3 conv_pre out, 4 x (last stage out), 5 conv_post out)
extern "C" const float* cv2_hift_debug_buffer(cv2_hift* h, int32_t which) {
    switch (which) { case 0: return h->melT; case 1: return h->f0; case 2: return h->sstft; case 3: return h->xpre; case 4: return h->sum; case 5: return h->post; default: return nullptr; }
}

// Test hook (tests/test_hift_gpu.py): -1 = the defaults (environment), 0 / 1 = without / with the fused ResBlock pairs (k_respair) and the
// per-layer XCD split of the convolution grids.  Calls of <= HG_MAX_T frames replay graphs captured under the mode of their first call.
static std::atomic<int> g_pair_mode{-1}, g_xcd_mode{-1};
extern "C" int cv2_hift_debug_modes(int32_t pair, int32_t xcd_split) {
    g_pair_mode = pair < 0 ? -1 : (pair != 0);
    g_xcd_mode = xcd_split < 0 ? -1 : (xcd_split != 0);
    return 0;
}

// How the convolutions that carry three-plane weights multiply: 2 = two bf16 planes per operand, three products (the default since round 6:
// against the three-plane result the waveform moves by <= 7e-6 abs -- 500 / 300 / 250 / 90 frames, profiles/r6_hift_planes.txt -- and
// against the REFERENCE's waveforms both forms sit at the same 4e-5 .. 1.4e-4, which is the f0 track's share; bar 5e-4), 0 = three planes,
// six products (CV2_HIFT_PLANES=3; fp32-equivalent), 1 = the fp32 matrix-core kernel (k_conv) for every convolution (CV2_HIFT_FP32=1;
// A/B switch).  cv2_hift_debug_precision overrides the environment (tests); graphs of short calls keep the mode of
// their first call.  The f0 predictor is not concerned: its convolutions carry no plane weights and stay on exact fp32 FMA chains.
static std::atomic<int> g_prec_mode{-1};
extern "C" int cv2_hift_debug_precision(int32_t mode) {
    CV2_CHECK(mode >= -1 && mode <= 2, "cv2_hift_debug_precision: mode %d (want -1 .. 2)", mode);
    g_prec_mode = mode;
    return 0;
}
static int hift_precision() {
    const int d = g_prec_mode.load();
    if (d >= 0) return d;
    static const int env = (getenv("CV2_HIFT_FP32") && getenv("CV2_HIFT_FP32")[0] == '1') ? 1
                         : (getenv("CV2_HIFT_PLANES") && getenv("CV2_HIFT_PLANES")[0] == '3') ? 0 : 2;
    return env;
}
static bool hift_fp32_only() { return hift_precision() == 1; }

static int conv_launch(const cv2_conv& cw, const float* x, int L_in, float* out, int ldo, long out_off, int L_out, int pre,
                       const float* alpha, float slope, const float* res, int ldres, int post, int acc, hipStream_t s,
                       int ld_in = 0, long flat_off = 0, long flat_n = 0) {
    ConvArgs a{};
    a.ld_in = ld_in; a.flat_off = flat_off; a.flat_n = flat_n;
    const bool fp32_only = hift_fp32_only();
    // (k_conv ignores ld_in / flat_off: with a flat window it would read x[t * Cin + c] past the source STFT)
    CV2_CHECK(ld_in == 0 || (cw.w3 && cw.taps == 1 && !fp32_only), "hift conv: flat windows need the three-plane kernel (k_conv6) and one tap");
    a.x = x; a.L_in = L_in; a.Cin = cw.cin; a.wp = cw.w; a.bias = cw.b; a.CinP = cw.cin_pad; a.CoutP = cw.cout_pad;
    a.Cout_store = cw.cout; a.taps = cw.taps; a.dil = cw.dil; a.pad_left = cw.pad_left; a.pre = pre; a.alpha = alpha; a.slope = slope;
    a.L_out = L_out; a.out = out; a.ldo = ldo; a.out_off = out_off; a.res = res; a.ldres = ldres; a.post = post; a.acc = acc;
    a.zs = g_hz_zs;
    CV2_CHECK(cw.cin_pad % 64 == 0 && cw.cout_pad % 64 == 0 && cw.w, "hift conv: bad packed weight (cin_pad %d cout_pad %d)", cw.cin_pad, cw.cout_pad);
    // How the 8 XCDs (one L2 each) share a launch: split nch ways over the output-channel tiles and 8 / nch ways over the frame tiles, the
    // weights then cross the fabric 8 / nch times and the input rows nch times (xcd_tile_split).  Round 3 always took nch = 1 (every L2
    // pulls all the weights: 8 x 19 MB for the first upsampling layer, whose input is 1 MB).  CV2_HIFT_XCD_SPLIT=0: that form (A/B)
    static const bool xcd_split_env = !(getenv("CV2_HIFT_XCD_SPLIT") && getenv("CV2_HIFT_XCD_SPLIT")[0] == '0');
    const int xcd_dbg = g_xcd_mode.load();
    const bool xcd_split = xcd_dbg < 0 ? xcd_split_env : xcd_dbg != 0;
    auto grid_for = [&](int BT, bool planes) {
        const int gx = (L_out + BT - 1) / BT, gy = cw.cout_pad / 64;
        int nch = 1;
        if (xcd_split && g_hz_n == 1 && gx * gy >= 16) {
            const double W = (double)cw.taps * cw.cin_pad * cw.cout_pad * (planes ? (hift_precision() == 2 ? 4 : 6) : 4), A = (double)L_in * cw.cin * 4;
            double best = 8 * W + A;
            for (int n = 2; n <= 8; n *= 2)
                if (gy % n == 0 && (8 / n) * W + n * A < best) { best = (8 / n) * W + n * A; nch = n; }
        }
        a.xcd_ch = nch;
        const int nfr = nch > 1 ? 8 / nch : 1;                   // (the unsplit order takes any grid: no padding blocks for short chunks)
        return dim3((gx + nfr - 1) / nfr * nfr, gy, g_hz_n);
    };
    if (cw.w3 && !fp32_only) {
        a.w3 = cw.w3;
        // the kernels with the tap count at compile time (counted waits around the weight register sets); CV2_HIFT_TAPS_CT=0: the run-time loop (A/B)
        static const bool taps_env = !(getenv("CV2_HIFT_TAPS_CT") && getenv("CV2_HIFT_TAPS_CT")[0] == '0');
        const int taps_ct = taps_env ? cw.taps : 0;
        // fewer 128-frame blocks than CUs (one utterance's 256- and 512-channel stages): 64-frame blocks; CV2_HIFT_BT64=0: A/B, diagnostics
        static const bool bt64 = !(getenv("CV2_HIFT_BT64") && getenv("CV2_HIFT_BT64")[0] == '0');
        const long blocks128 = (long)((L_out + CV_BT - 1) / CV_BT) * (cw.cout_pad / 64) * g_hz_n;
        // (measured, 500 frames: below 200 blocks 4.61 ms, below 400 -- the 128-channel stage too -- 4.55, below 600 4.64; 5.34 without)
        static const long bt64_max = getenv("CV2_HIFT_BT64_MAX") ? atol(getenv("CV2_HIFT_BT64_MAX")) : 200;     // (260, one 128-frame block per CU: a lone 900-frame call 5.07 -> 4.76 ms, but 32 utterances on the pool's four streams 81.0 -> 83.5 ms: kept at 200)
        const bool np2 = hift_precision() == 2;
        const size_t npl = np2 ? 2 : 3;
#define CV2_C6_GO(FT_)                                                                                          \
        do {                                                                                                      \
            if (np2) {                                                                                            \
                if (taps_ct == 3) hipLaunchKernelGGL((k_conv6<FT_, 3, 2>), g_, dim3(256), sm, s, a);              \
                else if (taps_ct == 7) hipLaunchKernelGGL((k_conv6<FT_, 7, 2>), g_, dim3(256), sm, s, a);         \
                else if (taps_ct == 11) hipLaunchKernelGGL((k_conv6<FT_, 11, 2>), g_, dim3(256), sm, s, a);       \
                else hipLaunchKernelGGL((k_conv6<FT_, 0, 2>), g_, dim3(256), sm, s, a);                           \
            } else {                                                                                              \
                if (taps_ct == 3) hipLaunchKernelGGL((k_conv6<FT_, 3>), g_, dim3(256), sm, s, a);                 \
                else if (taps_ct == 7) hipLaunchKernelGGL((k_conv6<FT_, 7>), g_, dim3(256), sm, s, a);            \
                else if (taps_ct == 11) hipLaunchKernelGGL((k_conv6<FT_, 11>), g_, dim3(256), sm, s, a);          \
                else hipLaunchKernelGGL((k_conv6<FT_, 0>), g_, dim3(256), sm, s, a);                              \
            }                                                                                                     \
        } while (0)
        if (bt64 && blocks128 < bt64_max) {
            const size_t sm = (size_t)(64 + (cw.taps - 1) * cw.dil) * C6_LD * 2 * npl;
            const dim3 g_ = grid_for(64, true);                // (sets a.xcd_ch first)
            CV2_C6_GO(1);
            CV2_LAUNCH_CHECK();
            return 0;
        }
        const size_t sm = (size_t)(CV_BT + (cw.taps - 1) * cw.dil) * C6_LD * 2 * npl;
        const dim3 g_ = grid_for(CV_BT, true);                 // (sets a.xcd_ch first)
        CV2_C6_GO(2);
#undef CV2_C6_GO
        CV2_LAUNCH_CHECK();
        return 0;
    }
    {
        static const bool bt64 = !(getenv("CV2_HIFT_BT64") && getenv("CV2_HIFT_BT64")[0] == '0');
        const long blocks128 = (long)((L_out + CV_BT - 1) / CV_BT) * (cw.cout_pad / 64) * g_hz_n;
        if (bt64 && blocks128 < 200) {                 // (the f0 predictor's 512-channel layers at one utterance: 4 x 8 blocks)
            const size_t sm = (size_t)(64 + (cw.taps - 1) * cw.dil) * CV_LD * 4;
            { const dim3 g_ = grid_for(64, false); hipLaunchKernelGGL(k_conv<1>, g_, dim3(256), sm, s, a); }      // (sets a.xcd_ch first)
            CV2_LAUNCH_CHECK();
            return 0;
        }
    }
    const size_t sm = (size_t)(CV_BT + (cw.taps - 1) * cw.dil) * CV_LD * 4;
    { const dim3 g_ = grid_for(CV_BT, false); hipLaunchKernelGGL(k_conv<2>, g_, dim3(256), sm, s, a); }      // (sets a.xcd_ch first)
    CV2_LAUNCH_CHECK();
    return 0;
}

// one fused (conv1, conv2) pair (k_respair); false: the shapes are not the fused kernel's
static bool respair_ok(const cv2_conv& c1, const cv2_conv& c2, int C) {
    return (C == 64 || C == 128) && c1.w3 && c2.w3 && c1.cin == C && c1.cout == C && c2.cin == C && c2.cout == C && c1.cin_pad == C && c1.cout_pad == C &&
           c2.cin_pad == C && c2.cout_pad == C && c1.taps == c2.taps && (c1.taps & 1) && c1.taps <= 11 && c2.dil == 1 && c1.b && c2.b &&
           c1.pad_left == c1.dil * (c1.taps - 1) / 2 && c2.pad_left == (c2.taps - 1) / 2;
}
static int respair_launch(const cv2_conv& c1, const float* al1, const cv2_conv& c2, const float* al2, const float* x, int L, int C, float* out, int acc,
                          hipStream_t s) {
    PairArgs a{x, L, c1.w3, c1.b, al1, c1.taps, c1.dil, c2.w3, c2.b, al2, out, acc, g_hz_zs};
    const int BTo = 128 - (c1.taps - 1);
    const size_t rows1 = 128 + (size_t)(c1.taps - 1) * c1.dil, rowsB = 128 + c1.taps - 1;
    const bool np2 = hift_precision() == 2;
    const size_t sm = std::max(rows1 * C6_LD, rowsB * (size_t)(C + 8)) * 2 * (np2 ? 2 : 3);
    static const bool taps_env = !(getenv("CV2_HIFT_TAPS_CT") && getenv("CV2_HIFT_TAPS_CT")[0] == '0');
    const int t = taps_env ? c1.taps : 0;
    const dim3 g((L + BTo - 1) / BTo, 1, g_hz_n);
#define CV2_RP_GO(CT, NT)                                                                                       \
    do {                                                                                                          \
        if (np2) {                                                                                                \
            if (t == 3) hipLaunchKernelGGL((k_respair<CT, 3, 2>), g, dim3(NT), sm, s, a);                           \
            else if (t == 7) hipLaunchKernelGGL((k_respair<CT, 7, 2>), g, dim3(NT), sm, s, a);                      \
            else if (t == 11) hipLaunchKernelGGL((k_respair<CT, 11, 2>), g, dim3(NT), sm, s, a);                    \
            else hipLaunchKernelGGL((k_respair<CT, 0, 2>), g, dim3(NT), sm, s, a);                                  \
        } else {                                                                                                  \
            if (t == 3) hipLaunchKernelGGL((k_respair<CT, 3>), g, dim3(NT), sm, s, a);                              \
            else if (t == 7) hipLaunchKernelGGL((k_respair<CT, 7>), g, dim3(NT), sm, s, a);                         \
            else if (t == 11) hipLaunchKernelGGL((k_respair<CT, 11>), g, dim3(NT), sm, s, a);                       \
            else hipLaunchKernelGGL((k_respair<CT, 0>), g, dim3(NT), sm, s, a);                                     \
        }                                                                                                         \
    } while (0)
    if (C == 64) CV2_RP_GO(1, 256);
    else CV2_RP_GO(2, 512);
#undef CV2_RP_GO
    CV2_LAUNCH_CHECK();
    return 0;
}

// ResBlock.forward (generator.py:94-101); x read-only, result accumulated into `dst` with mode `acc`
static int resblock(cv2_hift* h, const cv2_resblock& rb, const float* x, int L, int C, float* dst, int acc, hipStream_t s) {
    // 64- and 128-channel stages: every (dilated conv, conv) pair is ONE launch (k_respair); a block reads its neighbours' input rows, so the
    // pairs ping-pong x -> ra -> xt -> dst instead of running in place.  CV2_HIFT_PAIR=0: the two launches per pair (A/B, diagnostics)
    static const bool pair_env = !(getenv("CV2_HIFT_PAIR") && getenv("CV2_HIFT_PAIR")[0] == '0');
    const int pair_dbg = g_pair_mode.load();
    const bool pair_on = pair_dbg < 0 ? pair_env : pair_dbg != 0;
    const bool fp32_only = hift_fp32_only();
    // (a short signal leaves most CUs without a block, and a fused block's two convolutions run one after the other: below pair_min blocks
    // the two launches with their 64-frame blocks are faster -- 90 frames: 2.25 against 2.51 ms per call with every pair fused; with the
    // threshold at 100 blocks 500 / 250 / 150 / 90 frames take 2.93 / 2.48 / 2.27 / 2.25 ms against 3.35 / 2.66 / 2.40 / 2.25 unfused)
    static const long pair_min = getenv("CV2_HIFT_PAIR_MIN") ? atol(getenv("CV2_HIFT_PAIR_MIN")) : 100;
    const long pair_blocks = (long)((L + 117) / 118) * g_hz_n;
    if (pair_on && !fp32_only && pair_blocks >= pair_min && respair_ok(rb.c1[0], rb.c2[0], C) && respair_ok(rb.c1[1], rb.c2[1], C) && respair_ok(rb.c1[2], rb.c2[2], C)) {
        float* bufs[3] = {h->ra, h->xt, dst};
        const float* in = x;
        for (int i = 0; i < 3; i++) {
            if (respair_launch(rb.c1[i], rb.a1[i], rb.c2[i], rb.a2[i], in, L, C, bufs[i], i == 2 ? acc : ACC_STORE, s)) return -1;
            in = bufs[i];
        }
        return 0;
    }
    for (int i = 0; i < 3; i++) {
        const float* in = i == 0 ? x : h->ra;
        if (conv_launch(rb.c1[i], in, L, h->xt, C, 0, L, PRE_SNAKE, rb.a1[i], 0.f, nullptr, 0, POST_NONE, ACC_STORE, s)) return -1;
        float* o = i == 2 ? dst : h->ra;
        if (conv_launch(rb.c2[i], h->xt, L, o, C, 0, L, PRE_SNAKE, rb.a2[i], 0.f, in, C, POST_NONE, i == 2 ? acc : ACC_STORE, s)) return -1;
    }
    return 0;
}

__global__ void k_set_seed(uint32_t* p, uint32_t lo, uint32_t hi) { if (threadIdx.x == 0) { p[0] = lo; p[1] = hi; } }

static int hift_run(cv2_hift* h, const float* mel, int32_t T, const float* cache_source, int32_t n_cache, const float* noise, uint64_t seed,
                    const uint32_t* seed_dev, float* wav, float* source, hipStream_t s);

extern "C" int cv2_hift_inference(cv2_hift* h, const float* mel, int32_t T, const float* cache_source, int32_t n_cache,
                                  const float* noise, uint64_t seed, float* wav, float* source, void* stream) {
    CV2_CHECK(h && mel && wav && source, "cv2_hift_inference: null argument");
    CV2_CHECK(T >= 2 && T <= h->d.max_frames, "cv2_hift_inference: T=%d out of range (max %d)", T, h->d.max_frames);
    CV2_CHECK(n_cache >= 0 && n_cache <= 480 * T && (n_cache == 0 || cache_source), "cv2_hift_inference: bad cache_source");
    hipStream_t s = (hipStream_t)stream;
    g_hz_n = 1; g_hz_zs = 0;
    static const bool graphs_off = getenv("CV2_HIFT_GRAPH") && getenv("CV2_HIFT_GRAPH")[0] == '0';
    if (noise || T > HG_MAX_T || graphs_off) return hift_run(h, mel, T, cache_source, n_cache, noise, seed, nullptr, wav, source, s);
    // ---- graph path
    const long key = ((long)T << 32) | (long)n_cache;
    auto it = h->graphs.find(key);
    if (it == h->graphs.end()) {
        if ((int)h->graphs.size() >= HG_MAX_GRAPHS) {                 // bounded: drop one (chunk shapes of a serving process are few)
            (void)hipGraphExecDestroy(h->graphs.begin()->second);
            h->graphs.erase(h->graphs.begin());
        }
        hipGraph_t g;
        CapStream cap(h->cap_stream);
        if (!cap.s) return hift_run(h, mel, T, cache_source, n_cache, noise, seed, nullptr, wav, source, s);
        CV2_HIP(hipStreamBeginCapture(cap.s, hipStreamCaptureModeThreadLocal));
        const int rc = hift_run(h, h->g_mel, T, n_cache ? h->g_cs : nullptr, n_cache, nullptr, 0, h->g_seed, h->g_wav, h->g_src, cap.s);
        const hipError_t e = hipStreamEndCapture(cap.s, &g);
        if (rc) return rc;
        CV2_HIP(e);
        hipGraphExec_t ge;
        CV2_HIP(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CV2_HIP(hipGraphDestroy(g));
        it = h->graphs.emplace(key, ge).first;
    }
    CV2_HIP(hipMemcpyAsync(h->g_mel, mel, (size_t)T * 80 * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (n_cache) CV2_HIP(hipMemcpyAsync(h->g_cs, cache_source, (size_t)n_cache * sizeof(float), hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(k_set_seed, dim3(1), dim3(64), 0, s, h->g_seed, (uint32_t)seed, (uint32_t)(seed >> 32));
    CV2_HIP(hipGraphLaunch(it->second, s));
    CV2_HIP(hipMemcpyAsync(wav, h->g_wav, (size_t)480 * T * sizeof(float), hipMemcpyDeviceToDevice, s));
    CV2_HIP(hipMemcpyAsync(source, h->g_src, (size_t)480 * T * sizeof(float), hipMemcpyDeviceToDevice, s));
    CV2_LAUNCH_CHECK();
    return 0;
}

// Test hook: the f0 track (Hz, [T]) of the engine's LAST cv2_hift_inference call, copied to `out` on `stream` (f0_predictor.py:55-58: the
// fixtures of tests/golden/hift_*.npz hold the reference's)
extern "C" int cv2_hift_debug_f0(cv2_hift* h, float* out, int32_t T, void* stream) {
    CV2_CHECK(h && out && T >= 1 && T <= h->d.max_frames, "cv2_hift_debug_f0: bad argument");
    CV2_HIP(hipMemcpyAsync(out, h->f0, (size_t)T * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}

// n chunks of the SAME shape (T frames, n_cache cached source samples) as ONE set of launches: every kernel runs with gridDim.z = n, lane z
// on its own copy of the workspace.  The streaming scheduler's rounds hold one chunk per stream, all of one shape; as separate calls they
// are 8 x 330 small launches on four HIP streams (17 ms per round of 8), 31 % of a round's kernel time.
extern "C" int cv2_hift_inference_batch(cv2_hift* h, int32_t n, const float* const* mel, int32_t T, const float* const* cache_source, int32_t n_cache,
                                        const uint64_t* seeds, float* const* wav, float* const* source, void* stream) {
    CV2_CHECK(h && mel && seeds && wav && source, "cv2_hift_inference_batch: null argument");
    CV2_CHECK(n >= 1 && n <= (h->d.lanes > 1 ? h->d.lanes : 1), "cv2_hift_inference_batch: %d chunks, engine has %d lanes", n, h->d.lanes > 1 ? h->d.lanes : 1);
    CV2_CHECK(T >= 2 && T <= h->d.max_frames && T <= HG_MAX_T, "cv2_hift_inference_batch: T=%d out of range (max %d)", T, h->d.max_frames < HG_MAX_T ? h->d.max_frames : HG_MAX_T);
    CV2_CHECK(n_cache >= 0 && n_cache <= 480 * T && (n_cache == 0 || cache_source), "cv2_hift_inference_batch: bad cache_source");
    hipStream_t s = (hipStream_t)stream;
    const long zs = h->lane_floats;
    const long key = ((long)T << 32) | ((long)n_cache << 8) | (long)n | (1l << 62);
    auto it = h->graphs.find(key);
    if (it == h->graphs.end()) {
        if ((int)h->graphs.size() >= HG_MAX_GRAPHS) {
            (void)hipGraphExecDestroy(h->graphs.begin()->second);
            h->graphs.erase(h->graphs.begin());
        }
        hipGraph_t g;
        CapStream cap(h->cap_stream);
        CV2_CHECK(cap.s, "cv2_hift_inference_batch: no stream to capture on");
        CV2_HIP(hipStreamBeginCapture(cap.s, hipStreamCaptureModeThreadLocal));
        g_hz_n = n; g_hz_zs = zs;
        const int rc = hift_run(h, h->g_mel, T, n_cache ? h->g_cs : nullptr, n_cache, nullptr, 0, h->g_seed, h->g_wav, h->g_src, cap.s);
        g_hz_n = 1; g_hz_zs = 0;
        const hipError_t e = hipStreamEndCapture(cap.s, &g);
        if (rc) return rc;
        CV2_HIP(e);
        hipGraphExec_t ge;
        CV2_HIP(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CV2_HIP(hipGraphDestroy(g));
        it = h->graphs.emplace(key, ge).first;
    }
    for (int z = 0; z < n; z++) {
        CV2_CHECK(mel[z] && wav[z] && source[z] && (n_cache == 0 || cache_source[z]), "cv2_hift_inference_batch: null pointer for chunk %d", z);
        CV2_HIP(hipMemcpyAsync(h->g_mel + z * zs, mel[z], (size_t)T * 80 * sizeof(float), hipMemcpyDeviceToDevice, s));
        if (n_cache) CV2_HIP(hipMemcpyAsync(h->g_cs + z * zs, cache_source[z], (size_t)n_cache * sizeof(float), hipMemcpyDeviceToDevice, s));
        hipLaunchKernelGGL(k_set_seed, dim3(1), dim3(64), 0, s, reinterpret_cast<uint32_t*>(reinterpret_cast<float*>(h->g_seed) + z * zs), (uint32_t)seeds[z], (uint32_t)(seeds[z] >> 32));
    }
    CV2_HIP(hipGraphLaunch(it->second, s));
    for (int z = 0; z < n; z++) {
        CV2_HIP(hipMemcpyAsync(wav[z], h->g_wav + z * zs, (size_t)480 * T * sizeof(float), hipMemcpyDeviceToDevice, s));
        CV2_HIP(hipMemcpyAsync(source[z], h->g_src + z * zs, (size_t)480 * T * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    CV2_LAUNCH_CHECK();
    return 0;
}

static int hift_run(cv2_hift* h, const float* mel, int32_t T, const float* cache_source, int32_t n_cache, const float* noise, uint64_t seed,
                    const uint32_t* seed_dev, float* wav, float* source, hipStream_t s) {
    const cv2_hift_weights& w = h->w;
    const int Ls = 480 * T, F = Ls / 4 + 1;
    const int Z = g_hz_n; const long zs = g_hz_zs;
    hipLaunchKernelGGL(k_mel_tm, dim3((T * 80 + 255) / 256, 1, Z), dim3(256), 0, s, mel, h->melT, (int)T, zs);
    // f0 predictor: 5 x (conv k3 + ELU), Linear, abs
    {
        const float* in = h->melT; float* bufs[2] = {h->f0a, h->f0b};
        for (int i = 0; i < 5; i++) {
            if (conv_launch(w.f0_conv[i], in, T, bufs[i & 1], 512, 0, T, PRE_NONE, nullptr, 0.f, nullptr, 0, POST_ELU, ACC_STORE, s)) return -1;
            in = bufs[i & 1];
        }
        hipLaunchKernelGGL(k_f0_head, dim3((T + 3) / 4, 1, Z), dim3(256), 0, s, in, w.f0_w, w.f0_b, h->f0, (int)T, zs);
    }
    // source
    hipLaunchKernelGGL(k_phase, dim3(1, 1, Z), dim3(256), 0, s, (const float*)h->f0, h->phase, (int)T, zs);
    {
        SrcArgs a{h->f0, h->phase, T, noise, (uint32_t)seed, (uint32_t)(seed >> 32), seed_dev, w.src_w, w.src_b, cache_source, n_cache, source, zs};
        hipLaunchKernelGGL(k_source, dim3((Ls + 255) / 256, 1, Z), dim3(256), 0, s, a);
    }
    hipLaunchKernelGGL(k_stft, dim3(((long)F * 9 + 255) / 256, 1, Z), dim3(256), 0, s, (const float*)source, Ls, h->sstft, F, zs);
    // decode
    if (conv_launch(w.conv_pre, h->melT, T, h->xpre, 512, 0, T, PRE_NONE, nullptr, 0.f, nullptr, 0, POST_NONE, ACC_STORE, s)) return -1;
    const int ups_u[3] = {8, 5, 3}, chans[4] = {512, 256, 128, 64};
    const int sd_k[3] = {30, 6, 1}, sd_s[3] = {15, 3, 1}, sd_p[3] = {7, 1, 0};
    const float* xin = h->xpre;
    int Lin = T;
    for (int i = 0; i < 3; i++) {
        const int C = chans[i + 1], u = ups_u[i];
        int L = Lin * u;
        // leaky_relu(0.1) -> ConvTranspose1d as a polyphase conv: output [Lin][u*C] == [L][C]
        const long shift = i == 2 ? C : 0;                                            // reflect pad: frame t lands on row t+1
        if (conv_launch(w.ups[i], xin, Lin, h->x, u * C, shift, Lin, PRE_LRELU, nullptr, 0.1f, nullptr, 0, POST_NONE, ACC_STORE, s)) return -1;
        if (i == 2) { hipLaunchKernelGGL(k_reflect_row0, dim3(1, 1, Z), dim3(256), 0, s, h->x, C, zs); L += 1; }
        // source branch: source_down -> source_resblock, added into x
        {
            CV2_CHECK((F + 2 * sd_p[i] - sd_k[i]) / sd_s[i] + 1 == L, "hift: source_down length %d != %d", (F + 2 * sd_p[i] - sd_k[i]) / sd_s[i] + 1, L);
            static const bool sd_scalar = getenv("CV2_HIFT_SD_SCALAR") && getenv("CV2_HIFT_SD_SCALAR")[0] == '1';      // A/B switch (diagnostics)
            if (w.sd_conv[i].w3 && !sd_scalar && !hift_fp32_only()) {      // (CV2_HIFT_FP32=1: k_conv knows no flat windows -> the scalar kernel)
                // the strided convolution over [F][18] rows as a 1-tap convolution over flat windows of 18 k floats on the matrix cores (k_conv6)
                if (conv_launch(w.sd_conv[i], h->sstft, L, h->sd, C, 0, L, PRE_NONE, nullptr, 0.f, nullptr, 0, POST_NONE, ACC_STORE, s,
                                18 * sd_s[i], -18l * sd_p[i], 18l * F)) return -1;
            } else {
                SdArgs a{h->sstft, F, w.sd_w[i], w.sd_b[i], C, sd_k[i], sd_s[i], sd_p[i], h->sd, L, zs};
                hipLaunchKernelGGL(k_source_down, dim3(((long)L * C + 255) / 256, 1, Z), dim3(256), 0, s, a);
            }
            if (resblock(h, w.src_rb[i], h->sd, L, C, h->x, ACC_ADD, s)) return -1;
        }
        // MRF: mean of 3 ResBlocks
        for (int j = 0; j < 3; j++)
            if (resblock(h, w.rb[i * 3 + j], h->x, L, C, h->sum, j == 0 ? ACC_STORE : j == 1 ? ACC_ADD : ACC_ADD_DIV3, s)) return -1;
        // next stage input = MRF output (the next upsample reads `sum` and writes `x`: never in place)
        xin = h->sum; Lin = L;
    }
    // leaky_relu(0.01) -> conv_post -> exp / sin -> iSTFT -> clamp
    if (conv_launch(w.conv_post, xin, Lin, h->post, 18, 0, Lin, PRE_LRELU, nullptr, 0.01f, nullptr, 0, POST_NONE, ACC_STORE, s)) return -1;
    hipLaunchKernelGGL(k_istft_frames, dim3(((long)F * 16 + 255) / 256, 1, Z), dim3(256), 0, s, (const float*)h->post, F, h->frames, zs);
    hipLaunchKernelGGL(k_istft_ola, dim3((Ls + 255) / 256, 1, Z), dim3(256), 0, s, (const float*)h->frames, F, wav, Ls, 0.99f, zs);
    CV2_LAUNCH_CHECK();
    return 0;
}

__global__ void k_fade(float* x, const float* old_tail, const float* win, int w) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < w) x[i] = x[i] * win[i] + old_tail[i] * win[w + i];
}
extern "C" int cv2_fade_in_out(float* fade_in, const float* old_tail, const float* window, int32_t w, void* stream) {
    CV2_CHECK(fade_in && old_tail && window && w > 0, "cv2_fade_in_out: bad argument");
    hipLaunchKernelGGL(k_fade, dim3((w + 255) / 256), dim3(256), 0, (hipStream_t)stream, fade_in, old_tail, window, (int)w);
    CV2_LAUNCH_CHECK();
    return 0;
}

// cli/model.py:328-330: tts_mel = F.interpolate(tts_mel, size=int(T / speed), mode='linear') ahead of the vocoder (non-streaming calls
// with speed != 1).  align_corners = False: source position scale (j + 0.5) - 0.5 clamped at 0, scale = n_in / n_out in fp32, right
// neighbour clamped at the last frame.  rows = channels (80 per utterance).
__global__ __launch_bounds__(256) void k_interp_linear(const float* __restrict__ in, float* __restrict__ out, int n_in, int n_out, float scale) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n_out) return;
    const float* x = in + (size_t)blockIdx.y * n_in;
    const float src = fmaxf(scale * ((float)j + 0.5f) - 0.5f, 0.f);
    const int i0 = min((int)src, n_in - 1), i1 = min(i0 + 1, n_in - 1);
    const float w1 = src - (float)i0, w0 = 1.f - w1;
    out[(size_t)blockIdx.y * n_out + j] = w0 * x[i0] + w1 * x[i1];
}
extern "C" int cv2_interp_linear(const float* in, float* out, int32_t rows, int32_t n_in, int32_t n_out, void* stream) {
    CV2_CHECK(in && out && rows >= 1 && rows <= 65535 && n_in >= 1 && n_out >= 1, "cv2_interp_linear: bad argument");
    hipLaunchKernelGGL(k_interp_linear, dim3((n_out + 255) / 256, rows), dim3(256), 0, (hipStream_t)stream, in, out, (int)n_in, (int)n_out,
                       (float)n_in / (float)n_out);
    CV2_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ test hook: the conv pre-activations (Snake / leaky ReLU) as k_conv evaluates them
__global__ void k_dbg_pre(const float* x, float* out, int n, int pre, float alpha, float slope) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = pre_apply(x[i], pre, alpha, slope);
}
extern "C" int cv2_dbg_pre(const float* x, float* out, int64_t n, int32_t pre, float alpha, float slope, void* stream) {
    CV2_CHECK(x && out && n > 0 && n < (1ll << 30), "cv2_dbg_pre: bad argument");
    hipLaunchKernelGGL(k_dbg_pre, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, out, (int)n, (int)pre, alpha, slope);
    CV2_LAUNCH_CHECK();
    return 0;
}

