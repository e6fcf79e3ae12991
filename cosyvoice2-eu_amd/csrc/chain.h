// One-row decode step as ONE launch (llm.hip: k_step): for every layer the roles Q (RMSNorm + QKV + RoPE), A (attention over one
// 128-key tile of one kv head, on the matrix cores), O (tile merge + O projection), GU (gate/up + SwiGLU), D (down projection, split-K), then the head (final norm + llm_decoder).
//
// Why: at one row a layer's five weight-streaming kernels are latency-bound (1.6 us dispatch + ~1 us first dependent load + ~1 us
// of weight stream each, 24.6 us per layer for 30 MB).  Weights do not depend on activations: when the GEMVs share a launch, every
// block requests its weight fragments when it is dispatched, i.e. while the blocks in front of it are still computing, and what is
// left behind each dependency is the hand-off plus ~30 MFMAs.  The hardware dispatcher is the scheduler: blocks are laid out in
// dependency order (layer by layer, role by role), a finished block frees its slot for the next undispatched one, so the chip
// holds about one layer of look-ahead whose weights are already in registers when their operand arrives.
//
// Hand-offs (cdna_hip_programming.md Guideline 16, form R2): the data is the flag.  A value travels as one naturally aligned
// 8-byte granule {fp32 bits, tag = epoch}, written by ONE write-through (sc1) store and read with sc1 loads; no fence, no separate
// flag.  epoch = a device counter k_sample advances once per decode step; every layer has granule buffers of its own, so a granule
// of an earlier step never matches and nothing is re-initialised between steps (zeroed once at create, epoch starts at 1).
// A consumer first polls ONE granule of the producer expected last (cheap beside the weight streams), then sweeps its whole
// operand and checks every tag; the sweep repeats until all match.
//
// Forward progress: a block only waits for blocks with LOWER indices, and blocks are dispatched in index order (per XCD too), so
// the lowest unfinished block is always resident and never waits for an undispatched one.  Every wait is bounded (0.2 s) and
// reports through the slot's CV2_ST_ERR (3) instead of hanging; once one block has given up every other wait ends at its next check.
#pragma once
#include "skinny.h"
#ifndef R1_T_OPERAND
#define R1_T_OPERAND do { } while (0)
#define R1_T(i) do { } while (0)
#endif

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

#define CH_NP 2                                 // K splits of the down projection inside k_step (partials folded by the next consumer)
#ifndef SWEEP_SLEEP
#define SWEEP_SLEEP 1                           // s_sleep units between two sweeps of a direct poll (measured: 4 -> 1: -2.5 us per step)
#endif
#ifndef POLL_GAP
#define POLL_GAP 2                              // s_sleep units between two polls of a sentinel granule
#endif
#define GRAN_TIMEOUT_TICKS 20000000ull     // s_memrealtime runs at 100 MHz: 0.2 s

// (by value on purpose: clang lowers __builtin_bit_cast of a vector ELEMENT lvalue as a load from the vector's address, i.e. element 0)
__device__ __forceinline__ float u2f(unsigned u) { return __builtin_bit_cast(float, u); }

struct Gran {                 // all granule buffers of the engine behind one buffer descriptor (32-bit byte offsets)
    __amdgpu_buffer_rsrc_t rsrc; u64* base; unsigned epoch;
    int* err;                 // the error word every wait of the chain re-reads (the chain's first slot)
    int* err2;                // the slot of a block's second row (k_step2), else == err
    int* errs[4];             // every slot the block's CHAIN serves (k_step4: up to four rows; else {err, err2, err, err2}): a block that gives up
                              // flags them all -- its consumers see valid tags on whatever it published and never time out themselves, so a row the
                              // block did not flag would draw from garbage logits and commit (k_sample checks CV2_ST_ERR per slot)
    bool spec;                // several rows per launch: most blocks are dispatched AFTER their operand was published -- sweep once before
                              // any sentinel wait (one round trip instead of two when the operand is there, one wasted sweep when not)
    __device__ __forceinline__ void init(u64* b, unsigned bytes, unsigned ep, int* e, bool sp = false) {
        rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)b, 0, (int)bytes, 0x00020000);
        base = b; epoch = ep; err = err2 = e; spec = sp;
        errs[0] = errs[1] = errs[2] = errs[3] = e;
    }
    __device__ __forceinline__ void set_err2(int* e2) { err2 = e2; errs[1] = errs[3] = e2; }
    template <class F>
    __device__ __forceinline__ bool try_once(F f) const { return spec && __all(f()); }
    // idx = granule index from the start of the buffer
    __device__ __forceinline__ void store(unsigned idx, float v) const {
        __hip_atomic_store((gu64*)(base + idx), ((u64)epoch << 32) | (u64)__builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __device__ __forceinline__ u32x4 ld2(unsigned idx) const {      // two granules {v0, tag0, v1, tag1}; idx even
        return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, idx * 8u, 0, 16));   // aux 16 = sc1
    }
    __device__ __forceinline__ bool ld8(unsigned idx, f32x8& v) const {   // 8 consecutive granules; true when every tag matches
        u32x4 x[4];
#pragma unroll
        for (int i = 0; i < 4; i++) x[i] = ld2(idx + 2 * i);
        bool ok = true;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            ok &= x[i][1] == epoch && x[i][3] == epoch;
            v[2 * i] = u2f(x[i][0]); v[2 * i + 1] = u2f(x[i][2]);
        }
        return ok;
    }
    __device__ __forceinline__ bool ld4(unsigned idx, f32x4& v) const {   // 4 consecutive granules
        const u32x4 x0 = ld2(idx), x1 = ld2(idx + 2);
        v[0] = u2f(x0[0]); v[1] = u2f(x0[2]);
        v[2] = u2f(x1[0]); v[3] = u2f(x1[2]);
        return x0[1] == epoch && x0[3] == epoch && x1[1] == epoch && x1[3] == epoch;
    }
    __device__ __forceinline__ void fail() const {
        if ((threadIdx.x & 63) == 0) {
            __hip_atomic_store((__attribute__((address_space(1))) int*)err, 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store((__attribute__((address_space(1))) int*)err2, 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int i = 0; i < 4; i++) __hip_atomic_store((__attribute__((address_space(1))) int*)errs[i], 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // a wait gives up when its own time is over OR when any block of the launch has given up (the flag is re-read every 16 polls):
    // one broken dependency then costs one timeout, not one per block behind it
    __device__ __forceinline__ bool give_up(u64 t0) const {
        if (__hip_atomic_load((const __attribute__((address_space(1))) int*)err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 3) return true;
        if (__builtin_amdgcn_s_memrealtime() - t0 > GRAN_TIMEOUT_TICKS) { fail(); return true; }
        return false;
    }
    // one wave waits until the n <= 64 granules base[idx0 + lane * stride] carry the epoch (lanes >= n idle).  One poll in flight per
    // wave: two or three in flight from ONE wave measured slower (363 -> 374 -> 381 us per step; hipcc also serialises them, the
    // result registers are reused), and the gap between polls wants to be short (s_sleep 2: 361.6, 6: 362.9, 10: 368.5, 24: 389.9).
    __device__ __forceinline__ bool wait(unsigned idx0, unsigned stride, int n) const {
        const int lane = threadIdx.x & 63;
        const u64 t0 = __builtin_amdgcn_s_memrealtime();
        for (unsigned spin = 0;; spin++) {
            bool ok = true;
            if (lane < n) {
                const u64 x = __hip_atomic_load((const gu64*)(base + idx0 + lane * stride), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = (unsigned)(x >> 32) == epoch;
            }
            if (__all(ok)) return true;
            if ((spin & 15) == 15 && give_up(t0)) return false;
            __builtin_amdgcn_s_sleep(POLL_GAP);
        }
    }
    // the same with an index function: lane i < n polls granule idx(i)
    template <class F>
    __device__ __forceinline__ bool wait_f(int n, F idx) const {
        const int lane = threadIdx.x & 63;
        const u64 t0 = __builtin_amdgcn_s_memrealtime();
        const unsigned my = lane < n ? idx(lane) : 0u;
        for (unsigned spin = 0;; spin++) {
            bool ok = true;
            if (lane < n) {
                const u64 x = __hip_atomic_load((const gu64*)(base + my), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = (unsigned)(x >> 32) == epoch;
            }
            if (__all(ok)) return true;
            if ((spin & 15) == 15 && give_up(t0)) return false;
            __builtin_amdgcn_s_sleep(6);
        }
    }
    // wave-level retry loop around a per-lane sweep `f` (returns the lane's ok)
    template <class F>
    __device__ __forceinline__ void sweep(F f) const {
        const u64 t0 = __builtin_amdgcn_s_memrealtime();
        for (unsigned spin = 0;; spin++) {
            if (__all(f())) return;
            if ((spin & 15) == 15 && give_up(t0)) return;
            __builtin_amdgcn_s_sleep(SWEEP_SLEEP);
        }
    }
};

// ---- operand providers of row1_core: issue() runs before the weight loads, finish() after (all threads call both)

// x = base + p0 + p1 (the residual stream entering a layer: x_mid of the previous layer + its CH_NP down-projection
// partials, summed in this order), every vector [H] granules; or a plain fp32 vector (layer 0: the embedding k_sample left).
struct OpFold {
    const Gran* G; unsigned xg, dg; int H;      // granule indices of x_mid [H] and the partials [CH_NP][H] of the previous layer
    const float* plain; int dbg; unsigned arm;
    __device__ __forceinline__ void issue(int, int, bool) {}
    // one value at column c (call wave-uniformly)
    __device__ __forceinline__ float get1(int c, bool active) const {
        float v = 0.f;
        if (plain) { if (active) v = plain[c]; return v; }
        G->sweep([&]() {
            if (!active) return true;
            u64 x[1 + CH_NP];
            x[0] = __hip_atomic_load((const gu64*)(G->base + xg + c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int i = 0; i < CH_NP; i++) x[1 + i] = __hip_atomic_load((const gu64*)(G->base + dg + i * H + c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bool ok = true;
#pragma unroll
            for (int i = 0; i <= CH_NP; i++) ok &= (unsigned)(x[i] >> 32) == G->epoch;
            v = __builtin_bit_cast(float, (unsigned)x[0]);
#pragma unroll
            for (int i = 0; i < CH_NP; i++) v += __builtin_bit_cast(float, (unsigned)x[1 + i]);
            return ok;
        });
        return v;
    }
    static constexpr int IW = 4;                 // operand columns per thread: 5 vectors x 4 granules in flight per lane
    // one non-blocking look at the operand (k_step4: several rows' loads in flight together); false: not all there yet
    __device__ __forceinline__ bool attempt(int k, bool active, f32x8& r) const {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        bool ok = true;
        if (plain) { if (active) v = *reinterpret_cast<const f32x4*>(plain + k); }
        else if (active) {
            f32x4 p[CH_NP];
            ok = G->ld4(xg + k, v);
#pragma unroll
            for (int i = 0; i < CH_NP; i++) ok &= G->ld4(dg + i * H + k, p[i]);
#pragma unroll
            for (int i = 0; i < CH_NP; i++) v += p[i];
        }
        r = (f32x8){v[0], v[1], v[2], v[3], 0.f, 0.f, 0.f, 0.f};
        return ok;
    }
    __device__ __forceinline__ f32x8 finish(int k, int wave, int nitems, bool active, char*) {
        f32x8 r = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (wave * 64 >= nitems) return r;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (plain) {
            if (active) v = *reinterpret_cast<const f32x4*>(plain + k);
        } else {
            auto body = [&]() {
                bool ok = true;
                if (active) {
                    f32x4 p[CH_NP];
                    ok = G->ld4(xg + k, v);
#pragma unroll
                    for (int i = 0; i < CH_NP; i++) ok &= G->ld4(dg + i * H + k, p[i]);
#pragma unroll
                    for (int i = 0; i < CH_NP; i++) v += p[i];
                }
                return ok;
            };
            if (!G->try_once(body)) {
                G->wait(arm, 0, 1);                                 // armed: the previous layer's h has been published (the partials follow ~2 us later);
                                                                    // from here the sweep itself polls (few consumer blocks: Q 36, head 411 once per step)
                G->sweep(body);
            }
        }
        r[0] = v[0]; r[1] = v[1]; r[2] = v[2]; r[3] = v[3];
        return r;
    }
};
// a single granule vector (x_mid for gate/up, h for the down projection)
template <int IW_>
struct OpGran {
    static constexpr int IW = IW_;
    const Gran* G; unsigned g0, sentinel, sstride; int dbg;     // sentinel: one granule of the FIRST producer of this vector
    int grp = 0;                                                 // k_step2: which half of the block runs this provider (own arming word; `wave` is the wave within the half)
    bool nobar = false;                                          // k_step2: the caller has cleared the arming words behind a barrier of its own
    __device__ __forceinline__ void issue(int, int, bool) {}
    __device__ __forceinline__ bool attempt(int k, bool active, f32x8& v) const {
        v = (f32x8){0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (!active) return true;
        if (IW == 8) return G->ld8(g0 + k, v);
        f32x4 q = {0.f, 0.f, 0.f, 0.f};
        const bool ok = G->ld4(g0 + k, q);
        v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3];
        return ok;
    }
    __device__ __forceinline__ f32x8 finish(int k, int wave, int nitems, bool active, char* xch_) {
        f32x8 v = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        // armed by the first producer (cheap one-word polls while the vector is still far away), then the sweep itself polls: the
        // producers finish within ~0.6 us of each other, so this costs one or two extra sweeps and saves the round trip that a
        // wait for the LAST producer's granule would put in front of the sweep.  ONE wave per block polls the arming word (hundreds
        // of blocks wait for the same 128-byte line: every poll of it goes to the same memory channel); the others watch an LDS word.
        volatile int* armed = reinterpret_cast<volatile int*>(xch_) + grp;
        if (!nobar) {
            if (wave == 0 && (threadIdx.x & 63) == 0) *armed = 0;
            __syncthreads();
        }
        if (wave * 64 >= nitems) return v;
        f32x4 q = {0.f, 0.f, 0.f, 0.f};
        auto body = [&]() { return !active ? true : (IW == 8 ? G->ld8(g0 + k, v) : G->ld4(g0 + k, q)); };
        const bool have = G->try_once(body);
        if (wave == 0) {                 // (two to four staggered polling waves per block measured the same: 363.5 / 364.0 / 362.8 / 363.6 us per step --
            if (!have) G->wait(sentinel, 0, 1);     //  the arming is off the critical path, the sweep's own retries are what follows the last producer)
            if ((threadIdx.x & 63) == 0) *armed = 1;
        } else if (!have) {               // (bounded like every other wait: a leader that never arms must not hang the block)
            const u64 t0 = __builtin_amdgcn_s_memrealtime();
            for (unsigned spin = 0; *armed == 0; spin++) {
                __builtin_amdgcn_s_sleep(2);
                if ((spin & 1023) == 1023 && G->give_up(t0)) break;
            }
        }
        if (!have) G->sweep(body);
        if (IW != 8) { v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3]; }
        return v;
    }
};
// the attention output: the live 64-key tiles' unnormalised partials {o[rep * 64], (max, sum)[rep]} per kv head, combined here
// (flash-decoding merge).  A thread owns 4 columns and walks every live tile, AT_CHUNK tiles' loads in flight at a time.
#define AT_GSTRIDE 464                         // granules per (tile, kv head): rep * 64 outputs, then rep x {max, sum}; rep <= 7
#ifndef AT_CHUNK
#define AT_CHUNK 4                              // tiles per sweep of the O role (new-token partial folded first: 3: 339.5 / 370.1 us per step at 3-5 / 9-11 live tiles, 4: 335.3 / 360.5, 5: 336.2 / 367.4, 6: 361.8 / 399.2)
#endif
#ifndef O_WAIT_ALL
#define O_WAIT_ALL 0
#endif
#define AT_TILE 128                            // keys per attention block: two 64-key groups of 256 threads
struct OpAtt {
    static constexpr int IW = 4;
    const Gran* G; unsigned ag; int n_kv, rep, cnt; int dbg;      // cnt = live tiles
    const OpFold* fold; int col0;                                  // the block's 16 residual columns, fetched by 16 idle lanes beside the first chunk
    unsigned qg, kvg;                                              // q [n_q * 64] and the new token's key / value rows [2][n_kv][64] of this layer
    int t0 = 0, grp = 0;                                           // k_step2: first thread of the half that runs this provider, its residual slot
    __device__ __forceinline__ void issue(int, int, bool) {}
    __device__ __forceinline__ f32x8 finish(int k, int, int, bool act, char* xch_) {
        const int tid = (int)threadIdx.x - t0;
        const int hd = k >> 6, g = hd / rep, hh = hd - g * rep;
        const bool res = tid >= 240;                               // nitems = 224: the last 32 lanes own no item
        float* xch = reinterpret_cast<float*>(xch_) + grp * 16;
        // the new token attends to itself: score = q . k_new / 8 per head, value v_new -- one more partial {o = v_new, max = score,
        // sum = 1}.  Its three vectors (and the block's 16 residual columns) were published by the Q role ~2 us before the first
        // attention tile: they are fetched and folded FIRST, behind a sentinel of their own (the last value granule), so that the
        // tiles' sweep carries only tiles and the dot product is off the path behind the attention.
        float M, den;
        f32x4 acc;
        {
            f32x4 q4 = {0.f, 0.f, 0.f, 0.f}, k4 = q4, v4 = q4;
            float rv = 0.f;
            if (!G->spec) G->wait(kvg + 2 * n_kv * 64 - 1, 0, 1);        // (spec: the sweep below polls by itself)
            G->sweep([&]() {
                bool ok = true;
                if (act) {
                    ok = G->ld4(qg + k, q4);
                    ok &= G->ld4(kvg + g * 64 + (k & 63), k4);
                    ok &= G->ld4(kvg + (n_kv + g) * 64 + (k & 63), v4);
                } else if (res && !fold->plain) {
                    const int c = col0 + (tid & 15);
                    u64 x[1 + CH_NP];
                    x[0] = __hip_atomic_load((const gu64*)(G->base + fold->xg + c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                    for (int i = 0; i < CH_NP; i++)
                        x[1 + i] = __hip_atomic_load((const gu64*)(G->base + fold->dg + i * fold->H + c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                    for (int i = 0; i <= CH_NP; i++) ok &= (unsigned)(x[i] >> 32) == G->epoch;
                    rv = __builtin_bit_cast(float, (unsigned)x[0]);
#pragma unroll
                    for (int i = 0; i < CH_NP; i++) rv += __builtin_bit_cast(float, (unsigned)x[1 + i]);
                }
                return ok;
            });
            if (res) xch[1200 + (tid & 15)] = fold->plain ? fold->plain[col0 + (tid & 15)] : rv;
            float sc = (q4[0] * k4[0] + q4[1] * k4[1]) + (q4[2] * k4[2] + q4[3] * k4[3]);
            sc += dpp_mov_f32<0x128, 0xf>(0.f, sc);            // row_ror 8 / 4 / 2 / 1: the 16 lanes of a head (64 columns) sum up
            sc += dpp_mov_f32<0x124, 0xf>(0.f, sc);
            sc += dpp_mov_f32<0x122, 0xf>(0.f, sc);
            sc += dpp_mov_f32<0x121, 0xf>(0.f, sc);
            M = sc * 0.125f; den = 1.f; acc = v4;
        }
        asm volatile("" : "+v"(M), "+v"(acc));                     // (done before the wait below, not sunk behind it)
        if (!G->spec) G->wait(ag + rep * 64, AT_GSTRIDE, O_WAIT_ALL ? cnt * n_kv : 1);   // every live (tile, head) pair's (max, sum) granule, or only the first tile's (then the sweep polls)
        for (int s0 = 0; s0 < cnt; s0 += AT_CHUNK) {               // (block-uniform trip count)
            f32x4 o[AT_CHUNK]; float m[AT_CHUNK], l[AT_CHUNK];
            G->sweep([&]() {
                bool ok = true;
                if (act) {
#pragma unroll
                    for (int i = 0; i < AT_CHUNK; i++) {
                        const int s = min(s0 + i, cnt - 1);        // slots past the count repeat the last tile (loaded, not merged)
                        const unsigned a0 = ag + (unsigned)(s * n_kv + g) * AT_GSTRIDE;
                        ok &= G->ld4(a0 + hh * 64 + (k & 63), o[i]);
                        const u32x4 ml = G->ld2(a0 + rep * 64 + hh * 2);
                        ok &= ml[1] == G->epoch && ml[3] == G->epoch;
                        m[i] = u2f(ml[0]); l[i] = u2f(ml[2]);
                    }
                }
                return ok;
            });
#pragma unroll
            for (int i = 0; i < AT_CHUNK; i++) {
                if (s0 + i < cnt) {
                    const float Mn = fmaxf(M, m[i]);
                    const float w0 = __expf(M - Mn), w1 = __expf(m[i] - Mn);
                    den = den * w0 + l[i] * w1;
                    acc = acc * w0 + o[i] * w1;
                    M = Mn;
                }
            }
        }
        const float inv = 1.f / den;
        f32x8 v = {acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv, 0.f, 0.f, 0.f, 0.f};
        return v;
    }
};

// LDS of one GEMV block: operand stage (hi / lo planes, 64 B per 32-wide k-step and plane: only column 0 of the MFMA's B operand
// is real at one row, every lane of a 16-lane quarter reads the same 16 bytes), exchange area of OpAtt, reduction slots.
#define R1_STAGE_BYTES(nks) ((nks) * 128)
#define R1_XCH_BYTES (128 * 10 * 4)
#define R1_THREADS 512
__device__ __host__ constexpr int r1_smem_bytes(int nks) { return R1_STAGE_BYTES(nks) + R1_XCH_BYTES + 32 + 8 * 4 * 16; }

// out[f] (f < NWR * 16) = sum_k W[(tile0 + (f / 16) * tstride) * 16 + f % 16][k] * x[k] over the k-steps [ks0, ks1),
// x = op's vector [* RMSNorm weight, scaled by the row's rstd].  Returns feature f's value in thread f (other threads: 0).
// 512 threads = NWR x NWK waves; every wave's weight fragments (<= MAXKS) are requested first.
// A weight fragment by a GLOBAL load: the layer table's pointers come out of memory, so the compiler sees generic pointers and would emit
// flat_load (which also counts in lgkmcnt: every LDS / scalar wait then waits for the weights in flight)
typedef const __attribute__((address_space(1))) s16x8* gs16x8p;
template <bool NT>
__device__ __forceinline__ s16x8 ld_wfrag(const s16x8* p) {
    gs16x8p g = (gs16x8p)(uintptr_t)p;
    return NT ? __builtin_nontemporal_load(g) : *g;
}
struct R1NoHook { __device__ __forceinline__ void operator()() const {} };
// `issued` runs right after the weight requests: the place for a role's own dependent loads (Q: position -> RoPE table).
// NT: the weight fragments are read once per step (one row: non-temporal loads keep them out of L2); false when several rows' blocks
// stream the same tile one after the other on one XCD (k_step<true>): the first brings it into that L2, the siblings hit there.
template <int NWR, int NWK, int MAXKS, bool NORM, bool NT, class OP, class HOOK = R1NoHook>
__device__ __forceinline__ float row1_core(const uint16_t* __restrict__ W, int tile0, int tstride, int KS, int K, int ks0, int ks1, OP& op,
                                           const float* norm_w, float eps, char* smem, HOOK issued = HOOK()) {
    static_assert(NWR * NWK == 8, "row1_core: 512 threads");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave % NWR, wk = wave / NWR;
    const int nks = ks1 - ks0;
    const int w0 = ks0 + (nks * wk) / NWK, w1 = ks0 + (nks * (wk + 1)) / NWK;
    constexpr int IW = OP::IW;                                    // operand columns per thread (4 or 8)
    const int nitems = nks * (32 / IW);                           // item i belongs to thread i (<= 512)
    const bool active = tid < nitems;
    const int k = ks0 * 32 + tid * IW;
    char* stage = smem;
    char* xch = smem + R1_STAGE_BYTES(nks);
    float* sqs = reinterpret_cast<float*>(xch + R1_XCH_BYTES);     // [8] per-wave sums of squares
    float* red = reinterpret_cast<float*>(xch + R1_XCH_BYTES + 32);   // [NWK][NWR][4 quarters][4]
    op.issue(tid, nitems, active);
    f32x8 g0 = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
    if (NORM && active) {
        if (IW == 8) g0 = *reinterpret_cast<const f32x8*>(norm_w + k);
        else { const f32x4 g4 = *reinterpret_cast<const f32x4*>(norm_w + k); g0[0] = g4[0]; g0[1] = g4[1]; g0[2] = g4[2]; g0[3] = g4[3]; }
    }
    const char* wbase = reinterpret_cast<const char*>(W) + ((size_t)(tile0 + wr * tstride) * KS + w0) * 1024;
    const unsigned wlane = lane * 16;
    s16x8 abuf[MAXKS];
    const int nw = w1 - w0;
#pragma unroll
    for (int i = 0; i < MAXKS; i++) {
        const s16x8* wp_ = reinterpret_cast<const s16x8*>(wbase + (size_t)(i < nw ? i : (nw > 0 ? nw - 1 : 0)) * 1024 + wlane);
        abuf[i] = ld_wfrag<NT>(wp_);
    }
    __builtin_amdgcn_sched_barrier(0);
    issued();
    f32x8 v = op.finish(k, wave, nitems, active, xch);
    R1_T_OPERAND;
    if (NORM) {
        float sq = 0.f;
        if (active) {
            sq = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
            if (IW == 8) sq += (v[4] * v[4] + v[5] * v[5]) + (v[6] * v[6] + v[7] * v[7]);
        }
        sq = wave_sum(sq);
        if (lane == 0) sqs[wave] = sq;
    }
    if (active) {
        if (NORM) v = g0 * v;
        bf16x8 hi, lo;
        split8(v, hi, lo);
        if (IW == 8) {
            bf16x8* dst = reinterpret_cast<bf16x8*>(stage + (size_t)(tid >> 2) * 128) + (tid & 3);
            dst[0] = hi;
            dst[4] = lo;
        } else {                                                   // the lower four of the pair: half of a 16-byte k-group
            typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
            bf16x4* dst = reinterpret_cast<bf16x4*>(stage + (size_t)(tid >> 3) * 128) + (tid & 7);
            dst[0] = __builtin_shufflevector(hi, hi, 0, 1, 2, 3);
            dst[8] = __builtin_shufflevector(lo, lo, 0, 1, 2, 3);
        }
    }
    __syncthreads();
    R1_T(4);
    float rs = 1.f;
    if (NORM) rs = rsqrtf((((sqs[0] + sqs[1]) + (sqs[2] + sqs[3])) + ((sqs[4] + sqs[5]) + (sqs[6] + sqs[7]))) / (float)K + eps);    // waves without items contributed 0
    // The B operands of a group of k-steps are requested together, THEN the group's MFMAs run: left to itself the compiler reads
    // each step's operand right in front of its MFMA pair (ds_read -> wait -> 2 MFMAs, ~85 ns per k-step in the ISA), a serial
    // chain of LDS latencies.  A k-step beyond the wave's share multiplies a zero fragment (no branches inside a group).
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const s16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    constexpr int GRP = MAXKS <= 7 ? MAXKS : (MAXKS + 1) / 2;
#pragma unroll
    for (int g0 = 0; g0 < MAXKS; g0 += GRP) {
        bf16x8 bh[GRP], bl[GRP];
#pragma unroll
        for (int j = 0; j < GRP; j++) {
            const int i = g0 + j;
            if (i < MAXKS) {
                const int s = (w0 + i < w1 ? w0 + i : w1 - 1) - ks0;
                const bf16x8* xb = reinterpret_cast<const bf16x8*>(stage + (size_t)s * 128) + (lane >> 4);
                bh[j] = xb[0];
                bl[j] = xb[4];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < GRP; j++) {
            const int i = g0 + j;
            if (i < MAXKS) {
                const bf16x8 a = __builtin_bit_cast(bf16x8, w0 + i < w1 ? abuf[i] : zero8);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bh[j], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bl[j], acc, 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    R1_T(5);
    // every column of the tile holds the same vector: the lanes of column 0 carry it out
    if ((lane & 15) == 0) *reinterpret_cast<f32x4*>(red + ((wk * NWR + wr) * 4 + (lane >> 4)) * 4) = acc;
    __syncthreads();
    R1_T(6);
    float out = 0.f;
    if (tid < NWR * 16) {
        const int wr_ = tid >> 4, q = (tid >> 2) & 3, r = tid & 3;
        out = red[((0 * NWR + wr_) * 4 + q) * 4 + r];
#pragma unroll
        for (int j = 1; j < NWK; j++) out += red[((j * NWR + wr_) * 4 + q) * 4 + r];
        out *= rs;
    }
    return out;
}


// ---- two rows per block (k_step2): the rows are columns 0 and 1 of the MFMA's B operand, so one pass over the block's weight fragments
// and one set of MFMAs serve both.  SPLIT (operands of <= 256 items: Q, O, gate/up, head): the two halves of the block fetch and stage
// the two rows' operands at the same time (op0 by threads 0..255, op1 by 256..511, wave index relative to the half); otherwise (down
// projection, 304 items) all threads fetch row 0, then row 1.  Returns feature f of row c in thread c * NWR * 16 + f.  Per row the
// arithmetic (order of every sum) is row1_core's.
#define R2_STAGE_BYTES(nks) ((nks) * 256)
__device__ __host__ constexpr int r2_smem_bytes(int nks) { return R2_STAGE_BYTES(nks) + R1_XCH_BYTES + 32 + 2 * 8 * 4 * 16; }
template <int NWR, int NWK, int MAXKS, bool NORM, bool SPLIT, bool NT, class OP, class HOOK = R1NoHook>
__device__ __forceinline__ float row2_core(const uint16_t* __restrict__ W, int tile0, int tstride, int KS, int K, int ks0, int ks1, OP& op0, OP& op1,
                                           const float* norm_w, float eps, char* smem, HOOK issued = HOOK()) {
    static_assert(NWR * NWK == 8, "row2_core: 512 threads");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave % NWR, wk = wave / NWR;
    const int nks = ks1 - ks0;
    const int w0 = ks0 + (nks * wk) / NWK, w1 = ks0 + (nks * (wk + 1)) / NWK;
    constexpr int IW = OP::IW;
    const int nitems = nks * (32 / IW);                           // per row
    const int half = SPLIT ? (wave >> 2) : 0;                     // (wave-uniform)
    const int tl = SPLIT ? (tid & 255) : tid, hw = SPLIT ? (wave & 3) : wave;
    const bool active = tl < nitems;
    const int k = ks0 * 32 + tl * IW;
    char* stage = smem;
    char* xch = smem + R2_STAGE_BYTES(nks);
    float* sqs = reinterpret_cast<float*>(xch + R1_XCH_BYTES);     // [8] per-wave sums of squares (SPLIT: waves 0..3 row 0, 4..7 row 1)
    float* red = reinterpret_cast<float*>(xch + R1_XCH_BYTES + 32);   // [2 rows][NWK][NWR][4 quarters][4]
    f32x8 g0 = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
    if (NORM && active) {
        if (IW == 8) g0 = *reinterpret_cast<const f32x8*>(norm_w + k);
        else { const f32x4 g4 = *reinterpret_cast<const f32x4*>(norm_w + k); g0[0] = g4[0]; g0[1] = g4[1]; g0[2] = g4[2]; g0[3] = g4[3]; }
    }
    const char* wbase = reinterpret_cast<const char*>(W) + ((size_t)(tile0 + wr * tstride) * KS + w0) * 1024;
    const unsigned wlane = lane * 16;
    s16x8 abuf[MAXKS];
    const int nw = w1 - w0;
#pragma unroll
    for (int i = 0; i < MAXKS; i++) {
        const s16x8* wp_ = reinterpret_cast<const s16x8*>(wbase + (size_t)(i < nw ? i : (nw > 0 ? nw - 1 : 0)) * 1024 + wlane);
        abuf[i] = ld_wfrag<NT>(wp_);       // several pairs: the sibling chains' blocks of this tile follow on this XCD -> keep it in L2
    }
    __builtin_amdgcn_sched_barrier(0);
    issued();
    if (tid < 2) reinterpret_cast<volatile int*>(xch)[tid] = 0;     // the providers' arming words (OpGran::nobar)
    __syncthreads();
    auto put = [&](f32x8 v, int col) {                            // the row's hi / lo planes into column `col` of the operand stage
        if (!active) return;
        if (NORM) v = g0 * v;
        bf16x8 hi, lo;
        split8(v, hi, lo);
        if (IW == 8) {
            bf16x8* dst = reinterpret_cast<bf16x8*>(stage + (size_t)(tl >> 2) * 256 + col * 64) + (tl & 3);
            dst[0] = hi;
            dst[8] = lo;
        } else {
            typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
            bf16x4* dst = reinterpret_cast<bf16x4*>(stage + (size_t)(tl >> 3) * 256 + col * 64) + (tl & 7);
            dst[0] = __builtin_shufflevector(hi, hi, 0, 1, 2, 3);
            dst[16] = __builtin_shufflevector(lo, lo, 0, 1, 2, 3);
        }
    };
    auto sumsq = [&](const f32x8& v) {
        float sq = 0.f;
        if (active) {
            sq = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
            if (IW == 8) sq += (v[4] * v[4] + v[5] * v[5]) + (v[6] * v[6] + v[7] * v[7]);
        }
        return wave_sum(sq);
    };
    if constexpr (SPLIT) {
        f32x8 v = half == 0 ? op0.finish(k, hw, nitems, active, xch) : op1.finish(k, hw, nitems, active, xch);
        if (NORM) { const float sq = sumsq(v); if (lane == 0) sqs[wave] = sq; }
        put(v, half);
    } else {
        static_assert(SPLIT || !NORM, "row2_core: the serial form carries no RMSNorm");
        // both rows' vectors were published by the same producer blocks at the same time: one attempt with both rows' loads in flight
        // together (one round trip); a wave whose attempt fails fetches them one after the other the blocking way
        f32x8 v0 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, v1 = v0;
        bool both = false;
        if (IW == 8 && op0.G->spec) {
            bool ok = true;
            if (active) { ok = op0.G->ld8(op0.g0 + k, v0); ok &= op1.G->ld8(op1.g0 + k, v1); }
            both = __all(ok) && hw * 64 < nitems;
        }
        if (both) {                       // (the leader wave arms both words for the waves that go the blocking way)
            if (hw == 0 && lane == 0) { volatile int* aw = reinterpret_cast<volatile int*>(xch); aw[op0.grp] = 1; aw[op1.grp] = 1; }
        } else { v0 = op0.finish(k, hw, nitems, active, xch); v1 = op1.finish(k, hw, nitems, active, xch); }
        put(v0, 0);
        put(v1, 1);
    }
    __syncthreads();
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const s16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    constexpr int GRP = MAXKS <= 7 ? MAXKS : (MAXKS + 1) / 2;
    const int bcol = (lane & 15) == 1 ? 64 : 0;                   // B operand: column n = lane & 15 -> row n (columns >= 2 read row 0, results unused)
#pragma unroll
    for (int gg = 0; gg < MAXKS; gg += GRP) {
        bf16x8 bh[GRP], bl[GRP];
#pragma unroll
        for (int j = 0; j < GRP; j++) {
            const int i = gg + j;
            if (i < MAXKS) {
                const int s_ = (w0 + i < w1 ? w0 + i : w1 - 1) - ks0;
                const bf16x8* xb = reinterpret_cast<const bf16x8*>(stage + (size_t)s_ * 256 + bcol) + (lane >> 4);
                bh[j] = xb[0];
                bl[j] = xb[8];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < GRP; j++) {
            const int i = gg + j;
            if (i < MAXKS) {
                const bf16x8 a_ = __builtin_bit_cast(bf16x8, w0 + i < w1 ? abuf[i] : zero8);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_, bh[j], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_, bl[j], acc, 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if ((lane & 15) < 2) *reinterpret_cast<f32x4*>(red + (lane & 15) * (NWK * NWR * 16) + ((wk * NWR + wr) * 4 + (lane >> 4)) * 4) = acc;
    __syncthreads();
    float out = 0.f;
    if (tid < 2 * NWR * 16) {
        const int c = tid / (NWR * 16), t = tid - c * (NWR * 16);
        const int wr_ = t >> 4, q = (t >> 2) & 3, r_ = t & 3;
        const float* rc = red + c * (NWK * NWR * 16);
        out = rc[((0 * NWR + wr_) * 4 + q) * 4 + r_];
#pragma unroll
        for (int j = 1; j < NWK; j++) out += rc[((j * NWR + wr_) * 4 + q) * 4 + r_];
        if (NORM) out *= rsqrtf(((sqs[4 * c] + sqs[4 * c + 1]) + (sqs[4 * c + 2] + sqs[4 * c + 3])) / (float)K + eps);
    }
    return out;
}

// ---- four rows per block (k_step4): columns 0 .. 3 of the MFMA's B operand.  SPLIT (operands of <= 256 items): half h of the block
// serves rows 2 h and 2 h + 1 -- one attempt with both rows' loads in flight (the rows' producers publish together), the blocking
// providers one after the other where that fails; otherwise (down projection) all threads take rows (0, 1), then (2, 3) the same way.
// ops[r].grp must be r (own arming word per row, cleared here), OpGran::nobar set.  Returns feature f of row c in thread c * NWR * 16 + f.
#define R4_STAGE_BYTES(nks) ((nks) * 512)
__device__ __host__ constexpr int r4_smem_bytes(int nks) { return R4_STAGE_BYTES(nks) + R1_XCH_BYTES + 64 + 4 * 8 * 4 * 16; }
template <int NWR, int NWK, int MAXKS, bool NORM, bool SPLIT, bool NT, class OP, class HOOK = R1NoHook>
__device__ __forceinline__ float row4_core(const uint16_t* __restrict__ W, int tile0, int tstride, int KS, int K, int ks0, int ks1, OP (&ops)[4],
                                           const float* norm_w, float eps, char* smem, HOOK issued = HOOK()) {
    static_assert(NWR * NWK == 8, "row4_core: 512 threads");
    static_assert(SPLIT || !NORM, "row4_core: the serial form carries no RMSNorm");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave % NWR, wk = wave / NWR;
    const int nks = ks1 - ks0;
    const int w0 = ks0 + (nks * wk) / NWK, w1 = ks0 + (nks * (wk + 1)) / NWK;
    constexpr int IW = OP::IW;
    const int nitems = nks * (32 / IW);
    const int half = SPLIT ? (wave >> 2) : 0;
    const int tl = SPLIT ? (tid & 255) : tid, hw = SPLIT ? (wave & 3) : wave;
    const bool active = tl < nitems;
    const int k = ks0 * 32 + tl * IW;
    char* stage = smem;
    char* xch = smem + R4_STAGE_BYTES(nks);
    float* sqs = reinterpret_cast<float*>(xch + R1_XCH_BYTES);     // [4 rows][4 waves of the row's half]
    float* red = reinterpret_cast<float*>(xch + R1_XCH_BYTES + 64);   // [4 rows][NWK][NWR][4 quarters][4]
    f32x8 g0 = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
    if (NORM && active) {
        if (IW == 8) g0 = *reinterpret_cast<const f32x8*>(norm_w + k);
        else { const f32x4 g4 = *reinterpret_cast<const f32x4*>(norm_w + k); g0[0] = g4[0]; g0[1] = g4[1]; g0[2] = g4[2]; g0[3] = g4[3]; }
    }
    const char* wbase = reinterpret_cast<const char*>(W) + ((size_t)(tile0 + wr * tstride) * KS + w0) * 1024;
    const unsigned wlane = lane * 16;
    s16x8 abuf[MAXKS];
    const int nw = w1 - w0;
#pragma unroll
    for (int i = 0; i < MAXKS; i++) {
        const s16x8* wp_ = reinterpret_cast<const s16x8*>(wbase + (size_t)(i < nw ? i : (nw > 0 ? nw - 1 : 0)) * 1024 + wlane);
        abuf[i] = ld_wfrag<NT>(wp_);
    }
    __builtin_amdgcn_sched_barrier(0);
    issued();
    volatile int* aw = reinterpret_cast<volatile int*>(xch);
    if (tid < 4) aw[tid] = 0;
    __syncthreads();
    auto put = [&](f32x8 v, int col) {
        if (!active) return;
        if (NORM) v = g0 * v;
        bf16x8 hi, lo;
        split8(v, hi, lo);
        if (IW == 8) {
            bf16x8* dst = reinterpret_cast<bf16x8*>(stage + (size_t)(tl >> 2) * 512 + col * 64) + (tl & 3);
            dst[0] = hi;
            dst[16] = lo;
        } else {
            typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
            bf16x4* dst = reinterpret_cast<bf16x4*>(stage + (size_t)(tl >> 3) * 512 + col * 64) + (tl & 7);
            dst[0] = __builtin_shufflevector(hi, hi, 0, 1, 2, 3);
            dst[32] = __builtin_shufflevector(lo, lo, 0, 1, 2, 3);
        }
    };
    auto sumsq = [&](const f32x8& v) {
        float sq = 0.f;
        if (active) {
            sq = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
            if (IW == 8) sq += (v[4] * v[4] + v[5] * v[5]) + (v[6] * v[6] + v[7] * v[7]);
        }
        return wave_sum(sq);
    };
    auto pair = [&](OP& oa, OP& ob, int ra, int rb) {             // rows ra, rb of this thread group: fetch, (sum of squares), stage
        f32x8 va, vb;                                             // (the providers by reference: a run-time index into ops[] puts the array in scratch)
        bool both = false;
        if (oa.G->spec) {
            bool ok = oa.attempt(k, active, va);
            ok &= ob.attempt(k, active, vb);
            both = __all(ok) && hw * 64 < nitems;
        }
        if (both) { if (hw == 0 && lane == 0) { aw[ra] = 1; aw[rb] = 1; } }      // (the leader wave arms the rows' words for waves that go the blocking way)
        else { va = oa.finish(k, hw, nitems, active, xch); vb = ob.finish(k, hw, nitems, active, xch); }
        if (NORM) {
            const float sa = sumsq(va), sb = sumsq(vb);
            if (lane == 0) { sqs[ra * 4 + hw] = sa; sqs[rb * 4 + hw] = sb; }
        }
        put(va, ra);
        put(vb, rb);
    };
    if constexpr (SPLIT) { if (half == 0) pair(ops[0], ops[1], 0, 1); else pair(ops[2], ops[3], 2, 3); }
    else { pair(ops[0], ops[1], 0, 1); pair(ops[2], ops[3], 2, 3); }
    __syncthreads();
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const s16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    constexpr int GRP = MAXKS <= 7 ? MAXKS : (MAXKS + 1) / 2;
    const int bcol = ((lane & 15) < 4 ? (lane & 15) : 0) * 64;       // B operand: column n = lane & 15 -> row n (columns >= 4 read row 0, results unused)
#pragma unroll
    for (int gg = 0; gg < MAXKS; gg += GRP) {
        bf16x8 bh[GRP], bl[GRP];
#pragma unroll
        for (int j = 0; j < GRP; j++) {
            const int i = gg + j;
            if (i < MAXKS) {
                const int s_ = (w0 + i < w1 ? w0 + i : w1 - 1) - ks0;
                const bf16x8* xb = reinterpret_cast<const bf16x8*>(stage + (size_t)s_ * 512 + bcol) + (lane >> 4);
                bh[j] = xb[0];
                bl[j] = xb[16];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < GRP; j++) {
            const int i = gg + j;
            if (i < MAXKS) {
                const bf16x8 a_ = __builtin_bit_cast(bf16x8, w0 + i < w1 ? abuf[i] : zero8);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_, bh[j], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_, bl[j], acc, 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if ((lane & 15) < 4) *reinterpret_cast<f32x4*>(red + (lane & 15) * (NWK * NWR * 16) + ((wk * NWR + wr) * 4 + (lane >> 4)) * 4) = acc;
    __syncthreads();
    float out = 0.f;
    if (tid < 4 * NWR * 16) {
        const int c = tid / (NWR * 16), t = tid - c * (NWR * 16);
        const int wr_ = t >> 4, q = (t >> 2) & 3, r_ = t & 3;
        const float* rc = red + c * (NWK * NWR * 16);
        out = rc[((0 * NWR + wr_) * 4 + q) * 4 + r_];
#pragma unroll
        for (int j = 1; j < NWK; j++) out += rc[((j * NWR + wr_) * 4 + q) * 4 + r_];
        if (NORM) out *= rsqrtf(((sqs[4 * c] + sqs[4 * c + 1]) + (sqs[4 * c + 2] + sqs[4 * c + 3])) / (float)K + eps);
    }
    return out;
}

// ---- two row tiles per wave (k_step1: a QA block's query head, a gate/up block of two pairs): the eight waves are row1_core<2, 4, MAXKS>'s
// 2 (wr) x 4 (wk) grid, wave (wr, wk) owns the row tiles tile0 + wr and tile0 + wr + 2 over its K quarter -- a block covers four consecutive
// row tiles, feature i of tile q = wr + 2 j comes back in thread 16 q + i.  Per feature every sum runs in row1_core<2, 4, MAXKS>'s order
// (the same fragments in the same k order per wave, the same fold of the four K quarters, the same sum of squares), so the values do not
// depend on which of the two forms computed them.  Split in two so that a caller can put loads of its own (the QA role's cache rows)
// between the weight requests and the operand wait.
#define T2_RED_BYTES (4 * 4 * 4 * 4 * 4)                      // [4 K quarters][4 tiles][4 lane quarters][4] floats
__device__ __host__ constexpr int t2_smem_bytes(int nks) { return R1_STAGE_BYTES(nks) + R1_XCH_BYTES + 32 + T2_RED_BYTES; }
template <int MAXKS>
struct T2W { s16x8 a0[MAXKS], a1[MAXKS]; };
template <int MAXKS, bool NT>
__device__ __forceinline__ void t2_issue(const uint16_t* __restrict__ W, int tile0, int KS, T2W<MAXKS>& w) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave & 1, wk = wave >> 1;
    const int w0 = (KS * wk) / 4, w1 = (KS * (wk + 1)) / 4, nw = w1 - w0;
    const char* b0 = reinterpret_cast<const char*>(W) + ((size_t)(tile0 + wr) * KS + w0) * 1024 + lane * 16;
    const char* b1 = reinterpret_cast<const char*>(W) + ((size_t)(tile0 + wr + 2) * KS + w0) * 1024 + lane * 16;
#pragma unroll
    for (int i = 0; i < MAXKS; i++) {
        const size_t o = (size_t)(i < nw ? i : (nw > 0 ? nw - 1 : 0)) * 1024;
        const s16x8* p0 = reinterpret_cast<const s16x8*>(b0 + o);
        const s16x8* p1 = reinterpret_cast<const s16x8*>(b1 + o);
        w.a0[i] = ld_wfrag<NT>(p0);
        w.a1[i] = ld_wfrag<NT>(p1);
    }
    __builtin_amdgcn_sched_barrier(0);
}
template <int MAXKS, bool NORM, class OP>
__device__ __forceinline__ float t2_finish(const T2W<MAXKS>& w, int KS, int K, OP& op, const float* norm_w, float eps, char* smem) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave & 1, wk = wave >> 1;
    const int w0 = (KS * wk) / 4, w1 = (KS * (wk + 1)) / 4;
    constexpr int IW = OP::IW;
    const int nitems = KS * (32 / IW);
    const bool active = tid < nitems;
    const int k = tid * IW;
    char* stage = smem;
    char* xch = smem + R1_STAGE_BYTES(KS);
    float* sqs = reinterpret_cast<float*>(xch + R1_XCH_BYTES);
    float* red = reinterpret_cast<float*>(xch + R1_XCH_BYTES + 32);
    f32x8 g0 = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
    if (NORM && active) {
        if (IW == 8) g0 = *reinterpret_cast<const f32x8*>(norm_w + k);
        else { const f32x4 g4 = *reinterpret_cast<const f32x4*>(norm_w + k); g0[0] = g4[0]; g0[1] = g4[1]; g0[2] = g4[2]; g0[3] = g4[3]; }
    }
    f32x8 v = op.finish(k, wave, nitems, active, xch);
    R1_T_OPERAND;
    if (NORM) {
        float sq = 0.f;
        if (active) {
            sq = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
            if (IW == 8) sq += (v[4] * v[4] + v[5] * v[5]) + (v[6] * v[6] + v[7] * v[7]);
        }
        sq = wave_sum(sq);
        if (lane == 0) sqs[wave] = sq;
    }
    if (active) {
        if (NORM) v = g0 * v;
        bf16x8 hi, lo;
        split8(v, hi, lo);
        if (IW == 8) {
            bf16x8* dst = reinterpret_cast<bf16x8*>(stage + (size_t)(tid >> 2) * 128) + (tid & 3);
            dst[0] = hi;
            dst[4] = lo;
        } else {
            typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
            bf16x4* dst = reinterpret_cast<bf16x4*>(stage + (size_t)(tid >> 3) * 128) + (tid & 7);
            dst[0] = __builtin_shufflevector(hi, hi, 0, 1, 2, 3);
            dst[8] = __builtin_shufflevector(lo, lo, 0, 1, 2, 3);
        }
    }
    __syncthreads();
    R1_T(4);
    float rs = 1.f;
    if (NORM) rs = rsqrtf((((sqs[0] + sqs[1]) + (sqs[2] + sqs[3])) + ((sqs[4] + sqs[5]) + (sqs[6] + sqs[7]))) / (float)K + eps);
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    const s16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    // B operands in groups of GRP k-steps (row1_core's reason: requested together, then the group's MFMAs; all MAXKS at once would hold
    // 8 MAXKS registers beside the 8 MAXKS of weight fragments)
    constexpr int GRP = 4;
#pragma unroll
    for (int g = 0; g < MAXKS; g += GRP) {
        bf16x8 bh[GRP], bl[GRP];
#pragma unroll
        for (int j = 0; j < GRP; j++) {
            const int i = g + j;
            if (i < MAXKS) {
                const int s = (w0 + i < w1 ? w0 + i : w1 - 1);
                const bf16x8* xb = reinterpret_cast<const bf16x8*>(stage + (size_t)s * 128) + (lane >> 4);
                bh[j] = xb[0];
                bl[j] = xb[4];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < GRP; j++) {                              // the two tiles' chains interleaved: independent accumulators
            const int i = g + j;
            if (i < MAXKS) {
                const bf16x8 a0 = __builtin_bit_cast(bf16x8, w0 + i < w1 ? w.a0[i] : zero8);
                const bf16x8 a1 = __builtin_bit_cast(bf16x8, w0 + i < w1 ? w.a1[i] : zero8);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, bh[j], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bh[j], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, bl[j], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bl[j], acc1, 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    R1_T(5);
    if ((lane & 15) == 0) {
        *reinterpret_cast<f32x4*>(red + ((wk * 4 + wr) * 4 + (lane >> 4)) * 4) = acc0;
        *reinterpret_cast<f32x4*>(red + ((wk * 4 + wr + 2) * 4 + (lane >> 4)) * 4) = acc1;
    }
    __syncthreads();
    R1_T(6);
    float out = 0.f;
    if (tid < 64) {
        const int q = tid >> 4, lq = (tid >> 2) & 3, r = tid & 3;
        out = red[((0 * 4 + q) * 4 + lq) * 4 + r];
#pragma unroll
        for (int j = 1; j < 4; j++) out += red[((j * 4 + q) * 4 + lq) * 4 + r];
        out *= rs;
    }
    return out;
}
