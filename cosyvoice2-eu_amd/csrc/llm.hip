// Stage 1 on MI355X: Qwen2-0.5B speech-token LM decode (replaces cosyvoice/llm/llm.py:575-719 and the HF
// Qwen2ForCausalLM forward it calls at llm.py:107-117).
//
// One decode step = 5 kernels per layer + head + sampler, captured in a hipGraph per batch size:
//   k_qkv    : [sum partials -> x] RMSNorm -> QKV skinny GEMM -> +bias -> RoPE -> q buffer / KV cache write
//   k_attn   : GQA decode attention over the fp32 KV cache
//   k_oproj  : O projection (output kept separate; the residual add happens in k_gateup's prologue)
//   k_gateup : [x + o -> x_mid] RMSNorm -> gate/up skinny GEMM -> SiLU(g)*u
//   k_down   : down projection, split-K partials (summed by the next k_qkv / k_head prologue)
//   k_head   : final RMSNorm -> llm_decoder GEMV + bias -> logits
//   k_sample : log-softmax, EOS rules, greedy or RAS draw, token bookkeeping, next input embedding gather
// All are HBM-bound weight streams (skinny.h); the per-step algorithmic traffic is the bf16 weights once
// (727.6 MB at the real dims) plus the KV read.
#include "skinny.h"
#ifdef CV2_STAMPS
extern __device__ unsigned long long g_chain_t[1024][8];
#define R1_T(i) do { if (op.dbg >= 0 && threadIdx.x == 0) g_chain_t[op.dbg][i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define R1_T_OPERAND R1_T(3)
#endif
#include "chain.h"
#include "skinny_launch.h"
#include "gemm_launch.h"
#include "../../include/cv2_amd.h"
#include <algorithm>
#include <map>
#include <stdlib.h>
#include <stdarg.h>
#include <vector>

thread_local std::string g_cv2_err;
int cv2_fail(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_cv2_err = buf;
    return -1;
}
extern "C" const char* cv2_last_error(void) { return g_cv2_err.c_str(); }
extern "C" int cv2_version(void) { return CV2_ABI_VERSION; }

#define ST CV2_LLM_STATE_STRIDE

struct RowMap {          // row r of a launch -> (sequence slot, position)
    const int* state;    // decode: seq = slots ? slots[r] : r, pos = state[seq][POS]
    int prefill;         // prefill: seq = seq0, pos = pos0 + r
    int seq0, pos0;
    const int* slots;    // decode over the live slots only (cv2_llm_decode_rows): row -> slot, null = identity
    __device__ __forceinline__ void get(int r, int& seq, int& pos) const {
        if (prefill) { seq = seq0; pos = pos0 + r; }
        else { seq = slots ? slots[r] : r; pos = state[seq * ST + CV2_ST_POS]; }
    }
};

// ------------------------------------------------------------------ k_qkv
struct QkvArgs {
    const uint16_t* W; const float* bias;
    SkinnyX X; int KS, rows, K;
    int n_q, n_kv;
    const float* cosT; const float* sinT;       // [max_pos][32]
    float* q;                                   // [rows][n_q*64]
    float* kc; float* vc;                       // this layer's cache: [max_seqs][n_kv][max_pos][64]
    int max_pos;
    RowMap rm;
    const float* sq; int nsq; float eps;        // PRE operand left un-normalised by k_store<.., LAST>: rstd[r] = rsqrt(sum_b sq[r][b] / K + eps), as k_gateup
};
// block = (head, half): features {half*16 + i, half*16 + 32 + i : i < 16} so that the rotate-half partner is in-block
template <int NB, bool PRE = false>
__global__ __launch_bounds__(512) void k_qkv(QkvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int head = blockIdx.x >> 1, half = blockIdx.x & 1;   // heads: n_q query, then n_kv key, then n_kv value
    const int wr = (threadIdx.x >> 6) % 2;
    // epilogue operands of this thread's first element.  The position is requested FIRST (it returns first), the bias with it;
    // the RoPE table entries depend on the position and are requested from the hook, i.e. after the operand and weight loads
    // have been issued: the chain state -> pos -> cos/sin then runs under the weight stream and delays nothing.
    const bool rot = head < a.n_q + a.n_kv;
    const bool own = threadIdx.x < a.rows * 32;
    const int f0 = half * 16 + ((threadIdx.x >> 4) & 1) * 32 + (threadIdx.x & 15);
    int seq0 = 0, pos0 = 0;
    float pb0, pb1, pc0, ps0;
    // all of these are issued by every thread (row clamped), branch-free: a conditional load would make the number of loads in
    // flight unknown at compile time and turn the counted waits on the operand into waits on the weight stream
    a.rm.get(min((int)threadIdx.x >> 5, a.rows - 1), seq0, pos0);
    pb0 = a.bias[head * 64 + f0];
    pb1 = a.bias[head * 64 + (f0 ^ 32)];
    __shared__ float rs_s[SK_ROWS_CAP];
    float sqt = 0.f;                                    // (PRE with sq) thread (row, i of 16): its stride of the row's shares, fixed order
    auto hook = [&]() {
        pc0 = a.cosT[pos0 * 32 + (f0 & 31)];
        ps0 = a.sinT[pos0 * 32 + (f0 & 31)];
        if (PRE && a.sq) {
            const int rr = min((int)threadIdx.x >> 4, a.rows - 1);
            for (int b = threadIdx.x & 15; b < a.nsq; b += 16) sqt += a.sq[(size_t)rr * a.nsq + b];
        }
    };
    float* res = skinny_core<NB, 2, 4, 8, false, PRE>(a.W, head * 4 + half + 2 * wr, a.KS, a.rows, a.K, a.X, smem, hook);
    const int ld = NB * 16 + 1;
    const bool scaled = PRE && a.sq;
    if (scaled) {
        sqt += dpp_mov_f32<0xB1, 0xf>(0.f, sqt);
        sqt += dpp_mov_f32<0x4E, 0xf>(0.f, sqt);
        sqt += __shfl_xor(sqt, 4);
        sqt += __shfl_xor(sqt, 8);
        if ((threadIdx.x & 15) == 0 && (int)threadIdx.x < a.rows * 16) rs_s[threadIdx.x >> 4] = rsqrtf(sqt / (float)a.K + a.eps);
        __syncthreads();
    }
    for (int e = threadIdx.x; e < a.rows * 32; e += blockDim.x) {
        const int r = e >> 5, w = (e >> 4) & 1, i16 = e & 15;
        const int f = half * 16 + w * 32 + i16;           // feature within the head
        int seq = seq0, pos = pos0;
        float b0 = pb0, b1 = pb1, c = pc0, sn = ps0;
        if (e != (int)threadIdx.x) {
            a.rm.get(r, seq, pos);
            b0 = a.bias[head * 64 + f];
            if (rot) { b1 = a.bias[head * 64 + (f ^ 32)]; c = a.cosT[pos * 32 + (f & 31)]; sn = a.sinT[pos * 32 + (f & 31)]; }
        }
        const float rs = scaled ? rs_s[r] : 1.f;
        // (the scaled value is rounded BEFORE the bias is added, as in the <= 16-row form where skinny_core applies the row's rstd: hipcc would
        // contract the two into one fma)
        float t0 = res[(w * 16 + i16) * ld + r] * rs, t1 = res[((1 - w) * 16 + i16) * ld + r] * rs;
        asm volatile("" : "+v"(t0), "+v"(t1));
        float v = t0 + b0;
        if (rot) {                                         // rotate-half RoPE on q and k heads
            const float vp = t1 + b1;
            v = (f < 32) ? (v * c - vp * sn) : (v * c + vp * sn);
        }
        if (head < a.n_q) a.q[(size_t)r * a.n_q * 64 + head * 64 + f] = v;
        else if (head < a.n_q + a.n_kv)
            a.kc[(((size_t)seq * a.n_kv + (head - a.n_q)) * a.max_pos + pos) * 64 + f] = v;
        else
            a.vc[(((size_t)seq * a.n_kv + (head - a.n_q - a.n_kv)) * a.max_pos + pos) * 64 + f] = v;
    }
}

// ------------------------------------------------------------------ k_attn (decode / prefill rows), split over keys
// Flash-decoding: block = (kv head, key split, row).  The block stages its K and V key range (fp32) in LDS once and
// serves every query head of the GQA group from it, then leaves UNNORMALISED partial outputs and (max, sum) per head;
// the O-projection kernel combines the splits while it loads its operand (skinny.h, att_ns), so attention costs one
// launch and no extra pass.  Grid is fixed (graph capture): splits beyond the current length flag themselves empty.
struct AttnArgs {
    const float* q; const float* kc; const float* vc;
    float* part_o;            // [nsplit][SK_ROWS_CAP][n_q*64]
    float* part_ml;           // [nsplit][SK_ROWS_CAP][n_q][2]
    int* part_cnt;            // [SK_ROWS_CAP] number of non-empty splits of the row
    int n_q, n_kv, max_pos, nsplit, keys_per_split;
    RowMap rm;
    int rep;                  // n_q / n_kv (<= 8), passed so that no block divides
    // many-row launches (17..32 rows): the LAST split of a (row, kv head) to finish combines the splits and leaves the O projection's
    // operand planes (what k_prep<true> did as a launch of its own).  arrive = [SK_ROWS_CAP][n_kv] counters, zero between launches
    int* arrive; uint16_t* pre;
};
// Stores / loads that are coherent across the XCDs' L2s (sc1): what a last-arriving block reads of the other blocks' results
__device__ __forceinline__ void st_agent(float* p, float v) {
    __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_agent(const float* p) {
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
// Block-uniform: has every one of the `n` blocks that share `cnt` (this one included) left its results?  The callers' result stores are
// st_agent; they are complete (vmcnt) before the barrier, the counter is bumped after it.  The last block re-arms the counter.
__device__ __forceinline__ bool last_arriver(int* cnt, int n) {
    __shared__ int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const int old = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = old == n - 1;
        if (old == n - 1) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    return s_last != 0;
}
// element (row r, column c) of an operand in the layout the PRE kernels copy (skinny.h): [c / 32][r / 16][hi, lo][(c / 8) % 4][r % 16][c % 8]
__device__ __forceinline__ void put_planes(uint16_t* pre, int r, int c, float v) {
    uint16_t* d = pre + ((size_t)((c >> 5) * 2 + (r >> 4)) * 2) * 512 + (((c >> 3) & 3) * 16 + (r & 15)) * 8 + (c & 7);
    const uint16_t hb = f2bf(v);
    d[0] = hb;
    d[512] = f2bf(v - bf2f(hb));
}
#define AT_KB 64
// No LDS staging of K or V: thread (key, quarter) keeps its 16 dims of one key row in registers for the scores of all
// heads of the group, thread (dim, head pair) keeps the V column of the tile in registers; one round of global loads.
// A split of keys_per_split = 64 NSUB keys is handled by NSUB groups of 256 threads, one 64-key tile each, ALL of whose loads
// are requested at kernel entry: phase stamps showed every serial tile costing a full memory round trip (~1.5 us), so a
// 256-key split walked tile by tile took four of them.  The groups' (max, sum, output) triples are merged through LDS.
template <int NSUB>
__global__ __launch_bounds__(256 * NSUB) void k_attn(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) float qs[8 * 64];
    __shared__ __attribute__((aligned(16))) float ps[NSUB][8 * AT_KB];
    __shared__ __attribute__((aligned(16))) float po_s[NSUB][8 * 8 * 64];      // PV partials [group][key eighth][head][dim]
    __shared__ float run_m[NSUB][8], run_l[NSUB][8], tile_scale[NSUB][8];
    const int rep = a.rep;
    const int sp = blockIdx.x, g = blockIdx.y, r = blockIdx.z;    // grid = (key split, kv head, row)
    const int tid = threadIdx.x, lane = tid & 63;
    const int sub = __builtin_amdgcn_readfirstlane(tid >> 8);     // 64-key tile group of this thread
    const int t = tid & 255, w = t >> 6;
    // PV: thread = (4 dims d4, key eighth kq of 8 keys, head set hs of 4 heads): V rows are read as float4 (8 loads per thread,
    // 32 registers: with 16 waves per block the budget is 128)
    const int d4 = t & 15, kq = (t >> 4) & 7, hs = t >> 7;
    const int key_t = t >> 2, qd = t & 3;                         // scores: thread = (key, 16-dim quarter)
    const int split_lo = sp * a.keys_per_split;
    SK_STAMP_DECL;
    SK_STAMP(0);
    // Request order = arrival order: q (written by the previous kernel, needed first), the position, then K, then V.
    // The slot is known without touching memory (decode: slot = row), so K / V are requested BEFORE the position is known;
    // rows beyond the current length exist (the cache is allocated up to max_pos) and are masked once it has arrived.
    float qreg[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int e = tid + i * 256 * NSUB;
        qreg[i] = a.q[(size_t)r * a.n_q * 64 + g * rep * 64 + min(e, rep * 64 - 1)];
    }
    int seq, pos;
    a.rm.get(r, seq, pos);
    // Many rows per launch (batched decode: hundreds of blocks, throughput-bound): wait for the position and skip the splits past
    // the sequence's length BEFORE fetching their tiles - speculative tiles of empty splits were twice the live KV traffic at 32
    // rows.  Few rows (latency-bound): fetch first, mask later.
    if (gridDim.z > 4 && split_lo >= pos + 1) return;
    const int seq_s = a.rm.prefill ? a.rm.seq0 : (a.rm.slots ? seq : r);     // (live-row decode: one scalar load ahead of K / V)
    const float* K = a.kc + ((size_t)seq_s * a.n_kv + g) * a.max_pos * 64;
    const float* V = a.vc + ((size_t)seq_s * a.n_kv + g) * a.max_pos * 64;
    f32x4 kk[4];
    f32x4 vv[8];
    {
        const int j0 = split_lo + sub * AT_KB;
        // no clamping (one offset register + immediates): rows past max_pos of the last head run into the slack carve() leaves
        const unsigned ko = (unsigned)(j0 + key_t) * 64u + qd * 16, vo = (unsigned)(j0 + kq * 8) * 64u + d4 * 4;
#pragma unroll
        for (int i = 0; i < 4; i++) kk[i] = *reinterpret_cast<const f32x4*>(K + ko + 4 * i);
#pragma unroll
        for (int k = 0; k < 8; k++) vv[k] = *reinterpret_cast<const f32x4*>(V + vo + 64 * k);
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int e = tid + i * 256 * NSUB;
        if (e < rep * 64) qs[e] = qreg[i];
    }
    const int L = pos + 1;
    const int j_hi = min(L, split_lo + a.keys_per_split);
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) a.part_cnt[r] = (L + a.keys_per_split - 1) / a.keys_per_split;
    float* ml = a.part_ml + (((size_t)sp * SK_ROWS_CAP + r) * a.n_q + g * rep) * 2;
    SK_STAMP(1);                                                  // loads issued, q and position arrived
    if (split_lo >= j_hi) return;                                 // empty split: the consumer stops at part_cnt
    if (t < 8) { run_m[sub][t] = -INFINITY; run_l[sub][t] = 0.f; }
    f32x4 o[4];
#pragma unroll
    for (int i = 0; i < 4; i++) o[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // every group walks the same (block-uniform) number of tile rounds; a group whose tile is empty only keeps the barriers
    const int rounds = (j_hi - split_lo + AT_KB * NSUB - 1) / (AT_KB * NSUB);
    for (int it = 0; it < rounds; it++) {
        const int j0 = split_lo + (it * NSUB + sub) * AT_KB;
        const int n = max(0, min(AT_KB, j_hi - j0));
        if (it > 0) {                                             // later rounds of a long split (max_pos > 16 * 64 NSUB): same loads
            const unsigned ko = (unsigned)(j0 + key_t) * 64u + qd * 16, vo = (unsigned)(j0 + kq * 8) * 64u + d4 * 4;
#pragma unroll
            for (int i = 0; i < 4; i++) kk[i] = *reinterpret_cast<const f32x4*>(K + ko + 4 * i);
#pragma unroll
            for (int k = 0; k < 8; k++) vv[k] = *reinterpret_cast<const f32x4*>(V + vo + 64 * k);
        }
#pragma unroll
        for (int k = 0; k < 8; k++) vv[k] = kq * 8 + k < n ? vv[k] : (f32x4){0.f, 0.f, 0.f, 0.f};     // rows past the length may hold anything
        __syncthreads();                                          // qs ready; previous round's ps consumed
        if (it == 0) SK_STAMP(2);
        if (n > 0) {
#pragma unroll
            for (int h = 0; h < 8; h++) {
                if (h < rep) {
                    float acc = 0.f;
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const f32x4 qv = *reinterpret_cast<const f32x4*>(&qs[h * 64 + qd * 16 + 4 * i]);
                        acc += kk[i][0] * qv[0] + kk[i][1] * qv[1] + kk[i][2] * qv[2] + kk[i][3] * qv[3];
                    }
                    acc += dpp_mov_f32<0xB1, 0xf>(0.f, acc);      // quad_perm [1,0,3,2]: the four dim quarters of a key sit in one quad
                    acc += dpp_mov_f32<0x4E, 0xf>(0.f, acc);      // quad_perm [2,3,0,1]
                    if (qd == 0) ps[sub][h * AT_KB + key_t] = key_t < n ? acc * 0.125f : -INFINITY;
                }
            }
        }
        __syncthreads();
        if (it == 0) SK_STAMP(3);                                 // K arrived, scores done
        if (n > 0) {
            for (int h = w; h < rep; h += 4) {                   // per head: tile max / exp / sum, merged into the running pair
                const float s0 = ps[sub][h * AT_KB + lane];
                const float mt = wave_max(s0);
                const float mo = run_m[sub][h], mn = fmaxf(mo, mt);
                const float p0 = __expf(s0 - mn);
                ps[sub][h * AT_KB + lane] = p0;
                const float lt = wave_sum(p0);
                if (lane == 0) {
                    tile_scale[sub][h] = __expf(mo - mn);          // 0 on the first tile (mo = -inf)
                    run_l[sub][h] = run_l[sub][h] * tile_scale[sub][h] + lt;
                    run_m[sub][h] = mn;
                }
            }
        }
        __syncthreads();
        if (it == 0) SK_STAMP(4);                                 // softmax done
        if (n > 0) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int h = min(hs * 4 + i, rep - 1);               // a head beyond the group repeats the last one (not stored)
                o[i] *= tile_scale[sub][h];
#pragma unroll
                for (int k4 = 0; k4 < 2; k4++) {
                    const f32x4 pa = *reinterpret_cast<const f32x4*>(&ps[sub][h * AT_KB + kq * 8 + 4 * k4]);
#pragma unroll
                    for (int e = 0; e < 4; e++) o[i] += pa[e] * vv[4 * k4 + e];
                }
            }
        }
    }
    SK_STAMP(5);                                                  // V arrived, PV done (all rounds)
    // sum the eight key eighths and merge the groups through LDS, then store the unnormalised partial output
#pragma unroll
    for (int i = 0; i < 4; i++)
        if (hs * 4 + i < rep) *reinterpret_cast<f32x4*>(&po_s[sub][(kq * 8 + hs * 4 + i) * 64 + d4 * 4]) = o[i];
    __syncthreads();
    float* po = a.part_o + ((size_t)sp * SK_ROWS_CAP + r) * a.n_q * 64 + (size_t)g * rep * 64;
    for (int e = tid; e < rep * 64; e += 256 * NSUB) {
        const int h = e >> 6;
        float M = run_m[0][h];
#pragma unroll
        for (int u = 1; u < NSUB; u++) M = fmaxf(M, run_m[u][h]);
        float o = 0.f;
#pragma unroll
        for (int u = 0; u < NSUB; u++) {
            const float sc = NSUB == 1 ? 1.f : __expf(run_m[u][h] - M);          // 0 for a group that saw no key (max = -inf)
            const float* pp = &po_s[u][e];
            o += sc * (((pp[0] + pp[512]) + (pp[1024] + pp[1536])) + ((pp[2048] + pp[2560]) + (pp[3072] + pp[3584])));
        }
        if (a.arrive) st_agent(&po[e], o); else po[e] = o;
    }
    if (tid < rep) {
        float M = run_m[0][tid], l = 0.f;
#pragma unroll
        for (int u = 1; u < NSUB; u++) M = fmaxf(M, run_m[u][tid]);
#pragma unroll
        for (int u = 0; u < NSUB; u++) l += (NSUB == 1 ? 1.f : __expf(run_m[u][tid] - M)) * run_l[u][tid];
        if (a.arrive) { st_agent(&ml[tid * 2], M); st_agent(&ml[tid * 2 + 1], l); }
        else { ml[tid * 2] = M; ml[tid * 2 + 1] = l; }
    }
    SK_STAMP(6);
    SK_STAMP_FLUSH;
    if (!a.arrive) return;
    // the last split of this (row, kv head) to get here combines them (sk_finish_x's order: the same bits as k_prep<true>) and leaves
    // the planes of the group's rep x 64 columns
    const int ns = (L + a.keys_per_split - 1) / a.keys_per_split;
    if (!last_arriver(a.arrive + r * a.n_kv + g, ns)) return;
    for (int e = tid; e < rep * 64; e += 256 * NSUB) {
        const int c = g * rep * 64 + e;
        float mv[SK_MAXSPLIT], lv[SK_MAXSPLIT], ov[SK_MAXSPLIT];
#pragma unroll
        for (int s = 0; s < SK_MAXSPLIT; s++) {
            if (s < a.nsplit) {
                const float* mlp = a.part_ml + (((size_t)s * SK_ROWS_CAP + r) * a.n_q + (c >> 6)) * 2;
                mv[s] = ld_agent(mlp); lv[s] = ld_agent(mlp + 1);
                ov[s] = ld_agent(a.part_o + ((size_t)s * SK_ROWS_CAP + r) * a.n_q * 64 + c);
            }
        }
        float M = -INFINITY;
#pragma unroll
        for (int s = 0; s < SK_MAXSPLIT; s++) if (s < ns) M = fmaxf(M, mv[s]);
        float acc = 0.f, den = 0.f;
#pragma unroll
        for (int s = 0; s < SK_MAXSPLIT; s++) {
            if (s < ns) {
                const float w = __expf(mv[s] - M);
                den += w * lv[s];
                acc += w * ov[s];
            }
        }
        put_planes(a.pre, r, c, acc * (1.f / den));
    }
}
// The same kernel on the matrix cores (fp32 accuracy: three exact bf16 planes per operand, six products; see attn_role, the form inside
// k_step).  Grid, split layout and outputs are k_attn's.  A 64-key tile group = 4 waves x 16 keys: K rows / V rows go global -> registers
// in operand order (requested at entry, before the position has arrived), S^T = K Q^T and O^T = V^T P^T keep the head on the lane's
// column in both products, so the running (max, sum) and the rescale factor of a head never leave its lane; the waves of a block are
// merged once, through LDS, at the end.  After the loads: ~3.3 us of scalar FMAs and four block barriers -> ~1.5 us.
#define ATM_QLD 68
template <int NSUB>
__global__ __launch_bounds__(256 * NSUB) void k_attn_m(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) float qs[16 * ATM_QLD];                 // heads >= rep stay zero
    __shared__ __attribute__((aligned(16))) float po[4 * NSUB][8 * 64];             // per wave: O^T as [head][dim]
    __shared__ float wm[4 * NSUB][16], wl[4 * NSUB][16];
    const int rep = a.rep;
    const int sp = blockIdx.x, g = blockIdx.y, r = blockIdx.z;    // grid = (key split, kv head, row)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave of the block; its tile group is wv >> 2, its 16 keys (wv & 3)
    const int sub = wv >> 2, wk = wv & 3;
    const int c = lane & 15, g4 = lane >> 4;
    const int split_lo = sp * a.keys_per_split;
    float qreg[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int e = tid + i * 256 * NSUB;
        qreg[i] = a.q[(size_t)r * a.n_q * 64 + g * rep * 64 + min(e, rep * 64 - 1)];
    }
    int seq, pos;
    a.rm.get(r, seq, pos);
    if (gridDim.z > 4 && split_lo >= pos + 1) return;           // (many rows: skip the splits past the length before fetching, as k_attn)
    const int seq_s = a.rm.prefill ? a.rm.seq0 : (a.rm.slots ? seq : r);
    const float* K = a.kc + ((size_t)seq_s * a.n_kv + g) * a.max_pos * 64;
    const float* V = a.vc + ((size_t)seq_s * a.n_kv + g) * a.max_pos * 64;
    f32x4 kr[4]; float vr[16];
    auto fetch = [&](int kb) {                                    // this wave's 16 keys from kb on, in operand order
        const float* kp = K + (size_t)(kb + c) * 64 + 8 * g4;
        kr[0] = *reinterpret_cast<const f32x4*>(kp);      kr[1] = *reinterpret_cast<const f32x4*>(kp + 4);
        kr[2] = *reinterpret_cast<const f32x4*>(kp + 32); kr[3] = *reinterpret_cast<const f32x4*>(kp + 36);
        const float* vp = V + (size_t)(kb + 4 * g4) * 64 + c;
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int j = 0; j < 4; j++) vr[4 * t + j] = vp[j * 64 + 16 * t];
    };
    fetch(split_lo + sub * AT_KB + 16 * wk);
    for (int e = rep * ATM_QLD + tid; e < 16 * ATM_QLD; e += 256 * NSUB) qs[e] = 0.f;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int e = tid + i * 256 * NSUB;
        if (e < rep * 64) qs[(e >> 6) * ATM_QLD + (e & 63)] = qreg[i];
    }
    const int L = pos + 1;
    const int j_hi = min(L, split_lo + a.keys_per_split);
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) a.part_cnt[r] = (L + a.keys_per_split - 1) / a.keys_per_split;
    float* ml = a.part_ml + (((size_t)sp * SK_ROWS_CAP + r) * a.n_q + g * rep) * 2;
    if (split_lo >= j_hi) return;                                 // empty split: the consumer stops at part_cnt
    __syncthreads();                                              // q in LDS
    bf16x8 qb[2][3];
#pragma unroll
    for (int s2 = 0; s2 < 2; s2++) {
        const float* qp = qs + c * ATM_QLD + 32 * s2 + 8 * g4;
        planes8(*reinterpret_cast<const f32x4*>(qp), *reinterpret_cast<const f32x4*>(qp + 4), qb[s2][0], qb[s2][1], qb[s2][2]);
    }
    f32x4 o[4];                                                   // O^T: rows = dims 16 t + 4 g4 + reg, column = head c
#pragma unroll
    for (int t = 0; t < 4; t++) o[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float mrun = -INFINITY, lrun = 0.f;
    const int rounds = (j_hi - split_lo + AT_KB * NSUB - 1) / (AT_KB * NSUB);
    for (int it = 0; it < rounds; it++) {
        const int kb = split_lo + (it * NSUB + sub) * AT_KB + 16 * wk;
        bf16x8 ka[2][3];
        s16x4 va[4][3];
        planes8(kr[0], kr[1], ka[0][0], ka[0][1], ka[0][2]);
        planes8(kr[2], kr[3], ka[1][0], ka[1][1], ka[1][2]);
#pragma unroll
        for (int t = 0; t < 4; t++) {
            f32x4 v4;
#pragma unroll
            for (int j = 0; j < 4; j++) v4[j] = kb + 4 * g4 + j < j_hi ? vr[4 * t + j] : 0.f;      // rows past the length may hold anything
            planes4(v4, va[t][0], va[t][1], va[t][2]);
        }
        if (it + 1 < rounds) fetch(kb + AT_KB * NSUB);             // a long split's next round (max_pos > 16 x 64 NSUB)
        if (kb >= j_hi) continue;                                  // (wave-uniform) no key of this wave is live
        f32x4 sc = {0.f, 0.f, 0.f, 0.f};
        sc = mm6_32(sc, ka[0], qb[0]);
        sc = mm6_32(sc, ka[1], qb[1]);
        float mt = -INFINITY;
#pragma unroll
        for (int j = 0; j < 4; j++) { sc[j] = kb + 4 * g4 + j < j_hi ? sc[j] * 0.125f : -INFINITY; mt = fmaxf(mt, sc[j]); }
        mt = rows4_max(mt);
        const float mn = fmaxf(mrun, mt);                          // finite: the wave holds a live key
        const float scale = __expf(mrun - mn);                    // 0 on the wave's first live tile
        f32x4 p;
        float lt = 0.f;
#pragma unroll
        for (int j = 0; j < 4; j++) { p[j] = __expf(sc[j] - mn); lt += p[j]; }
        lt = rows4_sum(lt);
        lrun = lrun * scale + lt;
        mrun = mn;
        s16x4 pb[3];
        planes4(p, pb[0], pb[1], pb[2]);
#pragma unroll
        for (int t = 0; t < 4; t++) o[t] = mm6_16(o[t] * scale, va[t], pb);
    }
    // merge the waves of the block through LDS; store the unnormalised partial output and (max, sum) of the split
    if (c < 8) {
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int j = 0; j < 4; j++) po[wv][c * 64 + 16 * t + 4 * g4 + j] = o[t][j];
    }
    if (g4 == 0) { wm[wv][c] = mrun; wl[wv][c] = lrun; }
    __syncthreads();
    float* pout = a.part_o + ((size_t)sp * SK_ROWS_CAP + r) * a.n_q * 64 + (size_t)g * rep * 64;
    for (int e = tid; e < rep * 64; e += 256 * NSUB) {
        const int h = e >> 6;
        float M = wm[0][h];
#pragma unroll
        for (int u = 1; u < 4 * NSUB; u++) M = fmaxf(M, wm[u][h]);
        float acc = 0.f;
#pragma unroll
        for (int u = 0; u < 4 * NSUB; u++) acc += __expf(wm[u][h] - M) * po[u][e];      // a wave without a live key: max = -inf, weight 0, O^T = 0
        pout[e] = acc;
    }
    if (tid < rep) {
        float M = wm[0][tid], l = 0.f;
#pragma unroll
        for (int u = 1; u < 4 * NSUB; u++) M = fmaxf(M, wm[u][tid]);
#pragma unroll
        for (int u = 0; u < 4 * NSUB; u++) l += __expf(wm[u][tid] - M) * wl[u][tid];
        ml[tid * 2] = M; ml[tid * 2 + 1] = l;
    }
}
static void launch_attn(const AttnArgs& a, int rows, hipStream_t s) {
    const dim3 grid(a.nsplit, a.n_kv, rows);
    const int nsub = a.keys_per_split / AT_KB;
    // few rows are latency-bound and take the matrix-core kernel (1 / 2 / 8 rows: 614 -> 596, 659 -> 638, 908 -> 887 us per step); many rows are
    // throughput-bound, where its plane splits cost as many VALU operations as the scalar kernel's FMAs (16 rows: 1212 -> 1229, 32: 1276 -> 1356)
    static const bool scalar = getenv("CV2_DECODE_ATTN") && getenv("CV2_DECODE_ATTN")[0] == '0';      // A/B switch (diagnostics): the scalar-FMA kernels everywhere
    if (scalar || rows > 8) {
        if (nsub >= 4) hipLaunchKernelGGL(k_attn<4>, grid, dim3(1024), 0, s, a);
        else if (nsub == 2) hipLaunchKernelGGL(k_attn<2>, grid, dim3(512), 0, s, a);
        else hipLaunchKernelGGL(k_attn<1>, grid, dim3(256), 0, s, a);
        return;
    }
    if (nsub >= 4) hipLaunchKernelGGL(k_attn<4>, grid, dim3(1024), 0, s, a);         // (256-key splits: the matrix-core form at 1024 threads spilled 10 VGPRs -- the scalar kernel serves them)
    else if (nsub == 2) hipLaunchKernelGGL(k_attn_m<2>, grid, dim3(512), 0, s, a);
    else hipLaunchKernelGGL(k_attn_m<1>, grid, dim3(256), 0, s, a);
}

// split combine as its own pass (used when many rows share a launch: inside the O-projection every block would redo it)
struct CombArgs { const float* part_o; const float* part_ml; const int* part_cnt; float* out; int K; int nsplit; };
__global__ __launch_bounds__(128) void k_attn_combine(CombArgs a) {
    const int r = blockIdx.x, k = threadIdx.x * 8;
    if (k >= a.K) return;
    SkinnyX X{nullptr, a.part_o, a.nsplit, nullptr, 0.f, nullptr, a.part_ml, a.part_cnt};
    const f32x8 v = sk_load_x<true>(X, r, a.K, k);
    *reinterpret_cast<f32x8*>(a.out + (size_t)r * a.K + k) = v;
}

// ------------------------------------------------------------------ k_store: out = W f(x) (+bias)  (o-proj, down-proj, head)
struct StoreArgs {
    const uint16_t* W; const float* bias;       // bias may be null
    SkinnyX X; int KS, rows, K, N;
    float* out;                                 // gridDim.y == 1: [rows][N]; else partials [gridDim.y][SK_ROWS_CAP][N]
    // NEXT epilogue (many-row O projection): the block finishes the residual stream for its 16 columns and PREPARES the next kernel's
    // operand, so that no k_prep launch is needed between the O projection and gate/up: x_mid = resid + out -> x_out (fp32), the hi / lo
    // planes of next_g . x_mid in the layout the PRE kernels copy, and this block's share of every row's sum of squares (the consumer
    // sums the N / 16 shares in a fixed order and scales its OUTPUTS by the row's rstd: W (g . x) rstd = W (g . x rstd))
    const float* resid; const float* next_g; float* x_out; uint16_t* next_pre; float* next_sq;
    // LAST (many-row down projection, split-K over gridDim.y): the last K-slice block of a column tile to finish folds the slices in index
    // order onto the residual stream and runs the NEXT epilogue for the following layer's QKV kernel (what k_prep<false> did as a launch
    // of its own).  arrive = [N / 16] counters, zero between launches
    int* arrive;
    const float* sq; float eps;                 // PRE operand left un-normalised by k_prep (the head's final norm): out = (W (g . x)) rsqrt(sq[row] / K + eps) (+ bias)
};
template <int NB, int MAXKS, bool ATT = false, bool PRE = false, bool NEXT = false, bool LAST = false>
__global__ __launch_bounds__(256) void k_store(StoreArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* res = skinny_core<NB, 1, 4, MAXKS, ATT, PRE>(a.W, blockIdx.x, a.KS, a.rows, a.K, a.X, smem);
    const int ld = NB * 16 + 1;
    const int n0 = blockIdx.x * 16;
    float* out = a.out + (gridDim.y > 1 ? (size_t)blockIdx.y * SK_ROWS_CAP * a.N : 0);
    auto next = [&](int r, int i, float v) {             // v = the finished residual stream at (row r, column n0 + i)
        const int c = n0 + i;
        a.x_out[(size_t)r * a.N + c] = v;
        put_planes(a.next_pre, r, c, a.next_g[c] * v);
        float sq = v * v;                                 // the row's 16 columns sit in 16 consecutive lanes
        sq += dpp_mov_f32<0xB1, 0xf>(0.f, sq);            // quad_perm [1,0,3,2]
        sq += dpp_mov_f32<0x4E, 0xf>(0.f, sq);            // quad_perm [2,3,0,1]
        sq += __shfl_xor(sq, 4);
        sq += __shfl_xor(sq, 8);
        if (i == 0) a.next_sq[(size_t)r * gridDim.x + blockIdx.x] = sq;
    };
    for (int e = threadIdx.x; e < a.rows * 16; e += blockDim.x) {
        const int r = e >> 4, i = e & 15;
        float v = res[i * ld + r];
        if (PRE && a.sq) { v *= rsqrtf(a.sq[r] / (float)a.K + a.eps); asm volatile("" : "+v"(v)); }      // (rounded before the bias, as skinny_core's deferred scale)
        if (a.bias && blockIdx.y == 0) v += a.bias[n0 + i];
        if (LAST) st_agent(&out[(size_t)r * a.N + n0 + i], v);
        else if (NEXT) next(r, i, v + a.resid[(size_t)r * a.N + n0 + i]);
        else out[(size_t)r * a.N + n0 + i] = v;
    }
    if (LAST) {
        if (!last_arriver(a.arrive + blockIdx.x, gridDim.y)) return;
        for (int e = threadIdx.x; e < a.rows * 16; e += blockDim.x) {
            const int r = e >> 4, i = e & 15;
            float v = a.resid[(size_t)r * a.N + n0 + i], pv[SK_MAXNP];
#pragma unroll
            for (int p = 0; p < SK_MAXNP; p++)           // all slices in flight together; folded in index order (k_prep<false>'s)
                pv[p] = ld_agent(a.out + ((size_t)min(p, (int)gridDim.y - 1) * SK_ROWS_CAP + r) * a.N + n0 + i);
#pragma unroll
            for (int p = 0; p < SK_MAXNP; p++) if (p < (int)gridDim.y) v += pv[p];
            next(r, i, v);
        }
    }
}

// ------------------------------------------------------------------ k_gateup
struct GateUpArgs {
    const uint16_t* W; SkinnyX X; int KS, rows, K, inter;
    float* h;                                   // [rows][inter]
    uint16_t* hpre;                             // PRE kernels: h written as prepared hi/lo planes for the down projection instead
    const float* sq; int nsq; float eps;        // PRE operand left un-normalised by k_store<.., NEXT>: rstd[r] = rsqrt(sum_b sq[r][b] / K + eps)
};
#ifdef CV2_STAMPS
__global__ void k_stamp_set(int v) { if (threadIdx.x == 0) g_stamp_slot = v < 0 ? g_stamp_slot + 1 : v; }
#define STAMP_SET_ON(st_, v) hipLaunchKernelGGL(k_stamp_set, dim3(1), dim3(64), 0, st_, v)
#define STAMP_SET(v) STAMP_SET_ON(s, v)
extern "C" int cv2_debug_stamps(unsigned long long* out_host) {
    CV2_HIP(hipDeviceSynchronize());
    CV2_HIP(hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 64 * 8));
    return 0;
}
#endif
#ifndef CV2_STAMPS
#define STAMP_SET(v) do { } while (0)
#define STAMP_SET_ON(st_, v) do { } while (0)
#endif
// FT feature tiles per block (PRE only): the 112 KB operand of a 32-row launch leaves room for ONE block per CU, and inter / 16 = 304
// blocks then run in two rounds on 256 CUs; 152 blocks stage the operand once and stream two tiles' weights past it.
template <int NB, bool PRE = false, int FT = 1>
__global__ __launch_bounds__(512) void k_gateup(GateUpArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wr = (threadIdx.x >> 6) % 2;
    const int ld = NB * 16 + 1;
  for (int ft = 0; ft < FT; ft++) {
    const int ftile = blockIdx.x * FT + ft;
    float* res = skinny_core<NB, 2, 4, 8, false, PRE, SkNoHook, (FT > 1)>(a.W, ftile * 2 + wr, a.KS, a.rows, a.K, a.X, smem, SkNoHook(), ft > 0);   // rows 0-15 gate, 16-31 up
    for (int e = threadIdx.x; e < a.rows * 16; e += blockDim.x) {
        const int r = e >> 4, i = e & 15;
        float rs = 1.f;
        if (PRE && a.sq) {                       // the row's statistic from the producers' shares: the 16 threads of a row sum them in a fixed order
            float t = 0.f;
            for (int b = i; b < a.nsq; b += 16) t += a.sq[(size_t)r * a.nsq + b];
            t += dpp_mov_f32<0xB1, 0xf>(0.f, t);
            t += dpp_mov_f32<0x4E, 0xf>(0.f, t);
            t += __shfl_xor(t, 4);
            t += __shfl_xor(t, 8);
            rs = rsqrtf(t / (float)a.K + a.eps);
        }
        const float g = res[i * ld + r] * rs, u = res[(16 + i) * ld + r] * rs;
        const float hv = (g / (1.f + __expf(-g))) * u;
        if (PRE && a.hpre) {                     // LDS B-operand order of skinny.h: [k-block][row tile][hi, lo][quarter][row][8]
            const int c = ftile * 16 + i;
            uint16_t* d = a.hpre + ((size_t)((c >> 5) * 2 + (r >> 4)) * 2) * 512 + (((c >> 3) & 3) * 16 + (r & 15)) * 8 + (c & 7);
            const uint16_t hb = f2bf(hv);
            d[0] = hb;
            d[512] = f2bf(hv - bf2f(hb));
        } else {
            a.h[(size_t)r * a.inter + ftile * 16 + i] = hv;
        }
    }
    if (ft + 1 < FT) __syncthreads();            // the tile is consumed: the next call reuses the reduction buffers
  }
}

// ------------------------------------------------------------------ k_step: a whole one-row decode step in one launch (chain.h)
// Rows of one decode-step launch.  Three forms (all keep every sum of a row in k_step's order, so ids do not depend on the form):
//   k_step<true>  every row is a chain of k_step's blocks of its own, the rows' chains interleaved in the grid so that the blocks that
//                 stream one weight tile (one per row) share an XCD and its L2;
//   k_step2       the rows in pairs: one chain per pair, every GEMV block serves both rows as MFMA columns 0 / 1 (row2_core);
//   k_step4       the rows in fours: MFMA columns 0 .. 3 (row4_core; the O role stays two rows per block).
// Measured on MI355X at the configs[1] context (positions 323 .. 387), us per step (profiles/r4_rows_sweep.txt):
//   rows            1     2     3     4     6     8    10    12    16    20    24
//   k_step<true>   332   445   496   590   746   903
//   k_step2         -    430   564   554   616   708   822   877  1076
//   k_step4         -     -     -    561   691   709   801   826   921  1065  1175
//   launches       599   643   734   767   823   888  1000  1064  1162  1181  1205
// Every chain adds its ~12 600 blocks to a grid of which 512 are resident (2 blocks of 512 threads per CU at 109-125 VGPRs): ~80 us
// per one-row chain, ~92 us per pair, ~120-150 us per four (their operand gathers are 2 / 4 x a row's: the Q role's x + 2 partials is
// 86 KB at four rows against ~50 GB/s per workgroup).  Several chains read their weight tiles with plain loads (the sibling chains'
// blocks follow on the same XCD and hit its L2), one chain alone with non-temporal ones.  Policy: pairs up to 8 rows (3 rows: one
// chain per row), fours from 9 to 24, the launches beside other streams' kernels (CV2_DECODE_SHARED) and from 25 rows on.
#define CH_MAX_ROWS 24                           // hand-off buffer sets carved per engine
#define CH_ROWS_DEFAULT 24
static int chain_rows() {                        // CV2_CHAIN_ROWS = 1 .. 24: A/B switch (diagnostics)
    static const int v = [] { const char* e = getenv("CV2_CHAIN_ROWS"); const int x = e ? atoi(e) : CH_ROWS_DEFAULT; return x < 1 ? 1 : (x > CH_MAX_ROWS ? CH_MAX_ROWS : x); }();
    return v;
}
struct StepLayer { const uint16_t *wqkv, *wo, *wgu, *wdown; const float *bqkv, *ln1, *ln2; float *kc, *vc; };
// The table's pointers are loaded from memory, so the compiler knows them as generic pointers and would emit flat_load / flat_store for
// everything behind them (weights, norms, biases, cache rows) -- instructions that also count in lgkmcnt, so that every LDS or scalar wait
// waits for the weight fragments in flight.  Read through a view of the table whose members are typed as global pointers, the casts to
// the roles' generic parameters are visible to the compiler's address-space inference: global_load / global_store.
#define CV2_AS1 __attribute__((address_space(1)))
struct StepLayerG { const CV2_AS1 uint16_t *wqkv, *wo, *wgu, *wdown; const CV2_AS1 float *bqkv, *ln1, *ln2; CV2_AS1 float *kc, *vc; };
static_assert(sizeof(StepLayerG) == sizeof(StepLayer), "the global-pointer view of the layer table");
__device__ __forceinline__ StepLayer step_layer(const StepLayer* tab, int layer) {
    const StepLayerG g = reinterpret_cast<const StepLayerG*>(tab)[layer];
    return StepLayer{(const uint16_t*)g.wqkv, (const uint16_t*)g.wo, (const uint16_t*)g.wgu, (const uint16_t*)g.wdown,
                     (const float*)g.bqkv, (const float*)g.ln1, (const float*)g.ln2, (float*)g.kc, (float*)g.vc};
}
struct StepArgs {
    const StepLayer* layers; int n_layers;
    const uint16_t* wdec; const float* bdec; const float* final_norm; float* logits;
    const float* xin;                            // layer 0 input [hidden] (the embedding k_sample gathered)
    const int* state;                            // slot 0
    const float* cosT; const float* sinT;
    u64* gran; unsigned gran_bytes; const unsigned* epoch; int* err;
    int H, NQ, inter, n_q, n_kv, rep, max_pos, ntiles;
    float eps;
    int per;                                     // blocks per layer: Q, A, O, GU, D in this order
    unsigned gl, off_dg, off_qg, off_kv, off_ag, off_hg;     // granules per layer; offsets of the buffers inside a layer (x_mid at 0)
    int dbg_layer;
    // several rows (k_step<true>): every row is a chain of its own (own granule buffers, own slot: state record, KV cache, pending input);
    // the rows' blocks are interleaved in the grid so that the rows of one weight tile run on one XCD, one after the other
    int n_rows; const int* row_slots;            // row -> slot (null: identity)
    unsigned row_gran;                           // granules per row (n_layers * gl)
    long kv_slot;                                // floats between two slots of a layer's cache (n_kv * max_pos * 64)
    int ldl, head_blocks;                        // logits row stride (vocab_pad), head blocks per row (vocab_pad / 16)
    int spec;                                    // Gran::spec for k_step<true>
    const int* dbg_skip;                         // test hook (cv2_llm_debug_skip_publish): ((layer << 16) | r) + 1 of the Q-role block r of `layer` that does not publish; 0 = none (the same key in every one-launch form)
};
#ifdef CV2_STAMPS
__device__ unsigned long long g_chain_t[1024][8] = {};      // per block of one layer: start, result, published, operand ready (100 MHz ticks)
#define CH_T(i) do { if (dbg && threadIdx.x == 0) g_chain_t[r_dbg][i] = __builtin_amdgcn_s_memrealtime(); } while (0)
// slot 7 (round 6, tools/dbg_chain_tails.py): where the block ran -- XCC_ID (the XCD) in bits 32.., HW_ID (CU_ID 11:8, SH_ID 12, SE_ID 15:13) below
#define CH_WHERE() do { if (dbg && threadIdx.x == 0) { unsigned xcc_, hw_;                                              \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_HW_ID)" : "=s"(xcc_), "=s"(hw_));  \
        g_chain_t[r_dbg][7] = ((unsigned long long)(xcc_ & 15u) << 32) | hw_ | (1ull << 48); } } while (0)
extern "C" int cv2_debug_chain(unsigned long long* out_host) {
    CV2_HIP(hipDeviceSynchronize());
    CV2_HIP(hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_chain_t), sizeof(unsigned long long) * 1024 * 8));
    return 0;
}
#else
#define CH_T(i) do { } while (0)
#define CH_WHERE() do { } while (0)
#endif

// Attention of one 128-key tile for the rep query heads of kv head g over the keys ALREADY in the cache (positions < pos): eight waves
// take 16 keys each (attn_role below).  q arrives as granules from the Q role; the cache rows are plain reads, requested before
// anything else.  The new token's own key / value row is NOT waited for here (the key
// and value heads are the last blocks of the Q role: the tile that owned the row was the layer's straggler): the O role merges it as
// one more partial (chain.h, OpAtt).
#define AT_QLD 68                                // floats per q row in LDS (64 + 4: conflict-free ds_read_b128 of 16 rows)
#define AT_SMEM_FLOATS (16 * AT_QLD + 8 * 8 * 64 + 2 * 8 * 16)
#ifdef CV2_STAMPS
#define AT_T(i) do { if (dbg_slot >= 0 && threadIdx.x == 0) g_chain_t[dbg_slot][i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define AT_T(i) do { } while (0)
#endif
// One 128-key tile of one kv head against the step's rep (<= 8) query heads; wave w owns keys [16 w, 16 w + 16).
//   before q arrives (the block has nothing else to do): the wave's K rows and V rows are fetched in MFMA operand order and split
//     into planes -- K[key l & 15][dims 8 (l >> 4) .., + 32] = A operand of S^T = K Q^T, V[key 4 (l >> 4) + j][dim 16 t + (l & 15)]
//     = B operand of O = P V (16x16x16);
//   after: q -> LDS -> B fragments (head = column), 12 MFMAs -> S^T with its column on the lane and four keys in the registers, which IS
//     the A-operand layout of the 16x16x16 product (k = 4 (l >> 4) + j): softmax over the wave's 16 keys per head (registers + two
//     permlane swaps), P planes, 24 MFMAs -> the wave's partial O[head][dim]; the eight partials are merged through LDS with their
//     (max, sum) and published as the tile's granules (o unnormalised, max, sum -- what OpAtt merges across tiles).
// The scalar form this replaces took 3.3-4.0 us from q to the published tile (scores 1.2-1.9, softmax 0.6, PV 1.15, merge 0.35).
// The role in three parts, so that k_step1's QA blocks can put their own q computation between the second and the third:
//   att_prepare  cache rows -> registers -> planes (before q exists);   att_compute  q rows in LDS -> published tile.
struct AttTile { bf16x8 ka[2][3]; s16x4 vb[4][3]; int n; };
__device__ __forceinline__ void att_prepare(const float* K, const float* V, int pos, int j0, int nh, AttTile& T, char* smem) {
    float* qs = reinterpret_cast<float*>(smem);       // [16][AT_QLD]; rows >= nh stay zero
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g4 = lane >> 4;
    const int kb = j0 + 16 * w;                        // the wave's first key
    const int n = max(0, min(16, pos - kb));           // its cached keys (rows past the length may hold anything)
    T.n = n;
    f32x4 kr[4];
    float vr[16];
    {
        const float* kp = K + (size_t)(kb + c) * 64 + 8 * g4;
        kr[0] = *reinterpret_cast<const f32x4*>(kp);      kr[1] = *reinterpret_cast<const f32x4*>(kp + 4);
        kr[2] = *reinterpret_cast<const f32x4*>(kp + 32); kr[3] = *reinterpret_cast<const f32x4*>(kp + 36);
        const float* vp = V + (size_t)(kb + 4 * g4) * 64 + c;
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int j = 0; j < 4; j++) vr[4 * t + j] = vp[j * 64 + 16 * t];
    }
    for (int e = nh * AT_QLD + tid; e < 16 * AT_QLD; e += R1_THREADS) qs[e] = 0.f;
    planes8(kr[0], kr[1], T.ka[0][0], T.ka[0][1], T.ka[0][2]);
    planes8(kr[2], kr[3], T.ka[1][0], T.ka[1][1], T.ka[1][2]);
#pragma unroll
    for (int t = 0; t < 4; t++) {
        f32x4 v4;
#pragma unroll
        for (int j = 0; j < 4; j++) v4[j] = 4 * g4 + j < n ? vr[4 * t + j] : 0.f;
        planes4(v4, T.vb[t][0], T.vb[t][1], T.vb[t][2]);
    }
    // the planes must be IN registers before the polling starts: left to itself the compiler sinks the loads (and what hangs on them)
    // to their first use, behind the q wait
#pragma unroll
    for (int s = 0; s < 2; s++)
#pragma unroll
        for (int i = 0; i < 3; i++) asm volatile("" : "+v"(T.ka[s][i]));
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int i = 0; i < 3; i++) asm volatile("" : "+v"(T.vb[t][i]));
    __builtin_amdgcn_sched_barrier(0);
}
// q rows [nh][64] sit in qs (written by this block's threads, not yet synchronised); the tile's o [nh * 64] go to granules og_o ..,
// its (max, sum) pairs to og_ml + 2 h
// PO_H: heads per wave in the partial-output area (8; k_step1's one-head blocks: 1).  LDSKV: the tile's cache rows were left in LDS by
// att_dma_tile (kl / vl) instead of as planes in T -- they are turned into planes here, the keys first, the values behind the score products.
template <int PO_H = 8, bool LDSKV = false>
__device__ __forceinline__ void att_compute(const Gran& G, AttTile& T, int nh, unsigned og_o, unsigned og_ml, char* smem, int dbg_slot,
                                            const char* kl = nullptr, const char* vl = nullptr) {
    float* qs = reinterpret_cast<float*>(smem);
    float* po = qs + 16 * AT_QLD;                     // [8 waves][PO_H heads][64]
    float* wm = po + 8 * PO_H * 64; float* wl = wm + 128;    // [8 waves][16 heads] max, sum
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g4 = lane >> 4;
    const int n = T.n;
    __syncthreads();
    if constexpr (LDSKV) {                            // key rows: row 16 w + c, 16-byte chunks 2 g4, 2 g4 + 1, 8 + 2 g4, 9 + 2 g4 (slot = chunk ^ (row & 15))
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's own DMA (it reads only the rows it fetched): long landed, but say so
        const char* kp = kl + (size_t)(16 * w + c) * 256;
        const f32x4 k0 = *reinterpret_cast<const f32x4*>(kp + (((2 * g4) ^ c) << 4)), k1 = *reinterpret_cast<const f32x4*>(kp + (((2 * g4 + 1) ^ c) << 4));
        const f32x4 k2 = *reinterpret_cast<const f32x4*>(kp + (((8 + 2 * g4) ^ c) << 4)), k3 = *reinterpret_cast<const f32x4*>(kp + (((9 + 2 * g4) ^ c) << 4));
        planes8(k0, k1, T.ka[0][0], T.ka[0][1], T.ka[0][2]);
        planes8(k2, k3, T.ka[1][0], T.ka[1][1], T.ka[1][2]);
    }
    f32x4 sc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 2; s++) {
        const float* qp = qs + c * AT_QLD + 32 * s + 8 * g4;
        bf16x8 qb[3];
        planes8(*reinterpret_cast<const f32x4*>(qp), *reinterpret_cast<const f32x4*>(qp + 4), qb[0], qb[1], qb[2]);
        sc = mm6_32(sc, T.ka[s], qb);
    }
    if constexpr (LDSKV) {                            // value rows 16 w + 4 g4 + j, dims 16 t + c (the products above are still running)
#pragma unroll
        for (int t = 0; t < 4; t++) {
            f32x4 v4;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int r = 4 * g4 + j;
                const float v = *reinterpret_cast<const float*>(vl + (size_t)(16 * w + r) * 256 + (((4 * t + (c >> 2)) ^ r) << 4) + ((c & 3) << 2));
                v4[j] = r < n ? v : 0.f;
            }
            planes4(v4, T.vb[t][0], T.vb[t][1], T.vb[t][2]);
        }
    }
    AT_T(4);
    // sc[j] = score of key kb + 4 g4 + j against head c
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 4; j++) { sc[j] = 4 * g4 + j < n ? sc[j] * 0.125f : -INFINITY; mx = fmaxf(mx, sc[j]); }
    mx = rows4_max(mx);
    const float mref = mx == -INFINITY ? 0.f : mx;       // a wave without a cached key: P = 0, weight 0 in the merge
    f32x4 p;
    float ls = 0.f;
#pragma unroll
    for (int j = 0; j < 4; j++) { p[j] = __expf(sc[j] - mref); ls += p[j]; }
    ls = rows4_sum(ls);
    AT_T(5);
    s16x4 pa[3];
    planes4(p, pa[0], pa[1], pa[2]);
    f32x4 o[4];
#pragma unroll
    for (int t = 0; t < 4; t++) o[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 4; t++) o[t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pa[2], T.vb[t][0], o[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 4; t++) o[t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pa[1], T.vb[t][1], o[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 4; t++) o[t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pa[0], T.vb[t][2], o[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 4; t++) o[t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pa[1], T.vb[t][0], o[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 4; t++) o[t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pa[0], T.vb[t][1], o[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 4; t++) o[t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pa[0], T.vb[t][0], o[t], 0, 0, 0);
    // o[t][j] = head 4 g4 + j, dim 16 t + c
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int h = 4 * g4 + j;
        if (h < nh) {
#pragma unroll
            for (int t = 0; t < 4; t++) po[(w * PO_H + h) * 64 + 16 * t + c] = o[t][j];
        }
    }
    if (g4 == 0) { wm[w * 16 + c] = mx; wl[w * 16 + c] = ls; }
    __syncthreads();
    AT_T(6);
    if (tid < nh * 64) {
        const int h = tid >> 6, d = tid & 63;
        float M = wm[h];
#pragma unroll
        for (int i = 1; i < 8; i++) M = fmaxf(M, wm[i * 16 + h]);
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 8; i++) acc += __expf(wm[i * 16 + h] - M) * po[(i * PO_H + h) * 64 + d];       // (wave 0 always holds a cached key: M is finite)
        G.store(og_o + tid, acc);
    }
    if (tid < nh) {
        float M = wm[tid];
#pragma unroll
        for (int i = 1; i < 8; i++) M = fmaxf(M, wm[i * 16 + tid]);
        float l = 0.f;
#pragma unroll
        for (int i = 0; i < 8; i++) l += __expf(wm[i * 16 + tid] - M) * wl[i * 16 + tid];
        G.store(og_ml + tid * 2, M);
        G.store(og_ml + tid * 2 + 1, l);
    }
}
// k_step1's QA blocks: the tile's 128 key rows and 128 value rows (fp32, 256 B each) go to LDS by DMA and wait THERE for q -- as planes in
// registers (att_prepare) they would sit beside the block's 14 weight fragments per wave: 168 VGPRs, one block per CU.  One instruction =
// 1 KiB = four rows; lane l fetches chunk (l & 15) ^ (row & 15) of row l >> 4, so that chunk c of row r lands in slot c ^ (r & 15) and both
// fragment reads of att_compute (16 lanes: one chunk of 16 rows; or four rows x 16 consecutive dwords) touch every bank once.
__device__ __forceinline__ void att_dma_tile(const float* K, const float* V, int j0, char* kl, char* vl) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int r = 16 * w + 4 * i + (lane >> 4);
        const size_t off = (size_t)(j0 + r) * 64 + (((lane & 15) ^ (r & 15)) << 2);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(K + off),
                                         (__attribute__((address_space(3))) void*)(kl + (16 * w + 4 * i) * 256), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(V + off),
                                         (__attribute__((address_space(3))) void*)(vl + (16 * w + 4 * i) * 256), 16, 0, 0);
    }
}
__device__ __forceinline__ void attn_role(const Gran& G, const float* K, const float* V, int pos, int j0, int rep,
                                          unsigned qg, unsigned og, char* smem, int dbg_slot) {
    float* qs = reinterpret_cast<float*>(smem);
    const int tid = threadIdx.x;
    AttTile T;
    att_prepare(K, V, pos, j0, rep, T, smem);
    // q is polled directly: a few blocks per layer, one 8-byte load per thread -- cheaper than a sentinel round trip in front of the
    // sweep.  The poll starts once the previous layer's down projection has published (armed by the caller).
    {
        float q0 = 0.f;
        const bool mine = tid < rep * 64;
        G.sweep([&]() {
            if (!mine) return true;
            const u64 x0 = __hip_atomic_load((const gu64*)(G.base + qg + tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            q0 = __builtin_bit_cast(float, (unsigned)x0);
            return (unsigned)(x0 >> 32) == G.epoch;
        });
        if (mine) qs[(tid >> 6) * AT_QLD + (tid & 63)] = q0;
    }
    AT_T(3);
    att_compute(G, T, rep, og, og + rep * 64, smem, dbg_slot);
}

// MULTI: n_rows > 1.  Block order: layer by layer (then the head); inside a layer the role blocks go in groups of 8, a group's blocks
// once per row: position = group * 8 R + row * 8 + (block % 8).  Consecutive block ids go round the 8 XCDs, so the R blocks that stream
// one weight tile (one per row) share an XCD and follow each other within 8 R dispatches: one HBM read, R - 1 L2 hits.  A block still
// waits only for lower block ids (its own row's earlier roles), so the forward-progress argument of chain.h holds unchanged.
template <bool MULTI, bool NT = !MULTI>
__global__ __launch_bounds__(R1_THREADS) void k_step(StepArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int gb = blockIdx.x;
    int layer, r, row = 0;
    if (!MULTI) {
        layer = min(gb / a.per, a.n_layers);                 // the head's blocks (vocab_pad / 16 of them, possibly more than `per`) follow the layers
        r = gb - layer * a.per;
    } else {
        const int R = a.n_rows, lb = a.per * R;
        layer = min(gb / lb, a.n_layers);
        const int q = gb - layer * lb;
        const int nb = layer < a.n_layers ? a.per : a.head_blocks, full = nb >> 3, g8 = q / (8 * R);
        if (g8 < full) { const int rem = q - g8 * 8 * R; row = rem >> 3; r = g8 * 8 + (rem & 7); }
        else { const int rem = q - full * 8 * R, m = nb - full * 8; row = rem / m; r = full * 8 + rem - row * m; }
        const int slot = a.row_slots ? a.row_slots[row] : row;
        a.gran += (size_t)row * a.row_gran;
        a.state += slot * ST; a.err += slot * ST;
        a.xin += (size_t)row * a.H;
        a.logits += (size_t)row * a.ldl;
        a.kv_slot *= slot;
    }
    const int H = a.H;
    const int nQ = 2 * (a.n_q + 2 * a.n_kv), nA = a.ntiles * a.n_kv, nO = H / 16, nGU = a.inter / 16;
    const bool dbg = layer == a.dbg_layer;
    const int r_dbg = r, od = dbg ? r : -1; (void)r_dbg; (void)dbg;
    CH_T(0);
    CH_WHERE();
    Gran G;
    G.init(a.gran, a.gran_bytes, *a.epoch, a.err, MULTI && a.spec != 0);
    const unsigned gl = (unsigned)min(layer, a.n_layers - 1) * a.gl;        // this layer's granules (head: the last layer's)
    const unsigned gp = layer > 0 ? (unsigned)(layer - 1) * a.gl : 0u;      // the previous layer's
    if (layer >= a.n_layers) {      // head: final norm -> llm_decoder (+ bias) -> logits (read by k_sample, the next launch)
        OpFold op{&G, gl, gl + a.off_dg, H, nullptr, -1, gl + a.off_dg + H - 1};
        const float out = row1_core<1, 8, 4, true, NT>(a.wdec, r, 0, H / 32, H, 0, H / 32, op, a.final_norm, a.eps, smem);
        if (tid < 16) a.logits[r * 16 + tid] = out + a.bdec[r * 16 + tid];
        return;
    }
    StepLayer L = step_layer(a.layers, layer);
    if (MULTI) { L.kc += a.kv_slot; L.vc += a.kv_slot; }
    OpFold xin{&G, gp, gp + a.off_dg, H, layer == 0 ? a.xin : nullptr, od, gp + a.off_hg + a.inter - 1};
    if (r < nQ) {                   // ---- Q: RMSNorm -> QKV -> + bias -> RoPE -> q granules / key, value granules + cache rows
        const int head = r >> 1, half = r & 1;
        const int pos = a.state[CV2_ST_POS];
        const int skip = *a.dbg_skip;
        const int f = half * 16 + ((tid >> 4) & 1) * 32 + (tid & 15);
        const float bias = L.bqkv[head * 64 + f];
        float c, sn;
        auto hook = [&]() { c = a.cosT[pos * 32 + (f & 31)]; sn = a.sinT[pos * 32 + (f & 31)]; };
        float v = row1_core<2, 4, 7, true, NT>(L.wqkv, head * 4 + half, 2, H / 32, H, 0, H / 32, xin, L.ln1, a.eps, smem, hook) + bias;
        const float vp = __shfl(v, (tid & 63) ^ 16);
        if (head < a.n_q + a.n_kv) v = (f < 32) ? (v * c - vp * sn) : (v * c + vp * sn);      // rotate-half RoPE on q and k heads
        CH_T(1);
        if (skip == ((layer << 16) | r) + 1) return;   // (test hook: a hand-off that never arrives -> the consumers' bounded waits, CV2_ST_ERR = 3)
        if (tid < 32) {
            if (head < a.n_q) G.store(gl + a.off_qg + head * 64 + f, v);
            else if (head < a.n_q + a.n_kv) {
                const int kvh = head - a.n_q;
                G.store(gl + a.off_kv + kvh * 64 + f, v);
                L.kc[((size_t)kvh * a.max_pos + pos) * 64 + f] = v;
            } else {
                const int kvh = head - a.n_q - a.n_kv;
                G.store(gl + a.off_kv + (a.n_kv + kvh) * 64 + f, v);
                L.vc[((size_t)kvh * a.max_pos + pos) * 64 + f] = v;
            }
        }
        CH_T(2);
        return;
    }
    r -= nQ;
    if (r < nA) {                   // ---- A: one 64-key tile of one kv head
        const int tile = r / a.n_kv, g = r - tile * a.n_kv;
        const int pos = a.state[CV2_ST_POS];
        if (tile * AT_TILE >= pos) return;              // no cached key in this tile: the consumer derives the live count from pos too
        if (layer > 0 && !G.spec) G.wait(gp + a.off_dg + H - 1, H, CH_NP);   // armed: the previous layer's down projection has published
        attn_role(G, L.kc + (size_t)g * a.max_pos * 64, L.vc + (size_t)g * a.max_pos * 64, pos, tile * AT_TILE, a.rep,
                  gl + a.off_qg + g * a.rep * 64, gl + a.off_ag + (unsigned)r * AT_GSTRIDE, smem, od);
        CH_T(2);
        return;
    }
    r -= nA;
    if (r < nO) {                   // ---- O: attention combine -> O projection -> + residual -> x_mid granules
        const int pos = a.state[CV2_ST_POS];
        OpAtt op{&G, gl + a.off_ag, a.n_kv, a.rep, (pos + AT_TILE - 1) / AT_TILE, od, &xin, r * 16, gl + a.off_qg, gl + a.off_kv};
        const float o = row1_core<1, 8, 4, false, NT>(L.wo, r, 0, a.NQ / 32, a.NQ, 0, a.NQ / 32, op, nullptr, 0.f, smem);
        CH_T(1);
        if (tid < 16) G.store(gl + r * 16 + tid, reinterpret_cast<const float*>(smem + R1_STAGE_BYTES(a.NQ / 32))[1200 + tid] + o);
        CH_T(2);
        return;
    }
    r -= nO;
    if (r < nGU) {                  // ---- GU: RMSNorm -> gate / up -> SiLU(g) * u -> h granules
        OpGran<4> op{&G, gl, gl + 15, 0, od};
        const float v = row1_core<2, 4, 7, true, NT>(L.wgu, r * 2, 1, H / 32, H, 0, H / 32, op, L.ln2, a.eps, smem);
        const float u = __shfl(v, (tid & 15) + 16);            // threads 0..15 hold gate, 16..31 up (wave 0)
        CH_T(1);
        if (tid < 16) G.store(gl + a.off_hg + r * 16 + tid, (v / (1.f + __expf(-v))) * u);
        CH_T(2);
        return;
    }
    r -= nGU;
    {                               // ---- D: down projection, K split CH_NP ways -> partial granules
        const int sp = r / nO, tile = r - sp * nO;
        const int KS = a.inter / 32;
        const int ks0 = (int)(((unsigned)KS * sp) / CH_NP), ks1 = (int)(((unsigned)KS * (sp + 1)) / CH_NP);
        OpGran<8> op{&G, gl + a.off_hg, gl + a.off_hg + ks0 * 32 + 15, 0, od};
        const float v = row1_core<1, 8, 10, false, NT>(L.wdown, tile, 0, KS, a.inter, ks0, ks1, op, nullptr, 0.f, smem);
        CH_T(1);
        if (tid < 16) G.store(gl + a.off_dg + sp * H + tile * 16 + tid, v);
        CH_T(2);
    }
}

// ------------------------------------------------------------------ k_step1: round-5 experiments on the one-row step (CV2_STEP1, default off)
// k_step's chain is x -> Q -> A -> O -> GU -> D -> x: five all-to-all hand-offs per layer, each a fabric round trip behind the slowest
// producer.  Two restructurings, both bit-identical to k_step (every sum keeps its order: t2_finish, chain.h) and both measured NOT to pay
// (profiles/r5_decode_step_experiments.txt); kept behind the switch as the record of the A/B, like CV2_PRE_FUSE:
//   QA   a block per (query head, 128-key tile) computes the head's 64 q values ITSELF (the 115 KB of W_q rows requested at dispatch like
//        every weight fragment) and goes straight on to the tile's scores: no Q -> A hand-off, no wait for the slowest of 14 q blocks.  The
//        key / value heads of the new token keep their Q-role blocks.  The tile's cache rows wait in LDS (att_dma_tile) -- as planes in
//        registers beside 14 weight fragments per wave the kernel needs 168 VGPRs.  372 us per step against 342: the fused block's serial
//        work costs what the hop did, and every live tile re-reads W_q.
//   GU2  a gate/up block covers two (gate, up) tile pairs -- 152 blocks instead of 304 -- so that more than one layer of weight requests is
//        resident: the down projection's blocks start 4 us earlier and no longer wait for their weights, the fatter gate/up blocks publish
//        as much later.  342.5 us against 342.4.
// Block order per layer: Q (QA: the 2 x 2 n_kv key / value blocks only), A (QA: ntiles x 16 slots, head fastest), O, GU, D.
#define QA_HEADS 16
#define QA_TILE_BYTES (AT_TILE * 64 * 4)
#define QA_ATT_BYTES ((16 * AT_QLD + 8 * 64 + 2 * 8 * 16) * 4)
__host__ __device__ constexpr size_t qa_smem_bytes(int nks) { return 2 * (size_t)QA_TILE_BYTES + ((size_t)t2_smem_bytes(nks) > (size_t)QA_ATT_BYTES ? (size_t)t2_smem_bytes(nks) : (size_t)QA_ATT_BYTES); }
template <bool QA, bool GU2>
__global__ __launch_bounds__(R1_THREADS) void k_step1(StepArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int gb = blockIdx.x;
    const int H = a.H;
    const int nKV = 2 * 2 * a.n_kv, nQ = QA ? nKV : 2 * (a.n_q + 2 * a.n_kv), nA = QA ? a.ntiles * QA_HEADS : a.ntiles * a.n_kv, nO = H / 16,
              nGU = GU2 ? a.inter / 32 : a.inter / 16;
    const int per = nQ + nA + nO + nGU + CH_NP * nO;
    const int layer = min(gb / per, a.n_layers);
    int r = gb - layer * per;
    const bool dbg = layer == a.dbg_layer;
    const int r_dbg = r, od = dbg ? r : -1; (void)r_dbg; (void)dbg;
    CH_T(0);
    Gran G;
    G.init(a.gran, a.gran_bytes, *a.epoch, a.err, false);
    const unsigned gl = (unsigned)min(layer, a.n_layers - 1) * a.gl;
    const unsigned gp = layer > 0 ? (unsigned)(layer - 1) * a.gl : 0u;
    if (layer >= a.n_layers) {      // head
        OpFold op{&G, gl, gl + a.off_dg, H, nullptr, -1, gl + a.off_dg + H - 1};
        const float out = row1_core<1, 8, 4, true, true>(a.wdec, r, 0, H / 32, H, 0, H / 32, op, a.final_norm, a.eps, smem);
        if (tid < 16) a.logits[r * 16 + tid] = out + a.bdec[r * 16 + tid];
        return;
    }
    const StepLayer L = step_layer(a.layers, layer);
    OpFold xin{&G, gp, gp + a.off_dg, H, layer == 0 ? a.xin : nullptr, od, gp + a.off_hg + a.inter - 1};
    if (r < nQ) {                   // ---- Q (QA: only the new token's key / value heads)
        const int head = (QA ? a.n_q : 0) + (r >> 1), half = r & 1;
        const int pos = a.state[CV2_ST_POS];
        const int skip = *a.dbg_skip;
        const int f = half * 16 + ((tid >> 4) & 1) * 32 + (tid & 15);
        const float bias = L.bqkv[head * 64 + f];
        float c, sn;
        auto hook = [&]() { c = a.cosT[pos * 32 + (f & 31)]; sn = a.sinT[pos * 32 + (f & 31)]; };
        float v = row1_core<2, 4, 7, true, true>(L.wqkv, head * 4 + half, 2, H / 32, H, 0, H / 32, xin, L.ln1, a.eps, smem, hook) + bias;
        const float vp = __shfl(v, (tid & 63) ^ 16);
        if (head < a.n_q + a.n_kv) v = (f < 32) ? (v * c - vp * sn) : (v * c + vp * sn);
        CH_T(1);
        if (skip == ((layer << 16) | r) + 1) return;
        if (tid < 32) {
            if (head < a.n_q) G.store(gl + a.off_qg + head * 64 + f, v);
            else if (head < a.n_q + a.n_kv) {
                const int kvh = head - a.n_q;
                G.store(gl + a.off_kv + kvh * 64 + f, v);
                L.kc[((size_t)kvh * a.max_pos + pos) * 64 + f] = v;
            } else {
                const int kvh = head - a.n_q - a.n_kv;
                G.store(gl + a.off_kv + (a.n_kv + kvh) * 64 + f, v);
                L.vc[((size_t)kvh * a.max_pos + pos) * 64 + f] = v;
            }
        }
        CH_T(2);
        return;
    }
    r -= nQ;
    if (r < nA) {
        const int pos = a.state[CV2_ST_POS];
        if constexpr (QA) {         // ---- QA: query head `head` against the 128-key tile `tile` of its kv head
            // QA_HEADS (16) block slots per tile, the last ones empty: the blocks of one head (one per live tile) are 16 apart in the grid, i.e.
            // on ONE XCD (consecutive block ids go round the 8 XCDs) -- they all read the head's 115 KB of W_q rows, the first from HBM, the
            // others from that XCD's L2 (plain loads; tile-major with n_q slots and non-temporal loads the step took 398 us against 342:
            // +4.8 MB of weight traffic per layer)
            const int tile = r / QA_HEADS, head = r - tile * QA_HEADS;
            if (head >= a.n_q || (tile > 0 && tile * AT_TILE >= pos)) return;
            const int g = head / a.rep, hh = head - g * a.rep;
            // LDS: [key rows 32 K][value rows 32 K][work: the q computation's stage / exchange / reduction, then the attention's q rows / partials]
            char* kl = smem; char* vl = smem + QA_TILE_BYTES; char* wk = smem + 2 * QA_TILE_BYTES;
            T2W<7> w;
            t2_issue<7, false>(L.wqkv, head * 4, H / 32, w);
            const float c = a.cosT[pos * 32 + (tid & 31)], sn = a.sinT[pos * 32 + (tid & 31)];
            const float bias = L.bqkv[head * 64 + (tid & 63)];
            const bool keys = tile * AT_TILE < pos;                          // (tile 0 at position 0: q only)
            if (keys) att_dma_tile(L.kc + (size_t)g * a.max_pos * 64, L.vc + (size_t)g * a.max_pos * 64, tile * AT_TILE, kl, vl);
            float v = t2_finish<7, true>(w, H / 32, H, xin, L.ln1, a.eps, wk) + bias;
            const float vp = __shfl(v, (tid & 63) ^ 32);                     // rotate-half partner: feature f ^ 32 sits in lane f ^ 32 of wave 0
            {   // separately rounded products and sum, as the other forms' RoPE compiles (v_mul, v_mul, v_add / v_sub; here hipcc would
                // contract the sum into one fma: 1-ulp differences in a fifth of the q values)
                float vc = v * c, ps = vp * sn;
                asm volatile("" : "+v"(vc), "+v"(ps));                       // (the products leave the expression: nothing to contract)
                v = (tid & 32) ? vc + ps : vc - ps;
            }
            CH_T(1);
            __syncthreads();                                                 // the q computation's LDS is free: it becomes the attention's
            float* qs = reinterpret_cast<float*>(wk);
            for (int e = tid; e < 16 * AT_QLD; e += R1_THREADS) qs[e] = e < 64 ? v : 0.f;      // q row 0 (thread f < 64 holds feature f), rows 1 .. 15 zero
            if (tid < 64 && tile == 0) G.store(gl + a.off_qg + head * 64 + tid, v);             // the O role's new-token partial needs q too
            if (!keys) return;
            AttTile T;
            {
                const int w8 = __builtin_amdgcn_readfirstlane(tid >> 6);
                T.n = max(0, min(16, pos - (tile * AT_TILE + 16 * w8)));
            }
            const unsigned og = gl + a.off_ag + (unsigned)(tile * a.n_kv + g) * AT_GSTRIDE;
            att_compute<1, true>(G, T, 1, og + hh * 64, og + a.rep * 64 + hh * 2, wk, od, kl, vl);
        } else {                    // ---- A: one tile of one kv head, all its query heads (k_step's role)
            const int tile = r / a.n_kv, g = r - tile * a.n_kv;
            if (tile * AT_TILE >= pos) return;
            if (layer > 0) G.wait(gp + a.off_dg + H - 1, H, CH_NP);
            attn_role(G, L.kc + (size_t)g * a.max_pos * 64, L.vc + (size_t)g * a.max_pos * 64, pos, tile * AT_TILE, a.rep,
                      gl + a.off_qg + g * a.rep * 64, gl + a.off_ag + (unsigned)r * AT_GSTRIDE, smem, od);
        }
        CH_T(2);
        return;
    }
    r -= nA;
    if (r < nO) {                   // ---- O
        const int pos = a.state[CV2_ST_POS];
        OpAtt op{&G, gl + a.off_ag, a.n_kv, a.rep, (pos + AT_TILE - 1) / AT_TILE, od, &xin, r * 16, gl + a.off_qg, gl + a.off_kv};
        const float o = row1_core<1, 8, 4, false, true>(L.wo, r, 0, a.NQ / 32, a.NQ, 0, a.NQ / 32, op, nullptr, 0.f, smem);
        CH_T(1);
        if (tid < 16) G.store(gl + r * 16 + tid, reinterpret_cast<const float*>(smem + R1_STAGE_BYTES(a.NQ / 32))[1200 + tid] + o);
        CH_T(2);
        return;
    }
    r -= nO;
    if (r < nGU) {                  // ---- GU
        OpGran<4> op{&G, gl, gl + 15, 0, od};
        if constexpr (GU2) {        // tiles 4 r .. 4 r + 3 = (gate, up) of hidden columns [32 r, 32 r + 16) and [32 r + 16, 32 r + 32)
            T2W<7> w;
            t2_issue<7, true>(L.wgu, r * 4, H / 32, w);
            const float v = t2_finish<7, true>(w, H / 32, H, op, L.ln2, a.eps, smem);
            const float u = __shfl(v, (tid & 63) + 16);            // wave 0: lanes 0..15 gate a, 16..31 up a, 32..47 gate b, 48..63 up b
            CH_T(1);
            if (tid < 64 && (tid & 16) == 0) G.store(gl + a.off_hg + r * 32 + (tid >> 5) * 16 + (tid & 15), (v / (1.f + __expf(-v))) * u);
        } else {
            const float v = row1_core<2, 4, 7, true, true>(L.wgu, r * 2, 1, H / 32, H, 0, H / 32, op, L.ln2, a.eps, smem);
            const float u = __shfl(v, (tid & 15) + 16);
            CH_T(1);
            if (tid < 16) G.store(gl + a.off_hg + r * 16 + tid, (v / (1.f + __expf(-v))) * u);
        }
        CH_T(2);
        return;
    }
    r -= nGU;
    {                               // ---- D
        const int sp = r / nO, tile = r - sp * nO;
        const int KS = a.inter / 32;
        const int ks0 = (int)(((unsigned)KS * sp) / CH_NP), ks1 = (int)(((unsigned)KS * (sp + 1)) / CH_NP);
        OpGran<8> op{&G, gl + a.off_hg, gl + a.off_hg + ks0 * 32 + 15, 0, od};
        const float v = row1_core<1, 8, 10, false, true>(L.wdown, tile, 0, KS, a.inter, ks0, ks1, op, nullptr, 0.f, smem);
        CH_T(1);
        if (tid < 16) G.store(gl + a.off_dg + sp * H + tile * 16 + tid, v);
        CH_T(2);
    }
}

// ------------------------------------------------------------------ k_step2: 2 .. CH_MAX_ROWS rows in one launch, two rows per block
// The rows are taken in PAIRS: a pair is one chain of k_step's blocks in which every GEMV block serves both rows at once -- the rows are
// columns 0 and 1 of the MFMA's B operand (row2_core, chain.h), the block's weight fragments are fetched once, the two rows' operands are
// gathered by the two halves of the block side by side; hand-off buffers, slot (state record, KV cache, pending input) and epilogue are
// per row.  The attention role has no weights to share: its tiles stay one block per (row, tile, kv head).  ceil(R / 2) pairs are
// interleaved in the grid like the rows of k_step<true> (an odd row count: the last pair's second column repeats its first row and is
// not published).  Per row every sum runs in k_step's order, so the ids do not depend on which kernel served a row.
template <bool NT>
__global__ __launch_bounds__(R1_THREADS) void k_step2(StepArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, gb = blockIdx.x, R = a.n_rows, P = (R + 1) >> 1, H = a.H;
    const int nQ = 2 * (a.n_q + 2 * a.n_kv), nA1 = a.ntiles * a.n_kv, nA = 2 * nA1, nO = H / 16, nGU = a.inter / 16;
    const int per = a.per + nA1;
    const int lb = per * P;
    const int layer = min(gb / lb, a.n_layers);
    int r, chain;
    {
        const int q = gb - layer * lb;
        const int nb = layer < a.n_layers ? per : a.head_blocks, full = nb >> 3, g8 = q / (8 * P);
        if (g8 < full) { const int rem = q - g8 * 8 * P; chain = rem >> 3; r = g8 * 8 + (rem & 7); }
        else { const int rem = q - full * 8 * P, m = nb - full * 8; chain = rem / m; r = full * 8 + rem - chain * m; }
    }
    const bool two = 2 * chain + 1 < R;
    const int row0 = 2 * chain, row1 = two ? row0 + 1 : row0;
    const int slot0 = a.row_slots ? a.row_slots[row0] : row0, slot1 = a.row_slots ? a.row_slots[row1] : row1;
    const unsigned go0 = (unsigned)row0 * a.row_gran, go1 = (unsigned)row1 * a.row_gran;
    Gran G;
    G.init(a.gran, a.gran_bytes * (unsigned)R, *a.epoch, a.err + slot0 * ST, a.spec != 0);
    G.set_err2(a.err + slot1 * ST);
    const unsigned gl = (unsigned)min(layer, a.n_layers - 1) * a.gl;
    const unsigned gp = layer > 0 ? (unsigned)(layer - 1) * a.gl : 0u;
    const int c32 = (tid >> 5) & 1, c16 = (tid >> 4) & 1;          // the row of an epilogue thread: 32 (NWR = 2) / 16 (NWR = 1) features per row
    if (layer >= a.n_layers) {      // head
        OpFold op0{&G, go0 + gl, go0 + gl + a.off_dg, H, nullptr, -1, go0 + gl + a.off_dg + H - 1};
        OpFold op1{&G, go1 + gl, go1 + gl + a.off_dg, H, nullptr, -1, go1 + gl + a.off_dg + H - 1};
        const float out = row2_core<1, 8, 4, true, true, NT>(a.wdec, r, 0, H / 32, H, 0, H / 32, op0, op1, a.final_norm, a.eps, smem);
        if (tid < 32 && (c16 == 0 || two)) a.logits[(size_t)(c16 ? row1 : row0) * a.ldl + r * 16 + (tid & 15)] = out + a.bdec[r * 16 + (tid & 15)];
        return;
    }
    const StepLayer L = step_layer(a.layers, layer);
    OpFold xin0{&G, go0 + gp, go0 + gp + a.off_dg, H, layer == 0 ? a.xin + (size_t)row0 * H : nullptr, -1, go0 + gp + a.off_hg + a.inter - 1};
    OpFold xin1{&G, go1 + gp, go1 + gp + a.off_dg, H, layer == 0 ? a.xin + (size_t)row1 * H : nullptr, -1, go1 + gp + a.off_hg + a.inter - 1};
    if (r < nQ) {                   // ---- Q
        const int head = r >> 1, half = r & 1;
        const int slot = c32 ? slot1 : slot0;
        const int pos = a.state[slot * ST + CV2_ST_POS];
        const int f = half * 16 + ((tid >> 4) & 1) * 32 + (tid & 15);
        const float bias = L.bqkv[head * 64 + f];
        float c, sn;
        auto hook = [&]() { c = a.cosT[pos * 32 + (f & 31)]; sn = a.sinT[pos * 32 + (f & 31)]; };
        float v = row2_core<2, 4, 7, true, true, NT>(L.wqkv, head * 4 + half, 2, H / 32, H, 0, H / 32, xin0, xin1, L.ln1, a.eps, smem, hook) + bias;
        const float vp = __shfl(v, (tid & 63) ^ 16);
        if (head < a.n_q + a.n_kv) v = (f < 32) ? (v * c - vp * sn) : (v * c + vp * sn);
        if (*a.dbg_skip == ((layer << 16) | r) + 1) return;    // (test hook, as in k_step: Q block r of this layer keeps its results to itself)
        if (tid < 64 && (c32 == 0 || two)) {
            const unsigned o = (c32 ? go1 : go0) + gl;
            if (head < a.n_q) G.store(o + a.off_qg + head * 64 + f, v);
            else if (head < a.n_q + a.n_kv) {
                const int kvh = head - a.n_q;
                G.store(o + a.off_kv + kvh * 64 + f, v);
                L.kc[(size_t)slot * a.kv_slot + ((size_t)kvh * a.max_pos + pos) * 64 + f] = v;
            } else {
                const int kvh = head - a.n_q - a.n_kv;
                G.store(o + a.off_kv + (a.n_kv + kvh) * 64 + f, v);
                L.vc[(size_t)slot * a.kv_slot + ((size_t)kvh * a.max_pos + pos) * 64 + f] = v;
            }
        }
        return;
    }
    r -= nQ;
    if (r < nA) {                   // ---- A: one tile of one kv head of one row
        const int c = r / nA1, rr = r - c * nA1;
        if (c == 1 && !two) return;
        const int tile = rr / a.n_kv, g = rr - tile * a.n_kv;
        const int slot = c ? slot1 : slot0;
        const int pos = a.state[slot * ST + CV2_ST_POS];
        if (tile * AT_TILE >= pos) return;
        if (c) G.err = G.err2;
        const unsigned o = c ? go1 : go0;
        if (layer > 0 && !G.spec) G.wait(o + gp + a.off_dg + H - 1, H, CH_NP);
        attn_role(G, L.kc + (size_t)slot * a.kv_slot + (size_t)g * a.max_pos * 64, L.vc + (size_t)slot * a.kv_slot + (size_t)g * a.max_pos * 64, pos,
                  tile * AT_TILE, a.rep, o + gl + a.off_qg + g * a.rep * 64, o + gl + a.off_ag + (unsigned)rr * AT_GSTRIDE, smem, -1);
        return;
    }
    r -= nA;
    if (r < nO) {                   // ---- O
        const int pos0 = a.state[slot0 * ST + CV2_ST_POS], pos1 = a.state[slot1 * ST + CV2_ST_POS];
        OpAtt op0{&G, go0 + gl + a.off_ag, a.n_kv, a.rep, (pos0 + AT_TILE - 1) / AT_TILE, -1, &xin0, r * 16, go0 + gl + a.off_qg, go0 + gl + a.off_kv, 0, 0};
        OpAtt op1{&G, go1 + gl + a.off_ag, a.n_kv, a.rep, (pos1 + AT_TILE - 1) / AT_TILE, -1, &xin1, r * 16, go1 + gl + a.off_qg, go1 + gl + a.off_kv, 256, 1};
        const float ov = row2_core<1, 8, 4, false, true, NT>(L.wo, r, 0, a.NQ / 32, a.NQ, 0, a.NQ / 32, op0, op1, nullptr, 0.f, smem);
        if (tid < 32 && (c16 == 0 || two))
            G.store((c16 ? go1 : go0) + gl + r * 16 + (tid & 15), reinterpret_cast<const float*>(smem + R2_STAGE_BYTES(a.NQ / 32))[1200 + c16 * 16 + (tid & 15)] + ov);
        return;
    }
    r -= nO;
    if (r < nGU) {                  // ---- GU
        OpGran<4> op0{&G, go0 + gl, go0 + gl + 15, 0, -1, 0, true};
        OpGran<4> op1{&G, go1 + gl, go1 + gl + 15, 0, -1, 1, true};
        const float v = row2_core<2, 4, 7, true, true, NT>(L.wgu, r * 2, 1, H / 32, H, 0, H / 32, op0, op1, L.ln2, a.eps, smem);
        const float u = __shfl(v, (tid & 15) + 16 + 32 * c32);     // threads 0..15 hold gate, 16..31 up of row 0; 32..47 / 48..63 of row 1 (wave 0)
        if (tid < 64 && (tid & 31) < 16 && (c32 == 0 || two)) G.store((c32 ? go1 : go0) + gl + a.off_hg + r * 16 + (tid & 15), (v / (1.f + __expf(-v))) * u);
        return;
    }
    r -= nGU;
    {                               // ---- D
        const int sp = r / nO, tile = r - sp * nO;
        const int KS = a.inter / 32;
        const int ks0 = (int)(((unsigned)KS * sp) / CH_NP), ks1 = (int)(((unsigned)KS * (sp + 1)) / CH_NP);
        OpGran<8> op0{&G, go0 + gl + a.off_hg, go0 + gl + a.off_hg + ks0 * 32 + 15, 0, -1, 0, true};
        OpGran<8> op1{&G, go1 + gl + a.off_hg, go1 + gl + a.off_hg + ks0 * 32 + 15, 0, -1, 1, true};     // (own arming word: no barrier separates the two fetches)
        const float v = row2_core<1, 8, 10, false, false, NT>(L.wdown, tile, 0, KS, a.inter, ks0, ks1, op0, op1, nullptr, 0.f, smem);
        if (tid < 32 && (c16 == 0 || two)) G.store((c16 ? go1 : go0) + gl + a.off_dg + sp * H + tile * 16 + (tid & 15), v);
    }
}

// ------------------------------------------------------------------ k_step4: rows in groups of FOUR, four MFMA columns per GEMV block
// As k_step2 with row4_core: one chain of blocks per four rows.  The Q / gate-up / head blocks serve the four rows with the block's
// halves taking two rows each (both rows' operand loads in flight together), the down projection takes (0, 1) then (2, 3); the O role
// keeps k_step2's two-row blocks (its provider merges attention tiles in several dependent sweeps: two blocks per chain, rows (0, 1)
// and (2, 3), the second finds W_o in L2); attention tiles: one block per (row, tile, kv head).  A chain whose last rows do not exist
// (row count not a multiple of 4) repeats its first row in the empty columns and does not publish them.
template <bool NT>
__global__ __launch_bounds__(R1_THREADS) void k_step4(StepArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, gb = blockIdx.x, R = a.n_rows, P = (R + 3) >> 2, H = a.H;
    const int nQ = 2 * (a.n_q + 2 * a.n_kv), nA1 = a.ntiles * a.n_kv, nA = 4 * nA1, nO1 = H / 16, nO = 2 * nO1, nGU = a.inter / 16;
    const int per = a.per + 3 * nA1 + nO1;
    const int lb = per * P;
    const int layer = min(gb / lb, a.n_layers);
    int r, chain;
    {
        const int q = gb - layer * lb;
        const int nb = layer < a.n_layers ? per : a.head_blocks, full = nb >> 3, g8 = q / (8 * P);
        if (g8 < full) { const int rem = q - g8 * 8 * P; chain = rem >> 3; r = g8 * 8 + (rem & 7); }
        else { const int rem = q - full * 8 * P, m = nb - full * 8; chain = rem / m; r = full * 8 + rem - chain * m; }
    }
    const int nv = min(4, R - 4 * chain);                        // rows of this chain that exist
    int row[4], slot[4]; unsigned go[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        row[c] = 4 * chain + (c < nv ? c : 0);
        slot[c] = a.row_slots ? a.row_slots[row[c]] : row[c];
        go[c] = (unsigned)row[c] * a.row_gran;
    }
    Gran G;
    G.init(a.gran, a.gran_bytes * (unsigned)R, *a.epoch, a.err + slot[0] * ST, a.spec != 0);
#pragma unroll
    for (int c = 0; c < 4; c++) G.errs[c] = a.err + slot[c] * ST;      // a block that gives up flags every row of its chain (rows >= nv repeat slot[0])
    const unsigned gl = (unsigned)min(layer, a.n_layers - 1) * a.gl;
    const unsigned gp = layer > 0 ? (unsigned)(layer - 1) * a.gl : 0u;
    const int c32 = (tid >> 5) & 3, c16 = (tid >> 4) & 3;          // the row of an epilogue thread (NWR = 2: 32 features per row, NWR = 1: 16)
    auto sel = [&](int c, auto& arr) { return c == 0 ? arr[0] : c == 1 ? arr[1] : c == 2 ? arr[2] : arr[3]; };
    if (layer >= a.n_layers) {      // head
        OpFold ops[4];
#pragma unroll
        for (int c = 0; c < 4; c++) ops[c] = OpFold{&G, go[c] + gl, go[c] + gl + a.off_dg, H, nullptr, -1, go[c] + gl + a.off_dg + H - 1};
        const float out = row4_core<1, 8, 4, true, true, NT>(a.wdec, r, 0, H / 32, H, 0, H / 32, ops, a.final_norm, a.eps, smem);
        if (tid < 64 && c16 < nv) a.logits[(size_t)sel(c16, row) * a.ldl + r * 16 + (tid & 15)] = out + a.bdec[r * 16 + (tid & 15)];
        return;
    }
    const StepLayer L = step_layer(a.layers, layer);
    OpFold xin[4];
#pragma unroll
    for (int c = 0; c < 4; c++)
        xin[c] = OpFold{&G, go[c] + gp, go[c] + gp + a.off_dg, H, layer == 0 ? a.xin + (size_t)row[c] * H : nullptr, -1, go[c] + gp + a.off_hg + a.inter - 1};
    if (r < nQ) {                   // ---- Q
        const int head = r >> 1, half = r & 1;
        const int sl = sel(c32, slot);
        const int pos = a.state[sl * ST + CV2_ST_POS];
        const int f = half * 16 + ((tid >> 4) & 1) * 32 + (tid & 15);
        const float bias = L.bqkv[head * 64 + f];
        float c, sn;
        auto hook = [&]() { c = a.cosT[pos * 32 + (f & 31)]; sn = a.sinT[pos * 32 + (f & 31)]; };
        float v = row4_core<2, 4, 7, true, true, NT>(L.wqkv, head * 4 + half, 2, H / 32, H, 0, H / 32, xin, L.ln1, a.eps, smem, hook) + bias;
        const float vp = __shfl(v, (tid & 63) ^ 16);
        if (head < a.n_q + a.n_kv) v = (f < 32) ? (v * c - vp * sn) : (v * c + vp * sn);
        if (*a.dbg_skip == ((layer << 16) | r) + 1) return;    // (test hook, as in k_step)
        if (tid < 128 && c32 < nv) {
            const unsigned o = sel(c32, go) + gl;
            if (head < a.n_q) G.store(o + a.off_qg + head * 64 + f, v);
            else if (head < a.n_q + a.n_kv) {
                const int kvh = head - a.n_q;
                G.store(o + a.off_kv + kvh * 64 + f, v);
                L.kc[(size_t)sl * a.kv_slot + ((size_t)kvh * a.max_pos + pos) * 64 + f] = v;
            } else {
                const int kvh = head - a.n_q - a.n_kv;
                G.store(o + a.off_kv + (a.n_kv + kvh) * 64 + f, v);
                L.vc[(size_t)sl * a.kv_slot + ((size_t)kvh * a.max_pos + pos) * 64 + f] = v;
            }
        }
        return;
    }
    r -= nQ;
    if (r < nA) {                   // ---- A: one tile of one kv head of one row
        const int c = r / nA1, rr = r - c * nA1;
        if (c >= nv) return;
        const int tile = rr / a.n_kv, g = rr - tile * a.n_kv;
        const int sl = sel(c, slot);
        const int pos = a.state[sl * ST + CV2_ST_POS];
        if (tile * AT_TILE >= pos) return;
        G.err = G.err2 = a.err + sl * ST;
        const unsigned o = sel(c, go);
        if (layer > 0 && !G.spec) G.wait(o + gp + a.off_dg + H - 1, H, CH_NP);
        attn_role(G, L.kc + (size_t)sl * a.kv_slot + (size_t)g * a.max_pos * 64, L.vc + (size_t)sl * a.kv_slot + (size_t)g * a.max_pos * 64, pos,
                  tile * AT_TILE, a.rep, o + gl + a.off_qg + g * a.rep * 64, o + gl + a.off_ag + (unsigned)rr * AT_GSTRIDE, smem, -1);
        return;
    }
    r -= nA;
    if (r < nO) {                   // ---- O: rows (0, 1) or (2, 3) of the chain, two per block as in k_step2
        const int pr = r / nO1, t = r - pr * nO1;
        const int ca = 2 * pr, cb = 2 * pr + 1;
        if (ca >= nv) return;
        const bool two = cb < nv;
        const int sa = sel(ca, slot), sb = sel(cb, slot);
        const unsigned ga = sel(ca, go), gb_ = sel(cb, go);
        G.err = a.err + sa * ST; G.err2 = a.err + sb * ST;
        const int pos0 = a.state[sa * ST + CV2_ST_POS], pos1 = a.state[sb * ST + CV2_ST_POS];
        OpFold& xa = ca == 0 ? xin[0] : xin[2];
        OpFold& xb = cb == 1 ? xin[1] : xin[3];
        OpAtt op0{&G, ga + gl + a.off_ag, a.n_kv, a.rep, (pos0 + AT_TILE - 1) / AT_TILE, -1, &xa, t * 16, ga + gl + a.off_qg, ga + gl + a.off_kv, 0, 0};
        OpAtt op1{&G, gb_ + gl + a.off_ag, a.n_kv, a.rep, (pos1 + AT_TILE - 1) / AT_TILE, -1, &xb, t * 16, gb_ + gl + a.off_qg, gb_ + gl + a.off_kv, 256, 1};
        const float ov = row2_core<1, 8, 4, false, true, NT>(L.wo, t, 0, a.NQ / 32, a.NQ, 0, a.NQ / 32, op0, op1, nullptr, 0.f, smem);
        const int e = (tid >> 4) & 1;
        if (tid < 32 && (e == 0 || two))
            G.store((e ? gb_ : ga) + gl + t * 16 + (tid & 15), reinterpret_cast<const float*>(smem + R2_STAGE_BYTES(a.NQ / 32))[1200 + e * 16 + (tid & 15)] + ov);
        return;
    }
    r -= nO;
    G.err2 = a.err + sel(nv - 1, slot) * ST;
    if (r < nGU) {                  // ---- GU
        OpGran<4> ops[4];
#pragma unroll
        for (int c = 0; c < 4; c++) ops[c] = OpGran<4>{&G, go[c] + gl, go[c] + gl + 15, 0, -1, c, true};
        const float v = row4_core<2, 4, 7, true, true, NT>(L.wgu, r * 2, 1, H / 32, H, 0, H / 32, ops, L.ln2, a.eps, smem);
        const float u = __shfl(v, (tid & 15) + 16 + 32 * ((tid >> 5) & 1));     // within a wave: rows 2 w, 2 w + 1; gate in lanes 0..15 / 32..47, up in 16..31 / 48..63
        if (tid < 128 && (tid & 31) < 16 && c32 < nv) G.store(sel(c32, go) + gl + a.off_hg + r * 16 + (tid & 15), (v / (1.f + __expf(-v))) * u);
        return;
    }
    r -= nGU;
    {                               // ---- D
        const int sp = r / nO1, tile = r - sp * nO1;
        const int KS = a.inter / 32;
        const int ks0 = (int)(((unsigned)KS * sp) / CH_NP), ks1 = (int)(((unsigned)KS * (sp + 1)) / CH_NP);
        OpGran<8> ops[4];
#pragma unroll
        for (int c = 0; c < 4; c++) ops[c] = OpGran<8>{&G, go[c] + gl + a.off_hg, go[c] + gl + a.off_hg + ks0 * 32 + 15, 0, -1, c, true};
        const float v = row4_core<1, 8, 10, false, false, NT>(L.wdown, tile, 0, KS, a.inter, ks0, ks1, ops, nullptr, 0.f, smem);
        if (tid < 64 && c16 < nv) G.store(sel(c16, go) + gl + a.off_dg + sp * H + tile * 16 + (tid & 15), v);
    }
}

// ------------------------------------------------------------------ k_sample
// Philox4x32-10, the same function as cv2amd/philox.py (counter = (seq, step, trial, 0), key = seed)
__device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                           uint32_t out[4]) {
#pragma unroll
    for (int i = 0; i < 10; i++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ double u53(uint32_t hi, uint32_t lo) {
    return (double)((((uint64_t)hi << 32) | lo) >> 11) * (1.0 / 9007199254740992.0);
}

struct SampleArgs {
    const float* logits; int ldl;               // [rows][ldl]
    int* state; int* out_tokens; int max_out;
    const float* speech_emb; float* x_next; int hidden;
    int vocab, eos, max_pos;
    int bi_speech;                              // bistream: speech tokens per text block (mix_ratio[1] = 15)
    float top_p; int top_k, win; float rep_thr; // ras_sampling constants (conf/cosyvoice2.yaml:33-37): nucleus mass / size, repetition window, win_size * tau_r
    int prefill_seq, row, prefill_pos;          // prefill: one block, reads logits row `row`, updates slot prefill_seq,
    unsigned* epoch;                            // hand-off epoch of k_chain: advanced once per launch (chain.h)
    const int* slots; float* x_rows;            // decode over live slots: block b serves slot slots[b] and also leaves the next input in row b
};                                              // whose next KV position becomes prefill_pos (= prompt length)
#define SM_T 1024
#define SM_W (SM_T / 64)
#define SM_TOPK 25
struct ArgMax { float v; int i; };
__device__ __forceinline__ ArgMax am_better(ArgMax a, ArgMax b) {     // larger value, ties -> lower index
    return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}
__device__ __forceinline__ ArgMax block_argmax(ArgMax x, ArgMax* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ArgMax y{__shfl_xor(x.v, o), __shfl_xor(x.i, o)};
        x = am_better(x, y);
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = x;
    __syncthreads();
    ArgMax r = sh[0];
#pragma unroll
    for (int w = 1; w < SM_W; w++) r = am_better(r, sh[w]);
    return r;
}
__device__ __forceinline__ float block_sum(float v, float* sh) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = 0.f;
#pragma unroll
    for (int w = 0; w < SM_W; w++) r += sh[w];
    return r;
}
// order-preserving float -> uint key (larger float = larger key; every finite or -inf float maps to a key > 0) and back
__device__ __forceinline__ unsigned fkey(float f) { const unsigned u = __builtin_bit_cast(unsigned, f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float fkey_inv(unsigned k) { return __builtin_bit_cast(float, (k & 0x80000000u) ? (k ^ 0x80000000u) : ~k); }
// wave-wide unsigned max through DPP: row_shr 1/2/4/8 leave each 16-lane row's max in its lane 15, row_bcast 15 / 31 carry it
// across rows into lane 63 (invalid source lanes contribute the identity 0); the result is read back wave-uniform
template <int CTRL, int RMASK>
__device__ __forceinline__ unsigned dpp_umax(unsigned v) {
    const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, RMASK, 0xf, false);
    return o > v ? o : v;
}
__device__ __forceinline__ unsigned wave_umax(unsigned v) {
    v = dpp_umax<0x111, 0xf>(v); v = dpp_umax<0x112, 0xf>(v); v = dpp_umax<0x114, 0xf>(v); v = dpp_umax<0x118, 0xf>(v);
    v = dpp_umax<0x142, 0xa>(v); v = dpp_umax<0x143, 0xc>(v);
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned row0_umax(unsigned v) {       // max over lanes 0..15 only
    v = dpp_umax<0x111, 0xf>(v); v = dpp_umax<0x112, 0xf>(v); v = dpp_umax<0x114, 0xf>(v); v = dpp_umax<0x118, 0xf>(v);
    return (unsigned)__builtin_amdgcn_readlane((int)v, 15);
}
__device__ __forceinline__ double shfl_up_f64(double v, int o) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __shfl_up((int)b, o), hi = __shfl_up((int)(b >> 32), o);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
#define SM_NE ((6592 + SM_T - 1) / SM_T)                                      // logits per thread (strided: tid + e * SM_T)
#define SM_CAP 1024                                                           // candidate buffer of the top-k selection
__device__ __forceinline__ unsigned block_umax(unsigned k, unsigned* sh) {    // sh[SM_W]; every thread gets the block maximum
    k = wave_umax(k);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = k;
    __syncthreads();
    unsigned r = sh[0];
#pragma unroll
    for (int w = 1; w < SM_W; w++) r = sh[w] > r ? sh[w] : r;
    return r;
}
__global__ __launch_bounds__(SM_T) void k_sample(SampleArgs a) {
    __shared__ float lp[6592];
    __shared__ ArgMax sam[SM_W];
    __shared__ float ssum[SM_W];
    __shared__ unsigned smax[SM_W];
    __shared__ double rd[SM_T];
    __shared__ float candp[SM_TOPK];
    __shared__ int candi[SM_TOPK];
    __shared__ int s_top, s_need, s_done, s_ncand, s_cnt;
    __shared__ unsigned s_thr;
    __shared__ unsigned wl_v[SM_W * SM_TOPK];
    __shared__ int wl_i[SM_W * SM_TOPK];
    __shared__ unsigned rowmax[SM_T / 16];
    __shared__ unsigned ck[SM_CAP];
    __shared__ int ci[SM_CAP];
    __shared__ double s_u2;
    const int tid = threadIdx.x;
    const int seq = a.prefill_seq >= 0 ? a.prefill_seq : (a.slots ? a.slots[blockIdx.x] : (int)blockIdx.x);
    const int row = a.prefill_seq >= 0 ? a.row : blockIdx.x;
    int* st = a.state + seq * ST;
    const int V = a.vocab;
    SK_STAMP_DECL;
    SK_STAMP(0);
    // the logits (written by the head GEMV just before) are requested first, then the state record (uniform address: one scalar
    // load, read once instead of field by field where it is used), then the repetition window: three round trips in flight together
    float lv[SM_NE];
    {
        const float* lg = a.logits + (size_t)row * a.ldl;
#pragma unroll
        for (int e = 0; e < SM_NE; e++) lv[e] = lg[min(tid + e * SM_T, V - 1)];
    }
    int sv[ST];
    {
        const int* __restrict__ stc = st;
#pragma unroll
        for (int i = 0; i < ST; i++) sv[i] = stc[i];
    }
    // the first trial's uniforms depend on the state record only (counter = (slot, step, 0), key = seed): one lane of wave 1 draws them
    // while the logits are still on their way -- the ten Philox rounds were ~500 cycles of the sampling lane's serial draw phase
    __shared__ double s_u0[2];
    if (tid == 64) {
        uint32_t rn[4];
        philox4x32((uint32_t)seq, (uint32_t)sv[CV2_ST_STEP], 0u, 0u, (uint32_t)sv[CV2_ST_SEED_LO], (uint32_t)sv[CV2_ST_SEED_HI], rn);
        s_u0[0] = u53(rn[0], rn[1]); s_u0[1] = u53(rn[2], rn[3]);
    }
    // CV2_ST_ERR == 3: a hand-off of this step's k_step timed out (chain.h) -- the logits are not this step's: nothing is drawn and
    // nothing committed (state, tokens and the pending input stay as they were; the epoch still advances), so the host can clear the
    // flag and run the step again on the launches
    const int done = sv[CV2_ST_DONE] | (sv[CV2_ST_ERR] == 3 ? 1 : 0);
    const int step = sv[CV2_ST_STEP];
    // repetition window (last 10 emitted tokens): lanes 0..9 of wave 0 hold one entry each; a serial loop of dependent global
    // loads in the sampling thread cost ~10 memory round trips per step
    const int nhist = min(sv[CV2_ST_NOUT], a.max_out);                      // (nout never passes max_out: the slot finishes there)
    int hv = -1;
    if (tid < a.win && nhist - a.win + tid >= 0) hv = a.out_tokens[(size_t)seq * a.max_out + nhist - a.win + tid];
    // bistream (inference_bistream, llm.py:721-834): bimode 1 = text still expected (EOS always re-drawn, the fill id eos + 2 stops
    // the slot until the host has fed the next text block and is FORCED every n_speech + 1 entries once it has appeared), 2 = final
    // decode (EOS ends the request); out_tokens then holds every drawn id incl. fill / EOS, as the reference's list does
    const int bimode = sv[CV2_ST_BIMODE];
    const bool forced_fill = !done && bimode == 1 && sv[CV2_ST_NEXTFILL] != -1 && sv[CV2_ST_NOUT] == sv[CV2_ST_NEXTFILL];
    if (forced_fill && tid == 0) s_top = a.eos + 2;
    if (!done && !forced_fill) {
        const bool ignore_eos = bimode == 1 ? true : (bimode == 2 ? false : step < sv[CV2_ST_MINLEN]);
        const bool force = sv[CV2_ST_FORCE] != 0;
        const int mode = sv[CV2_ST_MODE];
        // masks on logits == masks on logp (log_softmax is monotone):
        //   step 0 never EOS (inference_wrapper only, llm.py:693-694); forced-length mode never draws ids >= eos
#pragma unroll
        for (int e = 0; e < SM_NE; e++) {
            const int i = tid + e * SM_T;
            if (i >= V || (bimode == 0 && step == 0 && i == a.eos) || (force && i >= a.eos)) lv[e] = -INFINITY;
        }
        if (mode == 0) {
            // greedy: argmax with EOS excluded while ignore_eos; ties -> lowest id
            ArgMax best{-INFINITY, 0x7fffffff};
#pragma unroll
            for (int e = 0; e < SM_NE; e++) {
                const int i = tid + e * SM_T;
                if (i < V && !(ignore_eos && i == a.eos)) best = am_better(best, ArgMax{lv[e], i});
            }
            best = block_argmax(best, sam);
            if (tid == 0) s_top = best.i;
        } else {
            // RAS (utils/common.py:111-139): top-p 0.8 / top-k 25 nucleus, repetition window 10, tau 0.1
            const int lane = tid & 63, w = tid >> 6;
            unsigned tk = 0u;                                                 // this thread's largest key (0 = none: every real key is > 0)
#pragma unroll
            for (int e = 0; e < SM_NE; e++) if (tid + e * SM_T < V) { const unsigned k = fkey(lv[e]); tk = k > tk ? k : tk; }
            SK_STAMP(1);                                                      // logits arrived
            const float m = fkey_inv(block_umax(tk, smax));
            float sum = 0.f;
#pragma unroll
            for (int e = 0; e < SM_NE; e++) if (tid + e * SM_T < V) sum += __expf(lv[e] - m);
            const float lse = m + __logf(block_sum(sum, ssum));
            tk = 0u;
#pragma unroll
            for (int e = 0; e < SM_NE; e++) {                                 // log_softmax (llm.py:690)
                lv[e] -= lse;
                if (tid + e * SM_T < V) { lp[tid + e * SM_T] = lv[e]; const unsigned k = fkey(lv[e]); tk = k > tk ? k : tk; }
            }
            if (tid == 0) s_cnt = 0;
            SK_STAMP(2);                                                      // log-softmax
            if (mode == 2) { if (tid == 0) s_top = 0; goto sample_done; }       // diagnostic exit points (tools/dbg_sample.py): after log-softmax
            // nucleus candidates (common.py:120-134) = head of the stable descending order (value desc, id asc).
            // Selection by a bound instead of 25 extract-max rounds (phase stamps: 39k of the kernel's 56k cycles):
            //   (a) the maximum of every 16-lane row (112 logits) through DPP row operations: 64 distinct elements;
            //   (b) their 25th largest value T (rank by counting, one wave) bounds the global 25th largest from below, so every
            //       nucleus candidate has key >= T; typically ~40 logits do;
            //   (c) those are compacted into LDS and ranked by counting under the total order (key desc, id asc): rank r < 25 is
            //       candidate r.  More than SM_CAP survivors (massively tied logits) take the extract-max path below instead.
            {
                unsigned rk = tk;
                rk = dpp_umax<0x111, 0xf>(rk); rk = dpp_umax<0x112, 0xf>(rk); rk = dpp_umax<0x114, 0xf>(rk); rk = dpp_umax<0x118, 0xf>(rk);
                if ((lane & 15) == 15) rowmax[tid >> 4] = rk;
            }
            __syncthreads();
            if (w == 0) {
                const unsigned mine = rowmax[lane];
                int r = 0;
#pragma unroll
                for (int j = 0; j < SM_T / 16; j++) { const unsigned o = rowmax[j]; r += (o > mine || (o == mine && j < lane)) ? 1 : 0; }
                if (r == a.top_k - 1) s_thr = mine;
            }
            __syncthreads();
            const unsigned thr = s_thr;
#pragma unroll
            for (int e = 0; e < SM_NE; e++) {
                const int i = tid + e * SM_T;
                const unsigned k = fkey(lv[e]);
                if (i < V && k >= thr) {
                    const int slot = atomicAdd(&s_cnt, 1);
                    if (slot < SM_CAP) { ck[slot] = k; ci[slot] = i; }
                }
            }
            __syncthreads();
            const int ncnd = s_cnt;
            int ncand_r = 0;                                                  // nucleus size, kept by thread 0 (the only reader)
            SK_STAMP(3);                                                      // candidates compacted
            if (ncnd <= SM_CAP && mode != 5) {                                // mode 5 (tests): force the extract-max path
                if (tid < ncnd) {
                    const unsigned mk = ck[tid]; const int mi = ci[tid];
                    int r = 0;
                    for (int j0 = 0; j0 < ncnd; j0 += 8) {                     // 8 independent LDS reads per wait (slots >= ncnd: masked)
                        unsigned o[8]; int oi[8];
#pragma unroll
                        for (int u = 0; u < 8; u++) { o[u] = ck[(j0 + u) & (SM_CAP - 1)]; oi[u] = ci[(j0 + u) & (SM_CAP - 1)]; }
#pragma unroll
                        for (int u = 0; u < 8; u++) r += (j0 + u < ncnd && (o[u] > mk || (o[u] == mk && oi[u] < mi))) ? 1 : 0;
                    }
                    if (r < a.top_k) { candi[r] = mi; candp[r] = __expf(fkey_inv(mk)); }
                }
                if (tid >= ncnd && tid < SM_TOPK) { candi[tid] = 0; candp[tid] = 0.f; }   // vocabulary smaller than the nucleus
                // the candidate lists are written by threads < max(ncnd, 25) and read by thread 0 alone (here and in the draw below): with
                // at most 64 survivors -- the usual ~40 -- all of that is wave 0, whose LDS operations execute in order: no block barrier
                // (each cost the sampling lane the arrival skew of 16 waves; the block meets again at the draw's barrier)
                if (ncnd > 64) __syncthreads();
                else __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                if (tid == 0) {
                    float cp[SM_TOPK];
#pragma unroll
                    for (int c = 0; c < SM_TOPK; c++) cp[c] = candp[c];        // all reads in flight, then the serial fp32 sum of the reference
                    // the nucleus grows while the mass BEFORE a candidate is < top_p (common.py:127-131): the reference's running fp32 sum is
                    // computed once as a chain of 25 additions (the same additions in the same order: equal bits up to where it stops, unused
                    // beyond), and since it never decreases the size is a COUNT of independent comparisons instead of a 25-step chain of
                    // compare -> select -> add on one lane
                    int nc = 0; float before = 0.f;
#pragma unroll
                    for (int c = 0; c < SM_TOPK; c++) {
                        nc += (c < a.top_k && before < a.top_p) ? 1 : 0;
                        before += cp[c];
                    }
                    ncand_r = nc;
                    s_ncand = nc;
                }
            } else {
                // Two stages, one block barrier: every wave extracts the top 25 of its own 1/16 of the vocabulary, wave 0 merges the
                // 16 sorted lists.  Cross-lane maxima go through DPP row operations instead of LDS-routed shuffles.
                constexpr int NE = SM_NE;
                unsigned kk[NE]; int ki[NE];                                   // this lane's elements, sorted: key desc, id asc
#pragma unroll
                for (int e = 0; e < NE; e++) {
                    const int i = tid + e * SM_T;
                    kk[e] = i < V ? fkey(lv[e]) : 0u;                          // 0 = no element (every real key is > 0)
                    ki[e] = i;
                }
#pragma unroll
                for (int p = 0; p < NE - 1; p++)
#pragma unroll
                    for (int q = 0; q < NE - 1 - p; q++) {                     // ids increase with q: strict > keeps equal keys in id order
                        const bool sw = kk[q + 1] > kk[q];
                        const unsigned tk2 = sw ? kk[q] : kk[q + 1]; const int ti = sw ? ki[q] : ki[q + 1];
                        kk[q] = sw ? kk[q + 1] : kk[q]; ki[q] = sw ? ki[q + 1] : ki[q];
                        kk[q + 1] = tk2; ki[q + 1] = ti;
                    }
                for (int c = 0; c < SM_TOPK; c++) {
                    const unsigned M = wave_umax(kk[0]);
                    const unsigned W = wave_umax((kk[0] == M && M != 0u) ? ~(unsigned)ki[0] : 0u);
                    const int wi = (int)~W;
                    if (lane == 0) { wl_v[w * SM_TOPK + c] = M; wl_i[w * SM_TOPK + c] = wi; }
                    if (ki[0] == wi && kk[0] == M) {                           // the owner pops its head
#pragma unroll
                        for (int e = 0; e < NE - 1; e++) { kk[e] = kk[e + 1]; ki[e] = ki[e + 1]; }
                        kk[NE - 1] = 0u;
                    }
                }
                __syncthreads();
                if (w == 0) {
                    int ptr = 0;
                    unsigned hk = lane < SM_W ? wl_v[lane * SM_TOPK] : 0u;
                    int hi = lane < SM_W ? wl_i[lane * SM_TOPK] : 0x7fffffff;
                    for (int c = 0; c < SM_TOPK; c++) {
                        const unsigned M = row0_umax(hk);
                        const unsigned W = row0_umax((hk == M && M != 0u) ? ~(unsigned)hi : 0u);
                        const int wi = (int)~W;
                        if (lane == 0) { candi[c] = wi; candp[c] = __expf(fkey_inv(M)); }
                        if (lane < SM_W && hi == wi && hk == M) {
                            ptr++;
                            hk = ptr < SM_TOPK ? wl_v[lane * SM_TOPK + ptr] : 0u;
                            hi = ptr < SM_TOPK ? wl_i[lane * SM_TOPK + ptr] : 0x7fffffff;
                        }
                    }
                    if (lane == 0) {
                        int nc = 0; float cum = 0.f;
                        while (nc < a.top_k && cum < a.top_p) { cum += candp[nc]; nc++; }
                        ncand_r = nc;
                        s_ncand = nc;
                    }
                }
            }
            const int ncand = ncand_r;                                        // (thread 0's own count; nobody else uses it)
            SK_STAMP(4);                                                      // candidates ranked
            if (mode == 4) { if (tid == 0) s_top = 0; goto sample_done; }       // after the candidate selection
            const int chunk = (V + SM_T - 1) / SM_T;
            bool rd_ready = false;
            int trial = 0;
            for (;;) {
                if (tid < 64) {                                                // wave 0: lane 0 draws, all lanes check the window
                    int top_l = 0;
                    if (tid == 0) {
                        double u1;
                        if (trial == 0) { u1 = s_u0[0]; s_u2 = s_u0[1]; }      // (drawn at kernel entry; the barriers since then order the LDS write)
                        else {
                            uint32_t rn[4];
                            philox4x32((uint32_t)seq, (uint32_t)step, (uint32_t)trial, 0u, (uint32_t)sv[CV2_ST_SEED_LO], (uint32_t)sv[CV2_ST_SEED_HI], rn);
                            u1 = u53(rn[0], rn[1]);
                            s_u2 = u53(rn[2], rn[3]);
                        }
                        // nucleus draw: inverse cdf over the candidate probabilities (float64 running sums in the reference's order); the
                        // probabilities are fetched from LDS together, and both passes are straight-line selects: with a branch per candidate
                        // the two 25-step loops were most of this phase's 5 700 cycles
                        float cp[SM_TOPK];
#pragma unroll
                        for (int c = 0; c < SM_TOPK; c++) cp[c] = candp[c];
                        // ONE chain of 25 float64 additions gives every prefix sum P[c] (the reference's running sum: same additions, same
                        // order); the total is P[ncand - 1], and as the sums never decrease the pick -- the first c < ncand with P[c] > u1 * total --
                        // is the COUNT of the c < ncand with P[c] <= threshold: independent comparisons, no second chain
                        double P[SM_TOPK];
                        double run = 0.0;
#pragma unroll
                        for (int c = 0; c < SM_TOPK; c++) { run += (double)cp[c]; P[c] = run; }
                        double csum = 0.0;
#pragma unroll
                        for (int c = 0; c < SM_TOPK; c++) csum = c == ncand - 1 ? P[c] : csum;
                        const double thr = u1 * csum;
                        int below = 0;
#pragma unroll
                        for (int c = 0; c < SM_TOPK; c++) below += (c < ncand && !(P[c] > thr)) ? 1 : 0;
                        const int pick = below < ncand - 1 ? below : ncand - 1;
                        top_l = candi[pick];
                    }
                    const int top = __builtin_amdgcn_readfirstlane(top_l);     // lane 0 is the first active lane
                    const int rep = __popcll(__ballot(hv == top));             // hv = -1 where the window has no entry
                    if (tid == 0) {
                        s_top = top;
                        s_need = (float)rep >= a.rep_thr;                      // rep_num >= win_size * tau_r -> random_sampling over the full vocabulary
                        s_done = 0;
                        if (!s_need) {
                            if (!ignore_eos || top != a.eos) s_done = 1;
                            else if (trial + 1 > 100) { st[CV2_ST_ERR] = 1; s_done = 1; }
                        }
                    }
                }
                __syncthreads();
                if (s_need) {
                    if (!rd_ready) {                                           // full-vocab cdf, chunked: thread t owns ids [t*chunk, (t+1)*chunk)
                        double cs = 0.0;
                        for (int i = tid * chunk; i < min(V, (tid + 1) * chunk); i++) cs += (double)__expf(lp[i]);
                        rd[tid] = cs;
                        rd_ready = true;
                        __syncthreads();
                    }
                    if (tid < 64) {
                        double mine = 0.0;
                        for (int j = 0; j < SM_T / 64; j++) mine += rd[tid * (SM_T / 64) + j];
                        double incl = mine;
#pragma unroll
                        for (int o = 1; o < 64; o <<= 1) { const double up = shfl_up_f64(incl, o); if (tid >= o) incl += up; }
                        const double tot = __builtin_bit_cast(double, ((long long)__shfl((int)(__builtin_bit_cast(long long, incl) >> 32), 63) << 32) |
                                                                         (unsigned int)__shfl((int)__builtin_bit_cast(long long, incl), 63));
                        const double thr = s_u2 * tot;
                        const unsigned long long crossed = __ballot(incl > thr);
                        const int owner = crossed ? __ffsll((long long)crossed) - 1 : 63;
                        if (tid == owner) {
                            double run = incl - mine;
                            int ch = (owner + 1) * (SM_T / 64) - 1;
                            for (int j = 0; j < SM_T / 64; j++) {
                                const int c = owner * (SM_T / 64) + j;
                                if (run + rd[c] > thr) { ch = c; break; }
                                run += rd[c];
                            }
                            int top = min(V, (ch + 1) * chunk) - 1;
                            for (int i = ch * chunk; i < min(V, (ch + 1) * chunk); i++) { run += (double)__expf(lp[i]); if (run > thr) { top = i; break; } }
                            s_top = top;
                            if (!ignore_eos || top != a.eos) s_done = 1;
                            else if (trial + 1 > 100) { st[CV2_ST_ERR] = 1; s_done = 1; }
                        }
                    }
                    __syncthreads();
                }
                if (s_done) break;
                trial++;
                __syncthreads();
            }
        }
    }
sample_done:
    __syncthreads();
    SK_STAMP(5);                                                              // token drawn
    const int top = done ? sv[CV2_ST_LAST] : s_top;
    // next input embedding (llm.py:711, 719); harmless for finished slots
    for (int k = tid; k < a.hidden; k += SM_T) {
        const float e = a.speech_emb[(size_t)top * a.hidden + k];
        a.x_next[(size_t)seq * a.hidden + k] = e;
        if (a.x_rows) a.x_rows[(size_t)row * a.hidden + k] = e;
    }
    if (tid == 0 && !done) {
        int nout = sv[CV2_ST_NOUT];
        int fin = 0;
        const int nstep = step + 1;
        if (bimode != 0) {
            const int fill = a.eos + 2;
            if (top == fill) st[CV2_ST_NEXTFILL] = nout + a.bi_speech + 1;                // llm.py:802-804 (also what the forced entry advances to)
            if (nout < a.max_out) a.out_tokens[(size_t)seq * a.max_out + nout] = top;     // out_tokens.append(top_ids), fill and EOS included
            nout++;
            if (top >= a.eos) {
                fin = 1;
                if (bimode == 1 && top == fill) st[CV2_ST_WAIT] = 1;                     // llm.py:807-808: wait for text
                else if (!(bimode == 2 && top == a.eos)) st[CV2_ST_ERR] = 2;             // llm.py:809-810, 829-831: ValueError
            }
        } else {
        if (top == a.eos) fin = 1;                                        // llm.py:707-708
        else if (top < a.eos) {                                           // normal token: emit
            if (nout < a.max_out) a.out_tokens[(size_t)seq * a.max_out + nout] = top;
            nout++;
        }                                                                 // top > eos: fed back, not emitted (llm.py:712-714)
        if (nstep >= sv[CV2_ST_MAXLEN]) fin = 1;                          // for i in range(max_len)
        }
        if (nout >= a.max_out) fin = 1;                                   // output buffer full (the host clamps max_len to max_out, so
                                                                          // this only guards a state record written behind its back)
        if (st[CV2_ST_ERR]) fin = 1;                                      // (may have been raised above: read back)
        const int pos = a.prefill_seq >= 0 ? a.prefill_pos : sv[CV2_ST_POS] + 1;
        if (pos + 1 >= a.max_pos) fin = 1;
        st[CV2_ST_NOUT] = nout; st[CV2_ST_STEP] = nstep; st[CV2_ST_LAST] = top; st[CV2_ST_DONE] = fin;
        // a finished slot idles on its last position; a slot waiting for text (fill) moves on: the text block goes to the next one
        if (!fin || a.prefill_seq >= 0 || (bimode == 1 && st[CV2_ST_WAIT])) st[CV2_ST_POS] = pos;
    }
    if (blockIdx.x == 0 && tid == 0) *a.epoch += 1u;
    SK_STAMP(6);
    SK_STAMP_FLUSH;
}

// ------------------------------------------------------------------ batched prefill (rows of several prompts at once)
// The prompt rows of all new requests are packed into one [M][hidden] matrix and go through the bf16 MFMA GEMM of gemm.h
// with the operand split x = hi + lo (two MFMAs per tile, ~16 mantissa bits of x, the same arithmetic class as the skinny
// decode kernels), so the weights are streamed once per prefill instead of once per 32 rows.
__device__ __forceinline__ uint32_t pk2(float a, float b) { return (uint32_t)f2bf(a) | ((uint32_t)f2bf(b) << 16); }

// RMSNorm + hi/lo split: x fp32 [M][H] -> bf16 planes; one wave per row; rows >= M_valid -> 0
// parts != null: x += parts[0] + .. + parts[np - 1] first (the split-K partials of the previous layer's down projection, index order), written back
__global__ __launch_bounds__(256) void k_rms_split(float* x, const float* g, float eps, uint16_t* hi, uint16_t* lo, int M_valid, int Mp, int H,
                                                   const float* parts = nullptr, int np = 0, long pstride = 0) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= Mp) return;
    const int per = H / 64;                                   // 14 for hidden 896
    float v[16];
    float sq = 0.f;
    for (int i = 0; i < per; i++) {
        const size_t o = (size_t)row * H + lane * per + i;
        v[i] = row < M_valid ? x[o] : 0.f;
        if (parts && row < M_valid) {
            for (int p = 0; p < np; p++) v[i] += parts[p * pstride + o];
            x[o] = v[i];
        }
        sq += v[i] * v[i];
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)H + eps);
    for (int i = 0; i < per; i += 2) {
        const int c = lane * per + i;
        const float a0 = g[c] * (v[i] * rstd), a1 = g[c + 1] * (v[i + 1] * rstd);
        const uint16_t h0 = f2bf(a0), h1 = f2bf(a1);
        *reinterpret_cast<uint32_t*>(hi + (size_t)row * H + c) = (uint32_t)h0 | ((uint32_t)h1 << 16);
        *reinterpret_cast<uint32_t*>(lo + (size_t)row * H + c) = pk2(a0 - bf2f(h0), a1 - bf2f(h1));
    }
}

// bias was added by the GEMM epilogue: RoPE on q / k, q -> qbuf, k / v -> cache at (slot, position) of the row
struct RopeArgs {
    const float* qkv; int ld; const int* row_seq; const int* row_pos; const float* cosT; const float* sinT;
    float* q; float* kc; float* vc; int n_q, n_kv, max_pos, M;
};
__global__ __launch_bounds__(256) void k_rope_cache(RopeArgs a) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int nh = a.n_q + 2 * a.n_kv;
    if (idx >= (long)a.M * nh * 32) return;
    const int i = idx & 31, head = (idx >> 5) % nh, row = idx / (32L * nh);
    const int seq = a.row_seq[row], pos = a.row_pos[row];
    const float v0 = a.qkv[(size_t)row * a.ld + head * 64 + i], v1 = a.qkv[(size_t)row * a.ld + head * 64 + 32 + i];
    float o0 = v0, o1 = v1;
    if (head < a.n_q + a.n_kv) {
        const float c = a.cosT[pos * 32 + i], sn = a.sinT[pos * 32 + i];
        o0 = v0 * c - v1 * sn;
        o1 = v1 * c + v0 * sn;
    }
    float* dst;
    if (head < a.n_q) dst = a.q + (size_t)row * a.n_q * 64 + head * 64;
    else if (head < a.n_q + a.n_kv) dst = a.kc + (((size_t)seq * a.n_kv + (head - a.n_q)) * a.max_pos + pos) * 64;
    else dst = a.vc + (((size_t)seq * a.n_kv + (head - a.n_q - a.n_kv)) * a.max_pos + pos) * 64;
    dst[i] = o0; dst[32 + i] = o1;
}

// causal attention of the batched prefill: PFM_ROWS prompt rows x one GQA group over the cache; output as hi/lo bf16 planes [M][n_q*64]
struct PfAttnArgs {
    const float* q; const float* kc; const float* vc; uint16_t* hi; uint16_t* lo;
    const int* seq_row0; const int* seq_len; const int* seq_slot; const int* seq_pos0;
    int n_q, n_kv, max_pos;
};
// On the matrix cores at fp32 accuracy (three exact bf16 planes per operand, six products; attn_role above is the one-row form; the
// scalar fp32 kernel this replaced ran at 7 TFLOP/s: 68 us per layer at one 307-row prompt, 848 us at 32 prompts).  Block = PFM_ROWS prompt rows x one GQA group x one sequence, 4 waves; the 8 x rep <= 64 (row, head) pairs are the COLUMNS of
// S^T = K Q^T and of O^T = V^T P^T, so a pair's running (max, sum) and its rescale factor live on the lane that holds its column in both
// products and nothing has to be exchanged between them.  Wave w owns keys [16 w, 16 w + 16) of every 64-key tile: K rows and V rows go
// global -> registers -> planes (each value split once, next tile's loads in flight), q planes are read from LDS; no barrier inside the
// key loop; the four waves' (max, sum, O^T) are merged through LDS at the end.  68 us -> 9 us per layer at one 307-row prompt.
#define PFM_ROWS 8
#define PFM_QLD 72                                 // bf16 per q-plane row (64 + 8: 144-B stride, conflict-free ds_read_b128)
constexpr size_t pfm_smem_bytes() { return (size_t)3 * 64 * PFM_QLD * 2 + (size_t)3 * (64 * 65 + 2 * 64) * 4; }
__global__ __launch_bounds__(256) void k_attn_prefill_mfma(PfAttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint16_t* qp = reinterpret_cast<uint16_t*>(smem);                        // [3 planes][64 pairs][PFM_QLD]
    float* mg = reinterpret_cast<float*>(smem + (size_t)3 * 64 * PFM_QLD * 2);   // waves 1..3: [64 pairs][65] O^T values, then max[64], sum[64]
    const int rep = a.n_q / a.n_kv;
    const int qt = blockIdx.x, g = blockIdx.y, sq = blockIdx.z;
    const int len = a.seq_len[sq];
    if (qt * PFM_ROWS >= len) return;
    const int row0 = a.seq_row0[sq] + qt * PFM_ROWS, slot = a.seq_slot[sq], pos0 = a.seq_pos0[sq] + qt * PFM_ROWS;
    const int nrow = min(PFM_ROWS, len - qt * PFM_ROWS);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g4 = lane >> 4;
    const float* K = a.kc + ((size_t)slot * a.n_kv + g) * a.max_pos * 64;
    const float* V = a.vc + ((size_t)slot * a.n_kv + g) * a.max_pos * 64;
    const int kend = pos0 + nrow;                                // keys 0 .. kend - 1 are visible to some row
    const int ntile = (kend + 63) >> 6;
    // this wave's 16 keys of a tile, in operand order (attn_role): K[key c][dims 8 g4 .., 32 + 8 g4 ..], V[key 4 g4 + j][dim 16 t + c]
    f32x4 kr[4]; float vr[16];
    auto fetch = [&](int tile) {
        const int kb = tile * 64 + 16 * w;
        const float* kp = K + (size_t)(kb + c) * 64 + 8 * g4;
        kr[0] = *reinterpret_cast<const f32x4*>(kp);      kr[1] = *reinterpret_cast<const f32x4*>(kp + 4);
        kr[2] = *reinterpret_cast<const f32x4*>(kp + 32); kr[3] = *reinterpret_cast<const f32x4*>(kp + 36);
        const float* vp = V + (size_t)(kb + 4 * g4) * 64 + c;
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int j = 0; j < 4; j++) vr[4 * t + j] = vp[j * 64 + 16 * t];
    };
    fetch(0);
    {   // q planes: pair p = head * 8 + row; thread = (pair, 16-dim quarter)
        const int p = tid >> 2, d0 = (tid & 3) * 16, i = p & 7, hh = p >> 3;
        const bool live = i < nrow && hh < rep;
        const float* src = a.q + (size_t)(row0 + i) * a.n_q * 64 + (g * rep + hh) * 64 + d0;
#pragma unroll
        for (int e = 0; e < 2; e++) {
            f32x4 x0 = {0.f, 0.f, 0.f, 0.f}, x1 = x0;
            if (live) { x0 = *reinterpret_cast<const f32x4*>(src + 8 * e); x1 = *reinterpret_cast<const f32x4*>(src + 8 * e + 4); }
            bf16x8 p0, p1, p2;
            planes8(x0, x1, p0, p1, p2);
            *reinterpret_cast<bf16x8*>(qp + (size_t)(0 * 64 + p) * PFM_QLD + d0 + 8 * e) = p0;
            *reinterpret_cast<bf16x8*>(qp + (size_t)(1 * 64 + p) * PFM_QLD + d0 + 8 * e) = p1;
            *reinterpret_cast<bf16x8*>(qp + (size_t)(2 * 64 + p) * PFM_QLD + d0 + 8 * e) = p2;
        }
    }
    __syncthreads();
    f32x4 o[4][4];                                               // [pair tile u][dim tile t]: rows = dims 16 t + 4 g4 + reg, column = pair 16 u + c
    float mrun[4], lrun[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        mrun[u] = -INFINITY; lrun[u] = 0.f;
#pragma unroll
        for (int t = 0; t < 4; t++) o[u][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const int my_i = c & 7;                                      // the row of this lane's pairs (pair 16 u + c -> row (16 u + c) & 7 = c & 7)
    for (int tile = 0; tile < ntile; tile++) {
        const int kb = tile * 64 + 16 * w;
        bf16x8 ka[2][3];
        s16x4 va[4][3];
        planes8(kr[0], kr[1], ka[0][0], ka[0][1], ka[0][2]);
        planes8(kr[2], kr[3], ka[1][0], ka[1][1], ka[1][2]);
#pragma unroll
        for (int t = 0; t < 4; t++) {
            f32x4 v4;
#pragma unroll
            for (int j = 0; j < 4; j++) v4[j] = kb + 4 * g4 + j < kend ? vr[4 * t + j] : 0.f;       // rows past the prompt may hold anything
            planes4(v4, va[t][0], va[t][1], va[t][2]);
        }
        if (tile + 1 < ntile) fetch(tile + 1);                   // in flight during this tile's products
        if (kb >= kend) continue;                                // (wave-uniform) none of this wave's keys is visible
#pragma unroll
        for (int u = 0; u < 4; u++) {
            f32x4 sc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++) {
                bf16x8 qb[3];
#pragma unroll
                for (int pl = 0; pl < 3; pl++)
                    qb[pl] = *reinterpret_cast<const bf16x8*>(qp + (size_t)(pl * 64 + 16 * u + c) * PFM_QLD + 32 * s2 + 8 * g4);
                sc = mm6_32(sc, ka[s2], qb);
            }
            // sc[j]: key kb + 4 g4 + j against pair 16 u + c (row my_i): visible when key <= pos0 + row
            float mt = -INFINITY;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                sc[j] = (my_i < nrow && kb + 4 * g4 + j <= pos0 + my_i) ? sc[j] * 0.125f : -INFINITY;
                mt = fmaxf(mt, sc[j]);
            }
            mt = rows4_max(mt);
            const float mn = fmaxf(mrun[u], mt);
            const float mref = mn == -INFINITY ? 0.f : mn;
            const float scale = __expf(mrun[u] - mref);           // 0 on the first visible tile (mrun = -inf)
            f32x4 p;
            float lt = 0.f;
#pragma unroll
            for (int j = 0; j < 4; j++) { p[j] = __expf(sc[j] - mref); lt += p[j]; }
            lt = rows4_sum(lt);
            lrun[u] = lrun[u] * scale + lt;
            mrun[u] = mn;
            s16x4 pb[3];
            planes4(p, pb[0], pb[1], pb[2]);
#pragma unroll
            for (int t = 0; t < 4; t++) {
                f32x4 acc = o[u][t] * scale;
                acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(va[t][2], pb[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(va[t][1], pb[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(va[t][0], pb[2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(va[t][1], pb[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(va[t][0], pb[1], acc, 0, 0, 0);
                o[u][t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(va[t][0], pb[0], acc, 0, 0, 0);
            }
        }
    }
    // merge: waves 1..3 park (O^T, max, sum), wave 0 folds them in and stores the normalised rows as hi / lo bf16 planes
    if (w > 0) {
        float* d = mg + (size_t)(w - 1) * (64 * 65 + 128);
#pragma unroll
        for (int u = 0; u < 4; u++) {
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int j = 0; j < 4; j++) d[(16 * u + c) * 65 + 16 * t + 4 * g4 + j] = o[u][t][j];
            if (g4 == 0) { d[64 * 65 + 16 * u + c] = mrun[u]; d[64 * 65 + 64 + 16 * u + c] = lrun[u]; }
        }
    }
    __syncthreads();
    if (w > 0) return;
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int p = 16 * u + c, i = p & 7, hh = p >> 3;
        float M = mrun[u];
#pragma unroll
        for (int x = 0; x < 3; x++) M = fmaxf(M, mg[(size_t)x * (64 * 65 + 128) + 64 * 65 + p]);
        const float mref = M == -INFINITY ? 0.f : M;
        const float w0 = __expf(mrun[u] - mref);
        float l = lrun[u] * w0;
        f32x4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; t++) acc[t] = o[u][t] * w0;
#pragma unroll
        for (int x = 0; x < 3; x++) {
            const float* d = mg + (size_t)x * (64 * 65 + 128);
            const float wx = __expf(d[64 * 65 + p] - mref);
            l += d[64 * 65 + 64 + p] * wx;
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[t][j] += wx * d[p * 65 + 16 * t + 4 * g4 + j];
        }
        if (i < nrow && hh < rep) {
            const float inv = l > 0.f ? 1.f / l : 0.f;
            const size_t off = (size_t)(row0 + i) * a.n_q * 64 + (g * rep + hh) * 64;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                uint16_t hb[4], lb[4];
#pragma unroll
                for (int j = 0; j < 4; j++) { const float v = acc[t][j] * inv; hb[j] = f2bf(v); lb[j] = f2bf(v - bf2f(hb[j])); }
                *reinterpret_cast<uint2*>(a.hi + off + 16 * t + 4 * g4) = make_uint2(hb[0] | ((uint32_t)hb[1] << 16), hb[2] | ((uint32_t)hb[3] << 16));
                *reinterpret_cast<uint2*>(a.lo + off + 16 * t + 4 * g4) = make_uint2(lb[0] | ((uint32_t)lb[1] << 16), lb[2] | ((uint32_t)lb[3] << 16));
            }
        }
    }
}

// gate / up (16-row tiles interleaved as packed for the decode kernels) -> SiLU(g) * u -> hi/lo planes [M][inter]
__global__ __launch_bounds__(256) void k_swiglu_split(const float* gu, uint16_t* hi, uint16_t* lo, int M_valid, int Mp, int inter) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)Mp * inter) return;
    const int c = idx % inter, row = idx / inter;
    float v = 0.f;
    if (row < M_valid) {
        const int t = c >> 4, i = c & 15;
        const float gt = gu[(size_t)row * 2 * inter + (2 * t) * 16 + i], up = gu[(size_t)row * 2 * inter + (2 * t + 1) * 16 + i];
        v = (gt / (1.f + __expf(-gt))) * up;
    }
    const uint16_t hb = f2bf(v);
    hi[idx] = hb; lo[idx] = f2bf(v - bf2f(hb));
}
// x += parts[0] + .. + parts[np - 1] (after the last layer of a split-K prefill)
__global__ void k_fold_parts(float* x, const float* parts, int np, long pstride, long n) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    float v = x[idx];
    for (int p = 0; p < np; p++) v += parts[p * pstride + idx];
    x[idx] = v;
}
__global__ void k_gather_rows(const float* x, const int* rows, float* out, int n, int H) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx < n * H) out[idx] = x[(size_t)rows[idx / H] * H + idx % H];
}

// ------------------------------------------------------------------ host side
struct cv2_llm {
    cv2_llm_dims d;
    std::vector<cv2_llm_layer> layers;
    cv2_llm_weights w;
    cv2_llm_io io;
    // workspace carve
    float *kc, *vc;            // [layers][max_seqs][n_kv][max_pos][64]
    float *xa, *xb;            // residual stream ping-pong [32][hidden]
    float *xnext;              // [32][hidden] next-step input embeddings (k_sample), by slot
    float *xrows;              // [32][hidden] the same by ROW of a decode over live slots (cv2_llm_decode_rows)
    int *row_slots;            // [32] row -> slot of that decode
    float *q, *att, *o;        // [32][n_q*64], attention partials [nsplit][32][n_q*64], [32][hidden]
    float *att_ml;             // [nsplit][32][n_q][2]
    int *att_cnt;              // [32] non-empty splits per row
    float *attc;               // [32][n_q*64] combined attention output (many-row path)
    uint16_t *xp2;             // prepared gate/up operand written by the O projection's epilogue (xp is still being read by that launch)
    float *sqp;                // [32][hidden / 16] shares of the rows' sums of squares
    float *sqp2;               // the same for the next layer's QKV operand (left by the down projection's last arrivers)
    int *arrive_att, *arrive_down;   // arrival counters of the many-row path's fused combines: [32][n_kv], [hidden / 16]
    int pre_fuse;              // CV2_PRE_FUSE=1 at creation (run_layers_pre): bit 0 the attention's combine, bit 1 the down projection's
    uint16_t *xp, *xp_h;       // prepared operand planes [K/32][2][hi, lo][512]: k_prep output; SwiGLU output of k_gateup<2, true>
    int nsplit, keys_per_split;
    float *hbuf;               // [32][inter]
    float *parts;              // [SK_MAXNP][32][hidden] split-K partials of the down projection
    // batched prefill buffers (rows = pf_rows, 0 = disabled)
    int pf_rows;
    float *pf_x, *pf_qkv, *pf_q, *pf_gu, *pf_last;
    uint16_t *pf_hi, *pf_lo;
    float *pf_parts;           // [PF_SPLITK][PF_SPLIT_ROWS][hidden] split-K partials of the down projection (prefills of <= PF_SPLIT_ROWS rows)
    int* pf_int;               // row_seq[pf_rows], row_pos[pf_rows], seq tables 5 x 32
    u64* gran;                 // k_step hand-off granules [layers][gl]
    unsigned gran_bytes;
    StepArgs step;             // launch arguments of k_step (fixed at create)
    StepLayer* step_layers;    // device copy of the per-layer pointer table
    int step_blocks;
    unsigned* epoch;           // hand-off epoch (device), advanced by k_sample
    size_t chain_off, chain_bytes;   // the region cv2_llm_create zeroes
    bool use_chain;            // a one-row decode step is ONE launch, k_step (CV2_LLM_CHAIN=0: the five launches per layer)
    std::map<int, hipGraphExec_t> graphs;
    hipStream_t cap_stream;    // private stream used only to capture the decode-step graph (the caller's may be the null stream)
};

// One prompt's prefill: the down projection (N = hidden, K = inter) is 35 blocks of 64 x 128 with a 76-step K loop each -- 159 MFLOP on one
// CU while 220 idle.  Up to PF_SPLIT_ROWS rows its K is split PF_SPLITK ways as a batch of GEMMs into partial matrices, which the next
// layer's k_rms_split (or k_fold_parts after the last layer) adds to the residual stream in index order.
#define PF_SPLITK 4
#define PF_SPLIT_ROWS 512
static size_t carve(const cv2_llm_dims& d, cv2_llm* h, char* base) {
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return base ? base + o : (char*)nullptr; };
    const size_t cache = (size_t)d.layers * d.max_seqs * d.n_kv * d.max_pos * 64 * sizeof(float);
    char* p;
    p = take(cache); if (h) h->kc = (float*)p;
    p = take(cache); if (h) h->vc = (float*)p;
    (void)take((size_t)4 * AT_KB * 64 * sizeof(float));      // k_attn requests whole 64-key tiles before it knows the length
    p = take((size_t)32 * d.hidden * 4); if (h) h->xa = (float*)p;
    p = take((size_t)32 * d.hidden * 4); if (h) h->xb = (float*)p;
    p = take((size_t)32 * d.hidden * 4); if (h) h->xnext = (float*)p;
    p = take((size_t)32 * d.hidden * 4); if (h) h->xrows = (float*)p;
    p = take(32 * 4); if (h) h->row_slots = (int*)p;
    p = take((size_t)32 * d.n_q * 64 * 4); if (h) h->q = (float*)p;
    p = take((size_t)SK_MAXSPLIT * 32 * d.n_q * 64 * 4); if (h) h->att = (float*)p;
    p = take((size_t)SK_MAXSPLIT * 32 * d.n_q * 2 * 4); if (h) h->att_ml = (float*)p;
    p = take(32 * 4); if (h) h->att_cnt = (int*)p;
    p = take((size_t)32 * d.n_q * 64 * 4); if (h) h->attc = (float*)p;
    p = take((size_t)(d.inter / 32) * 2 * 2 * 1024); if (h) h->xp = (uint16_t*)p;
    p = take((size_t)(d.inter / 32) * 2 * 2 * 1024); if (h) h->xp_h = (uint16_t*)p;
    p = take((size_t)(d.inter / 32) * 2 * 2 * 1024); if (h) h->xp2 = (uint16_t*)p;
    p = take((size_t)32 * (d.hidden / 16) * 4); if (h) h->sqp = (float*)p;
    p = take((size_t)32 * (d.hidden / 16) * 4); if (h) h->sqp2 = (float*)p;
    p = take((size_t)32 * d.hidden * 4); if (h) h->o = (float*)p;
    p = take((size_t)32 * d.inter * 4); if (h) h->hbuf = (float*)p;
    p = take((size_t)SK_MAXNP * 32 * d.hidden * 4); if (h) h->parts = (float*)p;
    if (h) h->chain_off = off;
    p = take(256); if (h) h->epoch = (unsigned*)p;
    p = take((size_t)SK_ROWS_CAP * d.n_kv * 4); if (h) h->arrive_att = (int*)p;          // (zeroed at create, re-armed by their last arriver)
    p = take((size_t)(d.hidden / 16) * 4); if (h) h->arrive_down = (int*)p;
    {   // per layer: x_mid [H], down partials [CH_NP][H], q [NQ], new key / value rows [2 n_kv 64], attention partials
        // [tiles][n_kv][AT_GSTRIDE], h [inter]
        const size_t ntiles = (d.max_pos + AT_TILE - 1) / AT_TILE;
        const size_t gl = (size_t)d.hidden * (1 + CH_NP) + (size_t)d.n_q * 64 + (size_t)2 * d.n_kv * 64 + ntiles * d.n_kv * AT_GSTRIDE + d.inter;
        const int grows = d.max_seqs < CH_MAX_ROWS ? d.max_seqs : CH_MAX_ROWS;      // one set of hand-off buffers per row of a k_step launch
        p = take((size_t)grows * d.layers * gl * 8); if (h) { h->gran = (u64*)p; h->gran_bytes = (unsigned)((size_t)d.layers * gl * 8); }
        p = take((size_t)d.layers * sizeof(StepLayer)); if (h) h->step_layers = (StepLayer*)p;
    }
    if (h) h->chain_bytes = off - h->chain_off;
    const size_t R = (size_t)(d.max_prefill_rows > 0 ? (d.max_prefill_rows + 127) / 128 * 128 : 0);
    if (h) h->pf_rows = (int)R;
    if (R) {
        const size_t nqkv = (size_t)(d.n_q + 2 * d.n_kv) * 64;
        p = take(R * d.hidden * 4); if (h) h->pf_x = (float*)p;
        p = take(R * nqkv * 4); if (h) h->pf_qkv = (float*)p;
        p = take(R * d.n_q * 64 * 4); if (h) h->pf_q = (float*)p;
        p = take(R * 2 * d.inter * 4); if (h) h->pf_gu = (float*)p;
        p = take((size_t)32 * d.hidden * 4); if (h) h->pf_last = (float*)p;
        p = take(R * d.inter * 2); if (h) h->pf_hi = (uint16_t*)p;
        p = take(R * d.inter * 2); if (h) h->pf_lo = (uint16_t*)p;
        p = take((2 * R + 6 * 32) * 4); if (h) h->pf_int = (int*)p;
        p = take((size_t)PF_SPLITK * PF_SPLIT_ROWS * d.hidden * 4); if (h) h->pf_parts = (float*)p;
    }
    return off;
}

extern "C" size_t cv2_llm_workspace_bytes(const cv2_llm_dims* d) { return carve(*d, nullptr, nullptr); }

extern "C" int cv2_llm_create(const cv2_llm_dims* d, const cv2_llm_weights* w, const cv2_llm_io* io, void* ws, size_t ws_bytes,
                              cv2_llm** out) {
    CV2_CHECK(d && w && io && ws && out, "cv2_llm_create: null argument");
    CV2_CHECK(d->hidden % 32 == 0 && d->inter % 32 == 0 && d->vocab_pad % 16 == 0, "cv2_llm_create: hidden/inter must be multiples of 32");
    CV2_CHECK(d->hidden / 32 <= 32 && d->n_q * 64 / 32 <= 32, "cv2_llm_create: hidden and n_q*64 must be <= 1024 (k-steps per wave)");
    CV2_CHECK(d->inter / 32 <= SK_MAXNP * 4 * 10, "cv2_llm_create: inter must be <= 5120 (k-steps per wave of the down projection)");
    CV2_CHECK(d->n_q % d->n_kv == 0 && d->n_q / d->n_kv <= 8 && d->max_seqs >= 1 && d->max_seqs <= 32, "cv2_llm_create: bad head counts / max_seqs");
    CV2_CHECK(d->vocab <= 6592 && d->eos < d->vocab, "cv2_llm_create: vocab too large for the sampler");
    CV2_CHECK(ws_bytes >= cv2_llm_workspace_bytes(d), "cv2_llm_create: workspace too small (%zu < %zu)", ws_bytes, cv2_llm_workspace_bytes(d));
    CV2_CHECK(d->top_k == 0 || (d->top_k >= 1 && d->top_k <= SM_TOPK && d->win_size >= 0 && d->win_size <= 64 && d->top_p > 0.f),
              "cv2_llm_create: sampler constants out of range (top_k 1..%d, win_size 0..64)", SM_TOPK);
    cv2_llm* h = new cv2_llm();
    h->d = *d;
    if (h->d.top_k == 0) { h->d.top_p = 0.8f; h->d.top_k = 25; h->d.win_size = 10; h->d.tau_r = 0.1f; }   // conf/cosyvoice2.yaml:33-37
    h->layers.assign(w->layers, w->layers + d->layers);
    {
        const char* e = getenv("CV2_PRE_FUSE");
        h->pre_fuse = !e ? 0 : e[0] == '1' ? 3 : e[0] == 'a' ? 1 : e[0] == 'd' ? 2 : 0;
    }
    h->w = *w;
    h->w.layers = h->layers.data();
    h->io = *io;
    carve(*d, h, (char*)ws);
    h->keys_per_split = AT_KB;
    while ((d->max_pos + h->keys_per_split - 1) / h->keys_per_split > SK_MAXSPLIT) h->keys_per_split *= 2;
    h->nsplit = (d->max_pos + h->keys_per_split - 1) / h->keys_per_split;
    // the stream the decode graphs are captured on is the engine's own (CV2_LLM_CAP_PERSIST=0: one per capture; measured the same -- the
    // population of HIP streams decides how they share the process's hardware queues, see cv2_hift_create)
    h->cap_stream = nullptr;
    if (!(getenv("CV2_LLM_CAP_PERSIST") && getenv("CV2_LLM_CAP_PERSIST")[0] == '0') &&
        hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking) != hipSuccess) {
        delete h;
        return cv2_fail("cv2_llm_create: hipStreamCreateWithFlags failed");
    }
    {   // hand-off state of k_step: every granule tag 0, epoch 1 (tags are compared with the epoch, never 0)
        const char* e = getenv("CV2_LLM_CHAIN");
        const int ntiles = (d->max_pos + AT_TILE - 1) / AT_TILE, rep = d->n_q / d->n_kv;
        h->use_chain = !(e && e[0] == '0') && d->hidden % 32 == 0 && d->inter % (32 * CH_NP) == 0 && d->vocab_pad % 16 == 0 &&
                       d->hidden / 32 <= 4 * 7 && d->n_q * 64 / 32 <= 8 * 4 && d->inter / 32 / CH_NP <= 8 * 10 && d->inter / 8 / CH_NP <= R1_THREADS &&
                       d->n_q * 64 / 4 <= 240 && (d->hidden / 16) % 4 == 0 && rep * 64 + rep * 2 <= AT_GSTRIDE && rep * 64 <= 512 && ntiles * d->n_kv <= 64 &&
                       d->hidden % 2 == 0;
        const unsigned one = 1u;
        std::vector<StepLayer> tab(d->layers);
        const size_t cache_l = (size_t)d->max_seqs * d->n_kv * d->max_pos * 64;
        for (int l = 0; l < d->layers; l++) {
            const cv2_llm_layer& L = h->layers[l];
            tab[l] = StepLayer{L.wqkv, L.wo, L.wgu, L.wdown, L.bqkv, L.ln1, L.ln2, h->kc + l * cache_l, h->vc + l * cache_l};
        }
        if (hipMemset((char*)ws + h->chain_off, 0, h->chain_bytes) != hipSuccess ||
            hipMemcpy(h->epoch, &one, sizeof(one), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(h->step_layers, tab.data(), tab.size() * sizeof(StepLayer), hipMemcpyHostToDevice) != hipSuccess) {
            if (h->cap_stream) (void)hipStreamDestroy(h->cap_stream);
            delete h;
            return cv2_fail("cv2_llm_create: initialising the hand-off state failed");
        }
        StepArgs& a = h->step;
        a = StepArgs{};
        a.layers = h->step_layers; a.n_layers = d->layers;
        a.wdec = h->w.wdec; a.bdec = h->w.bdec; a.final_norm = h->w.final_norm; a.logits = h->io.logits;
        a.xin = h->xnext; a.state = h->io.state; a.cosT = h->w.rope_cos; a.sinT = h->w.rope_sin;
        a.gran = h->gran; a.gran_bytes = h->gran_bytes; a.epoch = h->epoch; a.err = h->io.state + CV2_ST_ERR;
        a.H = d->hidden; a.NQ = d->n_q * 64; a.inter = d->inter; a.n_q = d->n_q; a.n_kv = d->n_kv; a.rep = rep; a.max_pos = d->max_pos;
        a.ntiles = ntiles; a.eps = d->rms_eps;
        a.per = 2 * (d->n_q + 2 * d->n_kv) + ntiles * d->n_kv + d->hidden / 16 + d->inter / 16 + CH_NP * (d->hidden / 16);
        a.off_dg = d->hidden; a.off_qg = a.off_dg + CH_NP * d->hidden; a.off_kv = a.off_qg + d->n_q * 64;
        a.off_ag = a.off_kv + 2 * d->n_kv * 64; a.off_hg = a.off_ag + ntiles * d->n_kv * AT_GSTRIDE; a.gl = a.off_hg + d->inter;
        a.dbg_layer = -1;
        a.dbg_skip = (const int*)(h->epoch + 16);
        a.n_rows = 1; a.row_slots = nullptr; a.row_gran = (unsigned)((size_t)d->layers * a.gl);
        a.kv_slot = (long)d->n_kv * d->max_pos * 64; a.ldl = d->vocab_pad; a.head_blocks = d->vocab_pad / 16;
        h->step_blocks = d->layers * a.per + d->vocab_pad / 16;
    }
    *out = h;
    return 0;
}

extern "C" int cv2_llm_one_launch_step(const cv2_llm* h) { return h && h->use_chain ? 1 : 0; }

// test / diagnostic hook: device addresses of the decode workspaces (tools/dbg_chain_vals.py compares the one-launch step with the launches)
extern "C" int cv2_llm_debug_ptrs(cv2_llm* h, uint64_t* out) {
    CV2_CHECK(h && out, "cv2_llm_debug_ptrs: null argument");
    const StepArgs& a = h->step;
    const uint64_t v[16] = {(uint64_t)h->q, (uint64_t)h->att, (uint64_t)h->att_ml, (uint64_t)h->o, (uint64_t)h->hbuf, (uint64_t)h->parts, (uint64_t)h->xa,
                            (uint64_t)h->xb, (uint64_t)h->gran, a.gl, a.off_dg, a.off_qg, a.off_kv, a.off_ag, a.off_hg, (uint64_t)h->kc};
    for (int i = 0; i < 16; i++) out[i] = v[i];
    return 0;
}

// test hook: Q-role block `q_block` of layer `layer` of every following one-launch step keeps its results to itself (layer < 0: off).
// The blocks behind it run into their bounded waits: CV2_ST_ERR = 3, the step commits nothing (k_sample), the host re-runs it on the launches.
extern "C" int cv2_llm_debug_skip_publish(cv2_llm* h, int32_t layer, int32_t q_block) {
    CV2_CHECK(h, "cv2_llm_debug_skip_publish: null handle");
    const int v = layer < 0 ? 0 : ((layer << 16) | q_block) + 1;
    CV2_HIP(hipDeviceSynchronize());
    CV2_HIP(hipMemcpy((void*)(h->epoch + 16), &v, sizeof(v), hipMemcpyHostToDevice));
    return 0;
}

extern "C" int cv2_llm_destroy(cv2_llm* h) {
    if (!h) return 0;
    for (auto& g : h->graphs) (void)hipGraphExecDestroy(g.second);
    if (h->cap_stream) (void)hipStreamDestroy(h->cap_stream);
    delete h;
    return 0;
}

static int init_attrs_once();

// one pass of the layers + head over `rows` rows whose input embeddings sit in xin [rows][hidden]
// (xin is neither xa nor xb: the caller's prompt embeddings or xnext)
template <int NB>
static int run_layers(cv2_llm* h, int rows, const float* xin, RowMap rm, hipStream_t s) {
    const cv2_llm_dims& d = h->d;
    const int H = d.hidden, KSH = H / 32;
    const size_t cache_l = (size_t)d.max_seqs * d.n_kv * d.max_pos * 64;
    const float* xcur = xin;     // residual stream entering the layer, before the pending down-proj partials are folded in
    int np = 0;
    for (int l = 0; l < d.layers; l++) {
        const cv2_llm_layer& L = h->layers[l];
        float* x1 = (xcur == h->xa) ? h->xb : h->xa;          // x after folding the partials
        {
            QkvArgs a{};
            a.W = L.wqkv; a.bias = L.bqkv;
            a.X = SkinnyX{xcur, h->parts, np, L.ln1, d.rms_eps, x1};
            a.KS = KSH; a.rows = rows; a.K = H; a.n_q = d.n_q; a.n_kv = d.n_kv;
            a.cosT = h->w.rope_cos; a.sinT = h->w.rope_sin;
            a.q = h->q; a.kc = h->kc + l * cache_l; a.vc = h->vc + l * cache_l; a.max_pos = d.max_pos; a.rm = rm;
            const size_t sm = skinny_smem_bytes<NB, 2, 4>(KSH);
            STAMP_SET(l == 1 ? 0 : 100);
            hipLaunchKernelGGL(k_qkv<NB>, dim3(2 * (d.n_q + 2 * d.n_kv), 1), dim3(512), sm, s, a);
            STAMP_SET(-1);
        }
        {
            AttnArgs a{h->q, h->kc + l * cache_l, h->vc + l * cache_l, h->att, h->att_ml, h->att_cnt, d.n_q, d.n_kv, d.max_pos, h->nsplit, h->keys_per_split, rm, d.n_q / d.n_kv};
            launch_attn(a, rows, s);
            STAMP_SET(-1);
        }
        if (rows <= 4) {     // few rows: combine the key splits while loading the O-projection operand (saves a launch)
            StoreArgs a{};
            a.W = L.wo; a.bias = nullptr;
            a.X = SkinnyX{nullptr, h->att, h->nsplit, nullptr, 0.f, nullptr, h->att_ml, h->att_cnt};
            a.KS = d.n_q * 64 / 32; a.rows = rows; a.K = d.n_q * 64; a.N = H; a.out = h->o;
            const size_t sm = skinny_smem_bytes<NB, 1, 4>(a.KS);
            hipLaunchKernelGGL((k_store<NB, 8, true>), dim3(H / 16, 1), dim3(256), sm, s, a);
            STAMP_SET(-1);
        } else {             // many rows: one combine pass, then a plain O-projection
            CombArgs c{h->att, h->att_ml, h->att_cnt, h->attc, d.n_q * 64, h->nsplit};
            hipLaunchKernelGGL(k_attn_combine, dim3(rows), dim3(128), 0, s, c);
            StoreArgs a{};
            a.W = L.wo; a.bias = nullptr;
            a.X = SkinnyX{h->attc, nullptr, 0, nullptr, 0.f, nullptr};
            a.KS = d.n_q * 64 / 32; a.rows = rows; a.K = d.n_q * 64; a.N = H; a.out = h->o;
            const size_t sm = skinny_smem_bytes<NB, 1, 4>(a.KS);
            hipLaunchKernelGGL((k_store<NB, 8>), dim3(H / 16, 1), dim3(256), sm, s, a);
        }
        float* x2 = (x1 == h->xa) ? h->xb : h->xa;            // x_mid = x1 + o
        {
            GateUpArgs a{};
            a.W = L.wgu;
            a.X = SkinnyX{x1, h->o, 1, L.ln2, d.rms_eps, x2};   // o is one "partial" laid out [SK_ROWS_CAP][H]
            a.KS = KSH; a.rows = rows; a.K = H; a.inter = d.inter; a.h = h->hbuf;
            const size_t sm = skinny_smem_bytes<NB, 2, 4>(KSH);
            hipLaunchKernelGGL(k_gateup<NB>, dim3(d.inter / 16, 1), dim3(512), sm, s, a);
            STAMP_SET(-1);
        }
        {
            StoreArgs a{};
            a.W = L.wdown; a.bias = nullptr;
            a.X = SkinnyX{h->hbuf, nullptr, 0, nullptr, 0.f, nullptr};
            a.KS = d.inter / 32; a.rows = rows; a.K = d.inter; a.N = H; a.out = h->parts;
            const size_t sm = skinny_smem_bytes<NB, 1, 4>(cdiv(a.KS, SK_MAXNP) + 1);
            hipLaunchKernelGGL((k_store<NB, 10>), dim3(H / 16, SK_MAXNP), dim3(256), sm, s, a);
            STAMP_SET(-1);
        }
        xcur = x2;
        np = SK_MAXNP;
    }
    {
        StoreArgs a{};
        a.W = h->w.wdec; a.bias = h->w.bdec;
        a.X = SkinnyX{xcur, h->parts, np, h->w.final_norm, d.rms_eps, nullptr};
        a.KS = KSH; a.rows = rows; a.K = H; a.N = d.vocab_pad; a.out = h->io.logits;
        const size_t sm = skinny_smem_bytes<NB, 1, 4>(KSH);
        hipLaunchKernelGGL((k_store<NB, 8>), dim3(d.vocab_pad / 16, 1), dim3(256), sm, s, a);
    }
    CV2_LAUNCH_CHECK();
    return 0;
}

// Many-row variant (17..32 rows: batched decode, 32-row prefill chunks): every operand is folded / normalised / split once
// by k_prep and the weight-streaming kernels copy it, instead of every block redoing it for all rows.
static int run_layers_pre(cv2_llm* h, int rows, const float* xin, RowMap rm, hipStream_t s) {
    const cv2_llm_dims& d = h->d;
    const int H = d.hidden, KSH = H / 32, NQ = d.n_q * 64;
    const size_t cache_l = (size_t)d.max_seqs * d.n_kv * d.max_pos * 64;
    // Default: 7 launches per layer, both operand preparations (k_prep) as launches of their own.  CV2_PRE_FUSE=1 / a / d at engine creation
    // (diagnostics, kept for the A/B): the preparations done by the producers' last-arriving blocks, both / the attention's / the down
    // projection's.  Measured on MI355X at 32 rows: 1474 / 1478 / 1287 us per step against 1292 -- the attention's seam inside the launch
    // (drain the write-through stores, the counter's round trip, agent-coherent loads of the other splits' results, all behind the slowest
    // split) costs ~12 us where the launch boundary with a 4.8 us k_prep costs less; the down projection's is worth what its launch was
    // (profiles/r4_pre_fuse_ab.txt).
    const bool fuse = h->pre_fuse & 1, fuse_d = h->pre_fuse & 2;      // ('a' / 'd': the attention's / the down projection's alone)
    const float* xcur = xin;
    int np = 0;
    bool ready = false;                                  // the layer's QKV operand was left by the previous down projection (planes in xp, shares in sqp2, x1 written)
    auto prep = [&](const SkinnyX& X, int K, bool att) {
        if (att) hipLaunchKernelGGL(k_prep<true>, dim3(rows), dim3(256), 0, s, X, K, h->xp, (float*)nullptr);
        else hipLaunchKernelGGL(k_prep<false>, dim3(rows), dim3(256), 0, s, X, K, h->xp, h->sqp2);      // (one share per row: the sum of squares)
    };
    SkinnyX pre{};
    pre.pre = h->xp;
    for (int l = 0; l < d.layers; l++) {
        const cv2_llm_layer& L = h->layers[l];
        float* x1 = (xcur == h->xa) ? h->xb : h->xa;
        if (!ready) prep(SkinnyX{xcur, h->parts, np, L.ln1, d.rms_eps, x1}, H, false);
        {
            QkvArgs a{};
            a.W = L.wqkv; a.bias = L.bqkv; a.X = pre;
            a.KS = KSH; a.rows = rows; a.K = H; a.n_q = d.n_q; a.n_kv = d.n_kv;
            a.cosT = h->w.rope_cos; a.sinT = h->w.rope_sin;
            a.q = h->q; a.kc = h->kc + l * cache_l; a.vc = h->vc + l * cache_l; a.max_pos = d.max_pos; a.rm = rm;
            a.sq = h->sqp2; a.nsq = ready ? H / 16 : 1; a.eps = d.rms_eps;     // the rows' sums of squares: the down projection's shares, or k_prep's total
            { const size_t sm = skinny_smem_bytes<2, 2, 4>(KSH); hipLaunchKernelGGL((k_qkv<2, true>), dim3(2 * (d.n_q + 2 * d.n_kv), 1), dim3(512), sm, s, a); }
        }
        // CV2_ATT_ONE_SPLIT=1 (round 6 experiment, A/B): ONE block of four 64-key tile groups per (row, kv head) walks all the keys and leaves
        // the O projection's operand planes itself -- no key splits, no combine launch (k_prep<true>), 64 blocks instead of ~640 + 32
        static const bool one_split = getenv("CV2_ATT_ONE_SPLIT") && getenv("CV2_ATT_ONE_SPLIT")[0] == '1';
        if (one_split) {
            const int kps = (d.max_pos + 255) / 256 * 256;
            AttnArgs a{h->q, h->kc + l * cache_l, h->vc + l * cache_l, h->att, h->att_ml, h->att_cnt, d.n_q, d.n_kv, d.max_pos, 1, kps, rm, d.n_q / d.n_kv};
            a.arrive = h->arrive_att; a.pre = h->xp;
            hipLaunchKernelGGL(k_attn<4>, dim3(1, d.n_kv, rows), dim3(1024), 0, s, a);
        } else {   // attention; the last split of a (row, kv head) combines the splits and leaves the O projection's operand planes
            AttnArgs a{h->q, h->kc + l * cache_l, h->vc + l * cache_l, h->att, h->att_ml, h->att_cnt, d.n_q, d.n_kv, d.max_pos, h->nsplit, h->keys_per_split, rm, d.n_q / d.n_kv};
            if (fuse) { a.arrive = h->arrive_att; a.pre = h->xp; }
            launch_attn(a, rows, s);
        }
        if (!fuse && !one_split) prep(SkinnyX{nullptr, h->att, h->nsplit, nullptr, 0.f, nullptr, h->att_ml, h->att_cnt}, NQ, true);
        float* x2 = (x1 == h->xa) ? h->xb : h->xa;
        {   // O projection; its epilogue adds the residual and prepares the gate/up operand (no k_prep launch in between)
            StoreArgs a{};
            a.W = L.wo; a.X = pre; a.KS = NQ / 32; a.rows = rows; a.K = NQ; a.N = H; a.out = h->o;
            a.resid = x1; a.next_g = L.ln2; a.x_out = x2; a.next_pre = h->xp2; a.next_sq = h->sqp;
            { const size_t sm = skinny_smem_bytes<2, 1, 4>(a.KS); hipLaunchKernelGGL((k_store<2, 8, false, true, true>), dim3(H / 16, 1), dim3(256), sm, s, a); }
        }
        {
            GateUpArgs a{};
            SkinnyX pre2{};
            pre2.pre = h->xp2;
            a.W = L.wgu; a.X = pre2; a.KS = KSH; a.rows = rows; a.K = H; a.inter = d.inter; a.h = h->hbuf; a.hpre = h->xp_h;
            a.sq = h->sqp; a.nsq = H / 16; a.eps = d.rms_eps;
            if ((d.inter / 16) % 2 == 0 && d.inter / 16 > 256) {     // more tiles than CUs: two per block, one round
                const size_t sm = skinny_smem_bytes_keep<2, 2, 4>(KSH);
                hipLaunchKernelGGL((k_gateup<2, true, 2>), dim3(d.inter / 32, 1), dim3(512), sm, s, a);
            } else { const size_t sm = skinny_smem_bytes<2, 2, 4>(KSH); hipLaunchKernelGGL((k_gateup<2, true>), dim3(d.inter / 16, 1), dim3(512), sm, s, a); }
        }
        {   // down projection, split-K; before the last layer's, the last K-slice block of a column tile finishes the residual stream
            // and prepares the next layer's QKV operand (no k_prep launch in between)
            StoreArgs a{};
            SkinnyX preh{};
            preh.pre = h->xp_h;
            a.W = L.wdown; a.X = preh; a.KS = d.inter / 32; a.rows = rows; a.K = d.inter; a.N = H; a.out = h->parts;
            const size_t sm = skinny_smem_bytes<2, 1, 4>(cdiv(a.KS, SK_MAXNP) + 1);
            ready = fuse_d && l + 1 < d.layers;
            if (ready) {
                a.resid = x2; a.next_g = h->layers[l + 1].ln1; a.x_out = (x2 == h->xa) ? h->xb : h->xa; a.next_pre = h->xp; a.next_sq = h->sqp2;
                a.arrive = h->arrive_down;
                hipLaunchKernelGGL((k_store<2, 10, false, true, false, true>), dim3(H / 16, SK_MAXNP), dim3(256), sm, s, a);
            } else hipLaunchKernelGGL((k_store<2, 10, false, true>), dim3(H / 16, SK_MAXNP), dim3(256), sm, s, a);
        }
        xcur = x2;
        np = SK_MAXNP;
    }
    prep(SkinnyX{xcur, h->parts, np, h->w.final_norm, d.rms_eps, nullptr}, H, false);
    {
        StoreArgs a{};
        a.W = h->w.wdec; a.bias = h->w.bdec; a.X = pre;
        a.KS = KSH; a.rows = rows; a.K = H; a.N = d.vocab_pad; a.out = h->io.logits;
        a.sq = h->sqp2; a.eps = d.rms_eps;
        { const size_t sm = skinny_smem_bytes<2, 1, 4>(KSH); hipLaunchKernelGGL((k_store<2, 8, false, true>), dim3(d.vocab_pad / 16, 1), dim3(256), sm, s, a); }
    }
    CV2_LAUNCH_CHECK();
    return 0;
}

template <typename F>
static int set_smem(F f, size_t bytes) {
    CV2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(f), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return 0;
}
static int init_attrs_once() {
    static std::atomic<bool> done{false};   // idempotent: a concurrent first call repeats the attribute calls rather than launch before they are in place
    if (done) return 0;
    const size_t big = 159 * 1024;                 // (dynamic + the kernels' few static words <= 160 KB)
    if (set_smem(k_qkv<1>, big) || set_smem(k_qkv<2>, big) || set_smem(k_gateup<1>, big) || set_smem(k_gateup<2>, big) ||
        set_smem((k_store<1, 8>), big) || set_smem((k_store<2, 8>), big) || set_smem((k_store<1, 8, true>), big) || set_smem((k_store<2, 8, true>), big) || set_smem((k_store<1, 10>), big) ||
        set_smem((k_store<2, 10>), big) || set_smem((k_qkv<2, true>), big) || set_smem((k_gateup<2, true>), big) || set_smem((k_gateup<2, true, 2>), big) ||
        set_smem((k_store<2, 8, false, true>), big) || set_smem((k_store<2, 8, false, true, true>), big) || set_smem((k_store<2, 10, false, true>), big) ||
        set_smem((k_store<2, 10, false, true, false, true>), big) || set_smem((k_step1<true, true>), big) || set_smem((k_step1<true, false>), big))
        return -1;
    done = true;
    return 0;
}

static int launch_sample(cv2_llm* h, int nblocks, int prefill_seq, int row, int prefill_pos, hipStream_t s, bool mapped = false) {
    const cv2_llm_dims& d = h->d;
    SampleArgs a{h->io.logits, d.vocab_pad, h->io.state, h->io.out_tokens, d.max_out, h->w.speech_emb, h->xnext,
                 d.hidden, d.vocab, d.eos, d.max_pos, 15, d.top_p, d.top_k, d.win_size, (float)d.win_size * d.tau_r, prefill_seq, row, prefill_pos,
                 h->epoch, mapped ? h->row_slots : nullptr, mapped ? h->xrows : nullptr};
    hipLaunchKernelGGL(k_sample, dim3(nblocks), dim3(SM_T), 0, s, a);
    CV2_LAUNCH_CHECK();
    return 0;
}

extern "C" int cv2_llm_debug_sample(cv2_llm* h, int32_t n_rows, void* stream) {
    CV2_CHECK(h && n_rows >= 1 && n_rows <= h->d.max_seqs, "cv2_llm_debug_sample: bad arguments");
    return launch_sample(h, n_rows, -1, 0, 0, (hipStream_t)stream);
}

extern "C" int cv2_llm_prefill(cv2_llm* h, int32_t seq, const float* embeds, int32_t len, void* stream) {
    CV2_CHECK(h && embeds, "cv2_llm_prefill: null argument");
    CV2_CHECK(seq >= 0 && seq < h->d.max_seqs, "cv2_llm_prefill: bad slot %d", seq);
    CV2_CHECK(len >= 1 && len + 1 < h->d.max_pos, "cv2_llm_prefill: prompt length %d does not fit max_pos %d", len, h->d.max_pos);
    if (init_attrs_once()) return -1;
    hipStream_t s = (hipStream_t)stream;
    for (int p0 = 0; p0 < len; p0 += 32) {
        const int rows = len - p0 < 32 ? len - p0 : 32;
        const float* xin = embeds + (size_t)p0 * h->d.hidden;
        RowMap rm{h->io.state, 1, seq, p0};
        int rc = rows <= 16 ? run_layers<1>(h, rows, xin, rm, s) : run_layers_pre(h, rows, xin, rm, s);
        if (rc) return rc;
    }
    // draw step 0 from the last row's logits; the slot's next KV position becomes len
    return launch_sample(h, 1, seq, (len - 1) % 32, len, s);
}

// Continue slot `seq` with `len` more input rows at positions pos0 .. pos0 + len - 1 (bistream: the next text block, or the final
// [pending input, remaining text, task id]), then draw from the last row under the slot's state (bimode, next fill index).
extern "C" int cv2_llm_extend(cv2_llm* h, int32_t seq, const float* embeds, int32_t len, int32_t pos0, void* stream) {
    CV2_CHECK(h && embeds, "cv2_llm_extend: null argument");
    CV2_CHECK(seq >= 0 && seq < h->d.max_seqs, "cv2_llm_extend: bad slot %d", seq);
    CV2_CHECK(len >= 1 && pos0 >= 0 && pos0 + len + 1 < h->d.max_pos, "cv2_llm_extend: rows [%d, %d) do not fit max_pos %d", pos0, pos0 + len, h->d.max_pos);
    if (init_attrs_once()) return -1;
    hipStream_t s = (hipStream_t)stream;
    for (int p0 = 0; p0 < len; p0 += 32) {
        const int rows = len - p0 < 32 ? len - p0 : 32;
        RowMap rm{h->io.state, 1, seq, pos0 + p0};
        int rc = rows <= 16 ? run_layers<1>(h, rows, embeds + (size_t)p0 * h->d.hidden, rm, s) : run_layers_pre(h, rows, embeds + (size_t)p0 * h->d.hidden, rm, s);
        if (rc) return rc;
    }
    return launch_sample(h, 1, seq, (len - 1) % 32, pos0 + len, s);
}

// Input rows of several slots at once through the GEMM path: embeds = the slots' rows concatenated [sum(lens)][hidden] fp32; slot i's rows
// sit at KV positions pos0[i] .. pos0[i] + lens[i] - 1 (pos0 == nullptr: 0, step 0 of inference_wrapper) and attend to everything the
// slot's cache holds below them; one draw per slot from its last row.
static int rows_batch(cv2_llm* h, int32_t n, const int32_t* slots, const int32_t* lens, const int32_t* pos0, const float* embeds, void* stream) {
    CV2_CHECK(h && slots && lens && embeds && n >= 1 && n <= 32, "cv2_llm_prefill_batch: bad argument");
    CV2_CHECK(h->pf_rows > 0, "cv2_llm_prefill_batch: created with max_prefill_rows == 0");
    if (init_attrs_once()) return -1;
    static std::atomic<bool> once{false};    // idempotent attribute call: a second thread repeats it rather than launch before it is in place
    if (!once) {
        CV2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_prefill_mfma), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        once = true;
    }
    const cv2_llm_dims& d = h->d;
    hipStream_t s = (hipStream_t)stream;
    int M = 0, maxlen = 0;
    for (int i = 0; i < n; i++) {
        CV2_CHECK(slots[i] >= 0 && slots[i] < d.max_seqs, "cv2_llm_prefill_batch: bad slot %d", slots[i]);
        const int p0 = pos0 ? pos0[i] : 0;
        CV2_CHECK(lens[i] >= 1 && p0 >= 0 && p0 + lens[i] + 1 < d.max_pos, "cv2_llm_prefill_batch: rows [%d, %d) do not fit max_pos %d", p0, p0 + lens[i], d.max_pos);
        for (int j = 0; j < i; j++) CV2_CHECK(slots[j] != slots[i], "cv2_llm_prefill_batch: slot %d listed twice", slots[i]);
        M += lens[i]; maxlen = lens[i] > maxlen ? lens[i] : maxlen;
    }
    const int Mp = (M + 127) / 128 * 128;
    CV2_CHECK(Mp <= h->pf_rows, "cv2_llm_prefill_batch: %d prompt rows exceed max_prefill_rows %d", M, h->pf_rows);
    const int R = h->pf_rows;
    std::vector<int> host(2 * R + 6 * 32, 0);
    int* row_seq = host.data(); int* row_pos = row_seq + R;
    int* row0 = row_pos + R; int* slen = row0 + 32; int* sslot = slen + 32; int* spos0 = sslot + 32; int* lastrow = spos0 + 32;
    for (int i = 0, r = 0; i < n; i++) {
        const int p0 = pos0 ? pos0[i] : 0;
        row0[i] = r; slen[i] = lens[i]; sslot[i] = slots[i]; spos0[i] = p0; lastrow[i] = r + lens[i] - 1;
        for (int t = 0; t < lens[i]; t++, r++) { row_seq[r] = slots[i]; row_pos[r] = p0 + t; }
    }
    CV2_HIP(hipMemcpyAsync(h->pf_int, host.data(), host.size() * sizeof(int), hipMemcpyHostToDevice, s));
    CV2_HIP(hipMemcpyAsync(h->pf_x, embeds, (size_t)M * d.hidden * sizeof(float), hipMemcpyDeviceToDevice, s));
    CV2_HIP(hipStreamSynchronize(s));
    const int* d_row_seq = h->pf_int; const int* d_row_pos = d_row_seq + R;
    const int* d_row0 = d_row_pos + R; const int* d_len = d_row0 + 32; const int* d_slot = d_len + 32; const int* d_pos0 = d_slot + 32; const int* d_last = d_pos0 + 32;
    const int H = d.hidden, NQ = d.n_q * 64, NQKV = (d.n_q + 2 * d.n_kv) * 64, I = d.inter;
    CV2_CHECK(H % 128 == 0 || H % 64 == 0, "prefill: hidden %% 64");
    CV2_CHECK(NQKV % 128 == 0 && H % 128 == 0 && (2 * I) % 128 == 0 && NQ % 64 == 0 && I % 64 == 0, "cv2_llm_prefill_batch: dims must be multiples of 128 / 64 for the GEMM tiles");
    const size_t cache_l = (size_t)d.max_seqs * d.n_kv * d.max_pos * 64;
    static const bool splitk_env = !(getenv("CV2_PREFILL_SPLITK") && getenv("CV2_PREFILL_SPLITK")[0] == '0');      // A/B switch (diagnostics)
    const bool splitk = splitk_env && Mp <= PF_SPLIT_ROWS && I % (PF_SPLITK * 64) == 0;
    const long pstride = (long)Mp * H;
    bool pending = false;                                  // the previous layer's down projection left partials
    for (int l = 0; l < d.layers; l++) {
        const cv2_llm_layer& L = h->layers[l];
        hipLaunchKernelGGL(k_rms_split, dim3(Mp / 4), dim3(256), 0, s, h->pf_x, L.ln1, d.rms_eps, h->pf_hi, h->pf_lo, M, Mp, H,
                           pending ? (const float*)h->pf_parts : nullptr, PF_SPLITK, pstride);
        pending = false;
        {
            GemmArgs g = gemm_args(h->pf_hi, H, 0, L.wqkv, Mp, NQKV, H);
            g.A_lo = h->pf_lo; g.bias = L.bqkv; g.out_f32 = h->pf_qkv; g.ldo = NQKV;
            if (gemm_launch_cfg(g, 4, 1, true, s)) return -1;
        }
        {
            RopeArgs r{h->pf_qkv, NQKV, d_row_seq, d_row_pos, h->w.rope_cos, h->w.rope_sin, h->pf_q, h->kc + l * cache_l, h->vc + l * cache_l,
                       d.n_q, d.n_kv, d.max_pos, M};
            hipLaunchKernelGGL(k_rope_cache, dim3(((long)M * (d.n_q + 2 * d.n_kv) * 32 + 255) / 256), dim3(256), 0, s, r);
        }
        {
            PfAttnArgs a{h->pf_q, h->kc + l * cache_l, h->vc + l * cache_l, h->pf_hi, h->pf_lo, d_row0, d_len, d_slot, d_pos0, d.n_q, d.n_kv, d.max_pos};
            hipLaunchKernelGGL(k_attn_prefill_mfma, dim3((maxlen + PFM_ROWS - 1) / PFM_ROWS, d.n_kv, n), dim3(256), pfm_smem_bytes(), s, a);
        }
        {
            GemmArgs g = gemm_args(h->pf_hi, NQ, 0, L.wo, Mp, H, NQ);
            g.A_lo = h->pf_lo; g.res = h->pf_x; g.ldres = H; g.out_f32 = h->pf_x; g.ldo = H;
            if (gemm_launch_cfg(g, 4, 1, true, s)) return -1;
        }
        hipLaunchKernelGGL(k_rms_split, dim3(Mp / 4), dim3(256), 0, s, h->pf_x, L.ln2, d.rms_eps, h->pf_hi, h->pf_lo, M, Mp, H, (const float*)nullptr, 0, 0L);
        {
            GemmArgs g = gemm_args(h->pf_hi, H, 0, L.wgu, Mp, 2 * I, H);
            g.A_lo = h->pf_lo; g.out_f32 = h->pf_gu; g.ldo = 2 * I;
            if (gemm_launch_cfg(g, 4, 1, true, s)) return -1;
        }
        hipLaunchKernelGGL(k_swiglu_split, dim3(((long)Mp * I + 255) / 256), dim3(256), 0, s, (const float*)h->pf_gu, h->pf_hi, h->pf_lo, M, Mp, I);
        if (splitk) {
            GemmArgs g = gemm_args(h->pf_hi, I, 0, L.wdown, Mp, H, I / PF_SPLITK);
            g.A_lo = h->pf_lo; g.a_bstride = I / PF_SPLITK; g.w_ks = I / 32; g.w_bstride = (long)(I / PF_SPLITK / 32) * 512;
            g.out_f32 = h->pf_parts; g.ldo = H; g.o_bstride = pstride;
            if (gemm_launch_cfg(g, 4, PF_SPLITK, true, s)) return -1;
            pending = true;
        } else {
            GemmArgs g = gemm_args(h->pf_hi, I, 0, L.wdown, Mp, H, I);
            g.A_lo = h->pf_lo; g.res = h->pf_x; g.ldres = H; g.out_f32 = h->pf_x; g.ldo = H;
            if (gemm_launch_cfg(g, 4, 1, true, s)) return -1;
        }
    }
    if (pending) hipLaunchKernelGGL(k_fold_parts, dim3(((long)M * H + 255) / 256), dim3(256), 0, s, h->pf_x, (const float*)h->pf_parts, PF_SPLITK, pstride, (long)M * H);
    // last row of every prompt -> final norm -> llm_decoder -> first draw
    hipLaunchKernelGGL(k_gather_rows, dim3((n * H + 255) / 256), dim3(256), 0, s, (const float*)h->pf_x, d_last, h->pf_last, n, H);
    {
        StoreArgs a{};
        a.W = h->w.wdec; a.bias = h->w.bdec;
        a.X = SkinnyX{h->pf_last, nullptr, 0, h->w.final_norm, d.rms_eps, nullptr};
        a.KS = H / 32; a.rows = n; a.K = H; a.N = d.vocab_pad; a.out = h->io.logits;
        if (n <= 16) { const size_t sm = skinny_smem_bytes<1, 1, 4>(a.KS); hipLaunchKernelGGL((k_store<1, 8>), dim3(d.vocab_pad / 16, 1), dim3(256), sm, s, a); }
        else { const size_t sm = skinny_smem_bytes<2, 1, 4>(a.KS); hipLaunchKernelGGL((k_store<2, 8>), dim3(d.vocab_pad / 16, 1), dim3(256), sm, s, a); }
    }
    for (int i = 0; i < n; i++)
        if (launch_sample(h, 1, slots[i], i, (pos0 ? pos0[i] : 0) + lens[i], s)) return -1;
    CV2_LAUNCH_CHECK();
    return 0;
}

// Step 0 of inference_wrapper for several slots at once: embeds = the prompts' rows concatenated [sum(lens)][hidden] fp32
extern "C" int cv2_llm_prefill_batch(cv2_llm* h, int32_t n, const int32_t* slots, const int32_t* lens, const float* embeds, void* stream) {
    return rows_batch(h, n, slots, lens, nullptr, embeds, stream);
}
// cv2_llm_extend for several slots in one pass over the weights (the text blocks of concurrent inference_bistream calls, llm.py:787-811)
extern "C" int cv2_llm_extend_batch(cv2_llm* h, int32_t n, const int32_t* slots, const int32_t* lens, const int32_t* pos0, const float* embeds,
                                    void* stream) {
    CV2_CHECK(pos0, "cv2_llm_extend_batch: null argument");
    return rows_batch(h, n, slots, lens, pos0, embeds, stream);
}

// one captured graph holds `unroll` consecutive decode steps for n_seqs slots (key = n_seqs * 64 + unroll)
static int get_graph(cv2_llm* h, int n_seqs, int unroll, bool one_launch, hipGraphExec_t* out, bool mapped = false) {
    const int key = ((n_seqs * 64 + unroll) * 2 + (one_launch ? 1 : 0)) * 2 + (mapped ? 1 : 0);
    auto it = h->graphs.find(key);
    if (it == h->graphs.end()) {
        hipGraph_t g;
        hipStream_t cs = h->cap_stream;
        const bool transient = cs == nullptr;
        if (transient) CV2_HIP(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
        CV2_HIP(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
        RowMap rm{h->io.state, 0, 0, 0, mapped ? h->row_slots : nullptr};
        const float* xin = mapped ? h->xrows : h->xnext;
        int rc = 0;
        for (int u = 0; u < unroll && !rc; u++) {
            if (one_launch) {
                const cv2_llm_dims& d = h->d;
                const int nks_max = std::max(std::max(d.hidden / 32, d.n_q * 64 / 32), cdiv(d.inter / 32, CH_NP));
                const size_t sm = std::max((size_t)r1_smem_bytes(nks_max), (size_t)AT_SMEM_FLOATS * sizeof(float));
                StepArgs a = h->step;
#ifdef CV2_STAMPS
                a.dbg_layer = getenv("CV2_DBG_LAYER") ? atoi(getenv("CV2_DBG_LAYER")) : 12;      // (tools/dbg_chain.py: the layer whose blocks are stamped)
#endif
                static const bool force_multi = getenv("CV2_CHAIN_FORCE_MULTI") != nullptr;       // diagnostics: one row through k_step<true>
                // one row: k_step, or with CV2_STEP1 = 1 / 2 / 3 k_step1 (round 5: bit 0 QA blocks that compute their query head themselves, bit 1
                // gate/up blocks of two tile pairs -- same ids and logits bit for bit, measured slower / equal: profiles/r5_decode_step_experiments.txt)
                const int step1_env = getenv("CV2_STEP1") ? atoi(getenv("CV2_STEP1")) : 0;      // (read at capture: a new engine takes the current value)
                const int step1 = step1_env & ((d.inter / 16) % 2 == 0 ? 3 : 1) & (d.n_q <= QA_HEADS ? 3 : 2);
                if (n_seqs == 1 && !mapped && !force_multi && step1) {
                    const bool qa = step1 & 1, gu2 = step1 & 2;
                    const int nO = d.hidden / 16;
                    const int per1 = (qa ? 4 * d.n_kv + a.ntiles * QA_HEADS : 2 * (d.n_q + 2 * d.n_kv) + a.ntiles * d.n_kv) + nO + (gu2 ? d.inter / 32 : d.inter / 16) + CH_NP * nO;
                    const size_t sm1 = std::max(sm, qa ? qa_smem_bytes(d.hidden / 32) : (size_t)t2_smem_bytes(d.hidden / 32));
                    const dim3 grid1(d.layers * per1 + d.vocab_pad / 16);
                    if (qa && gu2) hipLaunchKernelGGL((k_step1<true, true>), grid1, dim3(R1_THREADS), sm1, cs, a);
                    else if (qa) hipLaunchKernelGGL((k_step1<true, false>), grid1, dim3(R1_THREADS), sm1, cs, a);
                    else hipLaunchKernelGGL((k_step1<false, true>), grid1, dim3(R1_THREADS), sm1, cs, a);
                } else
                if (n_seqs == 1 && !mapped && !force_multi) hipLaunchKernelGGL((k_step<false, true>), dim3(h->step_blocks), dim3(R1_THREADS), sm, cs, a);
                else {                       // rows = slots 0 .. n - 1, or the live slots of cv2_llm_decode_rows (row -> slot map, inputs by row)
                    a.n_rows = n_seqs; a.row_slots = mapped ? h->row_slots : nullptr; a.xin = xin;
                    static const int spec_env = getenv("CV2_CHAIN_SPEC") ? atoi(getenv("CV2_CHAIN_SPEC")) : 1;       // A/B switches (diagnostics)
                    static const int nt_env = getenv("CV2_CHAIN_NT") ? atoi(getenv("CV2_CHAIN_NT")) : 0;
                    static const bool inter_env = getenv("CV2_CHAIN_MULTI") && getenv("CV2_CHAIN_MULTI")[0] == 'i';     // "inter": one chain per row (k_step<true>)
                    a.spec = n_seqs >= 2 ? spec_env : 0;
                    static const int quad_from = getenv("CV2_CHAIN_QUADS") ? atoi(getenv("CV2_CHAIN_QUADS")) : 9;      // rows from which the chains take four rows (A/B switch)
                    if (n_seqs >= quad_from && !inter_env) {
                        const int P = (n_seqs + 3) / 4;
                        const size_t sm4 = std::max((size_t)r4_smem_bytes(nks_max), (size_t)AT_SMEM_FLOATS * sizeof(float));
                        const dim3 grid4((d.layers * (a.per + 3 * a.ntiles * d.n_kv + d.hidden / 16) + d.vocab_pad / 16) * P);
                        if (P == 1 || nt_env) hipLaunchKernelGGL(k_step4<true>, grid4, dim3(R1_THREADS), sm4, cs, a);
                        else hipLaunchKernelGGL(k_step4<false>, grid4, dim3(R1_THREADS), sm4, cs, a);
                    } else
                    if (n_seqs >= 2 && n_seqs != 3 && !inter_env) {  // pairs of rows: two MFMA columns per block (3 rows: one chain per row is faster)
                        const int P = (n_seqs + 1) / 2;
                        const size_t sm2 = std::max((size_t)r2_smem_bytes(nks_max), (size_t)AT_SMEM_FLOATS * sizeof(float));
                        const dim3 grid2((d.layers * (a.per + a.ntiles * d.n_kv) + d.vocab_pad / 16) * P);
                        if (P == 1 || nt_env) hipLaunchKernelGGL(k_step2<true>, grid2, dim3(R1_THREADS), sm2, cs, a);      // one pair: every weight byte is read once
                        else hipLaunchKernelGGL(k_step2<false>, grid2, dim3(R1_THREADS), sm2, cs, a);
                    } else
                    if (nt_env || n_seqs == 1) hipLaunchKernelGGL((k_step<true, true>), dim3(h->step_blocks * n_seqs), dim3(R1_THREADS), sm, cs, a);
                    else hipLaunchKernelGGL((k_step<true, false>), dim3(h->step_blocks * n_seqs), dim3(R1_THREADS), sm, cs, a);
                }
                if (hipGetLastError() != hipSuccess) rc = cv2_fail("k_step launch failed");
            } else
            rc = n_seqs <= 16 ? run_layers<1>(h, n_seqs, xin, rm, cs) : run_layers_pre(h, n_seqs, xin, rm, cs);
            STAMP_SET_ON(cs, 5);
            if (!rc) rc = launch_sample(h, n_seqs, -1, 0, 0, cs, mapped);
        }
        hipError_t e = hipStreamEndCapture(cs, &g);
        if (transient) (void)hipStreamDestroy(cs);
        if (rc) return rc;
        CV2_HIP(e);
        hipGraphExec_t ge;
        CV2_HIP(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CV2_HIP(hipGraphDestroy(g));
        it = h->graphs.emplace(key, ge).first;
    }
    *out = it->second;
    return 0;
}

extern "C" int cv2_llm_decode_ex(cv2_llm* h, int32_t n_seqs, int32_t n_steps, int32_t flags, void* stream);
extern "C" int cv2_llm_decode(cv2_llm* h, int32_t n_seqs, int32_t n_steps, void* stream) { return cv2_llm_decode_ex(h, n_seqs, n_steps, 0, stream); }

extern "C" int cv2_llm_decode_ex(cv2_llm* h, int32_t n_seqs, int32_t n_steps, int32_t flags, void* stream) {
    CV2_CHECK(h, "cv2_llm_decode: null handle");
    // one row and the device to itself: the whole step is one launch (k_step); CV2_DECODE_SHARED asks for the launches instead -- k_step
    // keeps ~1000 polling waves resident, which slows kernels of other streams running beside it more than the launches do
    static const bool shared_one = getenv("CV2_SHARED_ONE_LAUNCH") != nullptr;        // A/B switch (diagnostics): the one-launch step also beside other streams' kernels
    const bool one_launch = n_seqs <= chain_rows() && h->use_chain && (shared_one || !(flags & CV2_DECODE_SHARED));
    CV2_CHECK(n_seqs >= 1 && n_seqs <= h->d.max_seqs, "cv2_llm_decode: n_seqs %d out of range", n_seqs);
    if (init_attrs_once()) return -1;
    hipStream_t s = (hipStream_t)stream;
    // graph replays cost ~8 us of host/queue gap each: 8 steps per replay where possible (finished slots idle inside a replay)
    constexpr int UNROLL = 8;
    hipGraphExec_t g8 = nullptr, g1 = nullptr;
    int i = 0;
    if (n_steps >= UNROLL) {
        if (get_graph(h, n_seqs, UNROLL, one_launch, &g8)) return -1;
        for (; i + UNROLL <= n_steps; i += UNROLL) CV2_HIP(hipGraphLaunch(g8, s));
    }
    if (i < n_steps) {
        if (get_graph(h, n_seqs, 1, one_launch, &g1)) return -1;
        for (; i < n_steps; i++) CV2_HIP(hipGraphLaunch(g1, s));
    }
    return 0;
}

// decode over the LIVE slots only: row r of every launch serves slot slots[r].  cv2_llm_decode(n_seqs) steps slots 0 .. n_seqs - 1
// whether or not they have finished (a finished slot idles, its row is still computed); a batch whose requests end at different
// lengths spends most of its steps on rows nobody needs.  The map lives in device memory (the captured graphs read it), it is
// written here in stream order together with the rows' pending input embeddings (k_sample keeps those by slot).
struct SetRowsArgs { int slots[32]; int n; int* dst; const float* xnext; float* xrows; int hidden; };
__global__ __launch_bounds__(256) void k_set_rows(SetRowsArgs a) {
    const int r = blockIdx.x, slot = a.slots[r];
    if (threadIdx.x == 0) a.dst[r] = slot;
    for (int k = threadIdx.x; k < a.hidden; k += 256) a.xrows[(size_t)r * a.hidden + k] = a.xnext[(size_t)slot * a.hidden + k];
}
extern "C" int cv2_llm_decode_rows(cv2_llm* h, const int32_t* slots, int32_t n_rows, int32_t n_steps, int32_t flags, void* stream) {
    CV2_CHECK(h && slots, "cv2_llm_decode_rows: null argument");
    CV2_CHECK(n_rows >= 1 && n_rows <= h->d.max_seqs && n_rows <= 32, "cv2_llm_decode_rows: n_rows %d out of range", n_rows);
    SetRowsArgs sa{};
    unsigned seen = 0;
    for (int r = 0; r < n_rows; r++) {
        CV2_CHECK(slots[r] >= 0 && slots[r] < h->d.max_seqs, "cv2_llm_decode_rows: bad slot %d", slots[r]);
        CV2_CHECK(!(seen >> slots[r] & 1u), "cv2_llm_decode_rows: slot %d listed twice", slots[r]);
        seen |= 1u << slots[r];
        sa.slots[r] = slots[r];
    }
    if (n_rows == 1 && slots[0] == 0) return cv2_llm_decode_ex(h, 1, n_steps, flags, stream);      // slot 0 alone: the one-row form of the one-launch step
    static const bool shared_one = getenv("CV2_SHARED_ONE_LAUNCH") != nullptr;
    const bool one_launch = n_rows <= chain_rows() && h->use_chain && (shared_one || !(flags & CV2_DECODE_SHARED));
    if (init_attrs_once()) return -1;
    hipStream_t s = (hipStream_t)stream;
    sa.n = n_rows; sa.dst = h->row_slots; sa.xnext = h->xnext; sa.xrows = h->xrows; sa.hidden = h->d.hidden;
    hipLaunchKernelGGL(k_set_rows, dim3(n_rows), dim3(256), 0, s, sa);
    CV2_LAUNCH_CHECK();
    constexpr int UNROLL = 8;
    hipGraphExec_t g8 = nullptr, g1 = nullptr;
    int i = 0;
    if (n_steps >= UNROLL) {
        if (get_graph(h, n_rows, UNROLL, one_launch, &g8, true)) return -1;
        for (; i + UNROLL <= n_steps; i += UNROLL) CV2_HIP(hipGraphLaunch(g8, s));
    }
    if (i < n_steps) {
        if (get_graph(h, n_rows, 1, one_launch, &g1, true)) return -1;
        for (; i < n_steps; i++) CV2_HIP(hipGraphLaunch(g1, s));
    }
    return 0;
}

int skinny_gemm_launch(const uint16_t* w, const float* bias, const float* x, float* out, int rows, int n, int k, hipStream_t s) {
    CV2_CHECK(w && x && out, "skinny_gemm: null argument");
    CV2_CHECK(rows >= 1 && rows <= 32 && n % 16 == 0 && k % 32 == 0, "skinny_gemm: need rows<=32, n%%16==0, k%%32==0");
    CV2_CHECK(k <= 1024, "skinny_gemm: k must be <= 1024");
    if (init_attrs_once()) return -1;
    StoreArgs a{};
    a.W = w; a.bias = bias; a.X = SkinnyX{x, nullptr, 0, nullptr, 0.f, nullptr};
    a.KS = k / 32; a.rows = rows; a.K = k; a.N = n; a.out = out;
    if (rows <= 16) {
        const size_t sm = skinny_smem_bytes<1, 1, 4>(a.KS);
        hipLaunchKernelGGL((k_store<1, 8>), dim3(n / 16, 1), dim3(256), sm, s, a);
    } else {
        const size_t sm = skinny_smem_bytes<2, 1, 4>(a.KS);
        hipLaunchKernelGGL((k_store<2, 8>), dim3(n / 16, 1), dim3(256), sm, s, a);
    }
    CV2_LAUNCH_CHECK();
    return 0;
}

extern "C" int cv2_skinny_gemm(const uint16_t* w, const float* bias, const float* x, float* out, int32_t rows, int32_t n,
                               int32_t k, void* stream) {
    return skinny_gemm_launch(w, bias, x, out, rows, n, k, (hipStream_t)stream);
}
