// Skeleton of a MANY-ROW (32 rows) one-launch decode step: what do a layer's hand-offs cost when the operands are bulk bf16 planes
// (115 KB per consumer) instead of 8-byte granules?  Flag form of cdna_hip_programming.md Guideline 16: producers write their slice with
// plain 16-byte stores, release at agent scope, then store one flag word {epoch}; a consumer's wave 0 polls ALL flags of the previous hop
// (one load instruction per 64 producers), the block acquires at agent scope and reads its operand with plain loads (L2 is then
// allowed to serve the blocks of an XCD after the first one fetched a line).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/rowchain.hip -o /tmp/rowchain && /tmp/rowchain
// Blocks are laid out in dependency order (layer by layer, hop by hop) as in k_step: a block only waits for lower block indices.
// No arithmetic: a block sums what it reads (the sum is checked: stale data fails the run) and writes a constant; optionally it
// streams its share of the layer's weights first (non-temporal loads, consumed at the end).  The same bodies run as one launch per hop
// for comparison (no flags, no fences).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define NH 12                                  // hops per layer at most (Args::nh are used)
#define THREADS 512
#ifndef WMAX
#define WMAX 32                                // 16-byte weight loads per thread at most (the estimator table: -DWMAX=96)
#endif
struct Hop { int nb; int rd_bytes; int rd_whole; int wr_total; int w_bytes; int share; };   // share: consecutive blocks that read the SAME slice (0 / 1: every block its own)   // blocks; operand bytes per block (whole prev buffer or own slice); bytes of this hop's output; weight bytes per block
struct Args {
    Hop hop[NH]; int nh; int per_layer; int layers; int dep;      // dep: a hop's producer is the hop `dep` places earlier (2: two row chains interleaved)
    float* out[NH]; int out_stride[NH];       // per layer (floats)
    unsigned* flags; int flag_off[NH];        // [layer][sum nb]
    const f32x4* w; unsigned epoch; unsigned* err; float* sink; int single_hop; int single_layer;
};

// SC1: the hand-off data travels with write-through (sc1) stores and sc1 loads, no fences (the form the granules of k_step use, without tags);
// else plain stores / loads between an agent-scope release and acquire fence.
template <bool CHAIN, bool WEIGHTS, bool SC1 = false>
__global__ __launch_bounds__(THREADS) void k_rows(Args a) {
    __shared__ int fail;
    __shared__ float red[THREADS / 64];
    const int tid = threadIdx.x;
    int l, h, i;
    if (CHAIN) {
        l = blockIdx.x / a.per_layer;
        int r = blockIdx.x - l * a.per_layer;
        for (h = 0; h < a.nh - 1 && r >= a.hop[h].nb; h++) r -= a.hop[h].nb;
        i = r;
    } else { l = a.single_layer; h = a.single_hop; i = blockIdx.x; }
    const Hop H = a.hop[h];
    const int gidx = l * a.per_layer + a.flag_off[h] + i;                        // the block's index in dependency order (= blockIdx.x of the one-launch form)
    const int ph = h < a.dep ? a.nh - a.dep + h : h - a.dep, pl = h < a.dep ? l - 1 : l;      // producer hop / layer
    // ---- weights first
    f32x4 wr[8];
    float wsum = 0.f;
    const int wl = WEIGHTS ? H.w_bytes / (THREADS * 16) : 0;                     // 16-byte loads per thread (<= 16: two batches of 8)
    if (WEIGHTS) {
        const f32x4* wp = a.w + ((size_t)gidx * WMAX) * THREADS + tid;
#pragma unroll
        for (int k = 0; k < 8; k++) wr[k] = k < wl ? __builtin_nontemporal_load(wp + (size_t)k * THREADS) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // ---- wait for every producer of the previous hop
    if (CHAIN && pl >= 0) {
        if (tid == 0) fail = 0;
        __syncthreads();
        if (tid < 64) {
            const unsigned* f = a.flags + (size_t)pl * a.per_layer + a.flag_off[ph];
            const int np = a.hop[ph].nb;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            for (unsigned spin = 0;; spin++) {
                bool ok = true;
                for (int k = tid; k < np; k += 64) ok &= __hip_atomic_load(f + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.epoch;
                if (__all(ok)) break;
                if ((spin & 31) == 31 && (__builtin_amdgcn_s_memrealtime() - t0 > 20000000ull || __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                    if (tid == 0) { atomicCAS(a.err, 0u, 100u + h); fail = 1; }
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        __syncthreads();
        if (fail) return;
        if (!SC1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    // ---- operand: the whole previous buffer or this block's slice of it
    float s = 0.f;
    if (pl >= 0) {
        const float* src = a.out[ph] + (size_t)pl * a.out_stride[ph];
        const int total = a.hop[ph].wr_total;
        long off = H.rd_whole ? 0 : ((long)(i / (H.share > 1 ? H.share : 1)) * H.rd_bytes) % (total - H.rd_bytes + 16);
        off &= ~15l;
        const f32x4* p = reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(src) + off);
        const int n16 = H.rd_bytes / 16;
        if (CHAIN && SC1) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, H.rd_bytes, 0x00020000);
            for (int k = tid; k < n16; k += THREADS) {
                const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, k * 16, 0, 16));      // aux 16 = sc1
                s += (v[0] + v[1]) + (v[2] + v[3]);
            }
        } else
        for (int k = tid; k < n16; k += THREADS) { const f32x4 v = p[k]; s += (v[0] + v[1]) + (v[2] + v[3]); }
    }
    // block sum -> check against what the producers wrote
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    float tot = 0.f;
    for (int k = 0; k < THREADS / 64; k++) tot += red[k];
    if (pl >= 0) {
        const float want = (float)(H.rd_bytes / 4) * (float)(1 + ((pl * a.nh + ph + a.epoch) & 7));      // (the constant moves with the epoch: a stale line fails)
        if (tid == 0 && tot != want) atomicCAS(a.err, 0u, 1000u + h);                 // (the first error stays)
    }
    if (WEIGHTS) {
#pragma unroll
        for (int k = 0; k < 8; k++) wsum += wr[k][0] + wr[k][3];
        for (int b = 8; b < wl; b += 8) {              // further batches of 8 (rows16 tables: up to 28 loads per thread)
            const f32x4* wp = a.w + ((size_t)gidx * WMAX + b) * THREADS + tid;
#pragma unroll
            for (int k = 0; k < 8; k++) { const f32x4 v = k + b < wl ? __builtin_nontemporal_load(wp + (size_t)k * THREADS) : (f32x4){0.f, 0.f, 0.f, 0.f}; wsum += v[0] + v[3]; }
        }
    }
    // ---- this block's slice of the hop's output
    {
        const int slice = (H.wr_total / H.nb) & ~15;
        float* dst = a.out[h] + (size_t)l * a.out_stride[h] + (size_t)i * (slice / 4);
        const float c = (float)(1 + ((l * a.nh + h + a.epoch) & 7));
        const f32x4 v = {c, c, c, c};
        if (CHAIN && SC1) {
            for (int k = tid; k < slice / 16; k += THREADS) {
                f32x4* q = reinterpret_cast<f32x4*>(dst) + k;
                asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(q), "v"(v) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else
        for (int k = tid; k < slice / 16; k += THREADS) reinterpret_cast<f32x4*>(dst)[k] = v;
        if (wsum == 12345.678f) a.sink[0] = wsum + tot;
    }
    if (CHAIN) {
        if (!SC1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(a.flags + (size_t)l * a.per_layer + a.flag_off[h] + i, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

static int run_table(const char* title, const Hop* hops, int nh, int dep = 1) {
    const int LAYERS = 24;
    Args a{};
    a.nh = nh; a.dep = dep;
    int per = 0;
    for (int h = 0; h < nh; h++) { a.hop[h] = hops[h]; a.flag_off[h] = per; per += hops[h].nb; }
    a.per_layer = per; a.layers = LAYERS;
    for (int h = 0; h < nh; h++) {
        const int slice = (hops[h].wr_total / hops[h].nb) & ~15;
        a.hop[h].wr_total = slice * hops[h].nb;       // what the blocks really write
        a.out_stride[h] = a.hop[h].wr_total / 4;
        CK(hipMalloc(&a.out[h], (size_t)LAYERS * a.hop[h].wr_total));
        CK(hipMemset(a.out[h], 0, (size_t)LAYERS * a.hop[h].wr_total));
    }
    for (int h = 0; h < nh; h++) {                    // operands never exceed what the producer hop wrote
        const int ph = h < dep ? nh - dep + h : h - dep;
        if (a.hop[h].rd_bytes > a.hop[ph].wr_total) a.hop[h].rd_bytes = a.hop[ph].wr_total;
        a.hop[h].rd_bytes &= ~15;
    }
    CK(hipMalloc(&a.flags, (size_t)LAYERS * per * 4));
    CK(hipMemset(a.flags, 0, (size_t)LAYERS * per * 4));
    CK(hipMalloc(&a.err, 4)); CK(hipMemset(a.err, 0, 4));
    CK(hipMalloc(&a.sink, 4));
    const size_t wbytes = (size_t)LAYERS * per * WMAX * THREADS * 16;
    f32x4* w; CK(hipMalloc(&w, wbytes)); CK(hipMemset(w, 0, wbytes));
    a.w = w;
    long wsum = 0, rsum = 0;
    for (int h = 0; h < nh; h++) { wsum += (long)a.hop[h].nb * a.hop[h].w_bytes; rsum += (long)a.hop[h].nb * a.hop[h].rd_bytes; }
    printf("%s: %d layers x %d blocks (%d hops); per layer %.1f MB of weights, %.1f MB of operand reads\n", title, LAYERS, per, nh, wsum / 1e6, rsum / 1e6);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    static unsigned epoch = 0;
    auto report = [&](const char* name, float ms, int reps) {
        unsigned err = 0; (void)hipMemcpy(&err, a.err, 4, hipMemcpyDeviceToHost);
        printf("  %-58s %8.1f us per step  %6.2f us per layer%s\n", name, ms * 1e3 / reps, ms * 1e3 / reps / LAYERS, err ? "   ERROR" : "");
        if (err) { printf("   error code %u\n", err); (void)hipMemset(a.err, 0, 4); }
    };
    for (int weights = 0; weights < 2; weights++) {
        const int reps = 20;
        float ms;
        for (int warm = 0; warm < 2; warm++) {
            CK(hipEventRecord(e0));
            for (int r = 0; r < reps; r++) {
                a.epoch = ++epoch;
                if (weights) hipLaunchKernelGGL((k_rows<true, true, true>), dim3(LAYERS * per), dim3(THREADS), 0, 0, a);
                else hipLaunchKernelGGL((k_rows<true, false, true>), dim3(LAYERS * per), dim3(THREADS), 0, 0, a);
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        }
        CK(hipEventElapsedTime(&ms, e0, e1));
        report(weights ? "ONE launch, flags + sc1 stores / sc1 loads, weight stream" : "ONE launch, flags + sc1 stores / sc1 loads, no weights", ms, reps);
        for (int warm = 0; warm < 2; warm++) {
            CK(hipEventRecord(e0));
            for (int r = 0; r < reps; r++) {
                a.epoch = ++epoch;
                for (int l = 0; l < LAYERS; l++)
                    for (int h = 0; h < nh; h++) {
                        a.single_layer = l; a.single_hop = h;
                        if (weights) hipLaunchKernelGGL((k_rows<false, true>), dim3(a.hop[h].nb), dim3(THREADS), 0, 0, a);
                        else hipLaunchKernelGGL((k_rows<false, false>), dim3(a.hop[h].nb), dim3(THREADS), 0, 0, a);
                    }
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        }
        CK(hipEventElapsedTime(&ms, e0, e1));
        report(weights ? "one launch per hop, with the weight stream" : "one launch per hop, no weights", ms, reps);
    }
    for (int h = 0; h < nh; h++) (void)hipFree(a.out[h]);
    (void)hipFree(a.flags); (void)hipFree(a.err); (void)hipFree(a.sink); (void)hipFree(w);
    return 0;
}

// Round 6 (the round-5 review's direction): 16 rows per chain -- the rows are the 16 columns of v_mfma_f32_16x16x32_bf16's B operand, every
// weight fragment is used by 16 rows --, two chains for 32 rows interleaved hop by hop in the grid (dep = 2: chain B's hop runs while chain
// A's next hop waits), FIVE hops per chain and layer (Q, A, O, gate/up, down: the folds and norms move into the consumers' operand loads, no
// prep / combine hops), operands as plain hi / lo bf16 planes by 16-byte loads, one flag per producer block.  Volumes per 16-row chain:
// x planes 57 KB; attention partials of 16 rows 115 KB (read whole by each O block: the combine is redone per block); h planes 311 KB.
static int rows16() {
    const int R = 16, Hd = 896, I = 4864;
    const int xb = R * Hd * 4;
    // (a) down K-split two ways (k_step's form): the next Q folds two partial sets + the residual = 3 x 57 KB per block
    const Hop c5[5] = {
        {36, 3 * xb, 1, R * 1152 * 4, 57344},         // Q: fold(2 partial sets + residual) -> norm -> QKV
        {64, 4608, 0, 2 * xb, 65536},                 // A: (row, kv head, 2 tiles); K / V tile as the "weights"; (o, max, sum) partials
        {56, 2 * xb, 1, xb, 28672},                   // O: combine of the 16 rows' partials inside the operand load -> O projection
        {152, xb, 1, R * I * 4, 114688},              // GU
        {112, R * I * 4 / 2, 0, 3 * xb, 77824},       // D: 56 tiles x 2 K halves -> two partial sets (+ the residual rows the O role wrote)
    };
    Hop two[10];
    for (int h = 0; h < 5; h++) two[2 * h] = two[2 * h + 1] = c5[h];
    if (run_table("rows16 (a): two 16-row chains interleaved, 5 hops each, down K-split x 2, consumers fold", two, 10, 2)) return 1;
    if (run_table("rows16 (a) as ONE 16-row chain alone (a 9..16-row step)", c5, 5, 1)) return 1;
    // (b) down unsplit (56 blocks ingest the whole 311 KB of h planes and 155 KB of weights each): Q reads one folded set
    Hop c5b[5];
    for (int h = 0; h < 5; h++) c5b[h] = c5[h];
    c5b[0] = Hop{36, xb, 1, R * 1152 * 4, 57344};
    c5b[4] = Hop{56, R * I * 4, 1, xb, 155648};
    for (int h = 0; h < 5; h++) two[2 * h] = two[2 * h + 1] = c5b[h];
    if (run_table("rows16 (b): the same with down unsplit (56 blocks x 311 KB of h planes), Q reads one folded set", two, 10, 2)) return 1;
    // (c) as (a) with gate/up as 76 blocks of 128 features (half the operand reads of the largest hop)
    Hop c5c[5];
    for (int h = 0; h < 5; h++) c5c[h] = c5[h];
    c5c[3] = Hop{76, xb, 1, R * I * 4, 229376};
    for (int h = 0; h < 5; h++) two[2 * h] = two[2 * h + 1] = c5c[h];
    if (run_table("rows16 (c): as (a) with gate/up as 76 blocks of 128 features", two, 10, 2)) return 1;
    return 0;
}

// Round 6: the flow estimator's transformer block at ONE utterance (M = 2 048 packed rows: the CFG pair of a 10 s utterance) as a chain of
// its three launches -- QKV projection (k_gemm 64 x 128 tiles: 384 blocks, 32 KB of x rows + 64 KB of weights each, 6 MB of q / k / v^T out),
// attention (k_attn_est_dma4: 256 blocks = 32 query tiles x 8 heads; the 32 blocks of a head read the same 258 KB of K / V^T), tail
// (k_tail_panel<2>: 256 workgroups, 16 KB of attention rows + 0.77 MB of weights each) -- as ONE dependency-ordered launch (flags, write-through
// stores, sc1 loads: what an in-launch hand-off of bulk data needs on per-XCD L2s that are not coherent with each other) against one launch
// per hop (plain stores / loads: the consumers' re-reads are served by L2).  Build with -DWMAX=96 (the tail's 0.77 MB per workgroup).
static int estimator() {
    const Hop est[3] = {
        {384, 32768, 0, 2048 * 1536 * 2, 65536, 12},          // QKV: the 12 column tiles of a 64-row tile read the same x rows
        {256, 258560, 0, 2048 * 512 * 2, 0, 32},              // attention: the 32 query tiles of a (sequence, head) share its K / V^T
        {256, 16384, 0, 2 * 2048 * 256 * 2, 786432, 2},       // tail: the two workgroups of a 16-row panel share its attention rows
    };
    return run_table("flow estimator, one utterance: transformer block = QKV -> attention -> tail (3 hops)", est, 3, 1);
}

int main(int argc, char** argv) {
    if (argc > 1 && argv[1][0] == 'r') return rows16();      // ./rowchain rows16
    if (argc > 1 && argv[1][0] == 'e') return estimator();   // ./rowchain_est estimator   (built with -DWMAX=96)
    const int R = 32, Hd = 896, I = 4864;
    const int xb = R * Hd * 4;                          // one activation vector set as hi / lo bf16 planes (or fp32): 114 688 B
    // hop: blocks, operand bytes per block, whole?, output bytes, weight bytes per block
    const Hop today[7] = {
        {36, xb, 1, R * 1152 * 4, 57344},             // Q: x planes -> q, k, v (147 KB); 2.06 MB of W_qkv
        {128, 4608, 0, 2 * xb, 65536},                // A: (row, kv head, 2 tiles): its q slice; K / V tile as the "weights"; partials
        {32, 2 * 3584, 0, xb, 0},                     // C: combine a row's tiles -> attention planes
        {56, xb, 1, xb, 28672},                       // O: attention planes -> x_mid; 1.6 MB of W_o
        {152, xb, 1, R * I * 4, 114688},              // GU: x_mid planes -> h (622 KB); 17.4 MB
        {280, R * I * 4 / 5, 0, 5 * xb, 31232},       // D: K slice of h -> 5 partial sets; 8.7 MB
        {32, 6 * 3584, 0, xb, 0},                     // F: fold + norm of a row -> x planes
    };
    if (run_table("today's many-row step (round 3 table)", today, 7)) return 1;
    // K-sliced QKV (round-4 review): a block = 64 features x K / 4 -- 18 feature tiles x 4 slices = 72 blocks, each ingests a QUARTER of the
    // operand planes (28.7 KB) and 28.7 KB of weights instead of 115 + 57 KB; the four partial sets are folded by the consumer (an attention
    // block reads its q slice from four sets: 18.4 KB instead of 4.6)
    Hop ksq[7];
    for (int h = 0; h < 7; h++) ksq[h] = today[h];
    ksq[0] = Hop{72, xb / 4, 0, 4 * R * 1152 * 4, 28672};
    ksq[1] = Hop{128, 4 * 4608, 0, 2 * xb, 65536};
    if (run_table("K-sliced QKV (72 blocks of 64 features x K / 4, the attention folds four partial sets)", ksq, 7)) return 1;
    // ... and the O projection K-sliced too: 14 feature tiles x 4 slices = 56 blocks of 28.7 + 28.7 KB; its four partial sets + the residual
    // cannot be folded by gate/up (every one of its 152 blocks needs whole rows: 5 x 115 KB each), so a fold hop of 32 row blocks follows
    // (what the O projection's NEXT epilogue does in-kernel today): eight hops
    Hop kso[8];
    kso[0] = ksq[0]; kso[1] = ksq[1]; kso[2] = today[2];
    kso[3] = Hop{56, xb / 4, 0, 4 * xb, 28672};
    kso[4] = Hop{32, 5 * 3584, 0, xb, 0};
    kso[5] = today[4]; kso[6] = today[5]; kso[7] = today[6];
    if (run_table("K-sliced QKV and O (+ a fold hop behind the O projection: 8 hops)", kso, 8)) return 1;
    return 0;
}
