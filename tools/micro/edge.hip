// Calibration of the persistent decode step's skeleton: what does one layer's chain of in-launch hand-offs cost when every
// hand-off is an 8-byte {tag, value} granule (sc1 store, relaxed agent-scope polling; cdna_hip_programming.md Guideline 16 R2)?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/edge.hip -o /tmp/edge && /tmp/edge
// One block per CU, 256 threads.  Per layer five phases with the real producer / consumer counts and granule volumes of the
// Qwen2-0.5B decode layer (no arithmetic, a checksum carries the dependency); optionally every block also streams its share of
// the layer's 29 MB of weights (non-temporal loads into registers, consumed one layer later) beside the chain.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef unsigned long long u64;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct Phase { int c0, c1; int gbuf, gper, gshared; int pbuf, pper; };   // consumers [c0,c1): gather `gper` granules of buffer gbuf
                                                                          // (gshared: every consumer reads the same region, else its own
                                                                          // slice), then publish `pper` granules into buffer pbuf
#define NPH 5
struct Args { u64* bufs[NPH]; int bsize[NPH]; Phase ph[NPH]; int layers; unsigned epoch; unsigned* abort_flag; float* out;
              const u32x4* w; long w_per_layer16; int wloads; };

__device__ __forceinline__ u64 ld_g(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_g(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int WL, bool SENT>
__global__ __launch_bounds__(256, 1) void k_chain(Args a) {
    __shared__ float red[4];
    __shared__ int fail;
    const int tid = threadIdx.x, b = blockIdx.x;
    if (tid == 0) fail = 0;
    __syncthreads();
    u32x4 wreg[WL > 0 ? WL : 1];
    float carry = 0.f;
    const u64 t_start = __builtin_amdgcn_s_memrealtime();
    if (WL > 0) {
#pragma unroll
        for (int i = 0; i < WL; i++) wreg[i] = __builtin_nontemporal_load(a.w + ((long)b * WL + i) * 256 + tid);
    }
    for (int l = 0; l < a.layers; l++) {
        for (int p = 0; p < NPH; p++) {
            const Phase ph = a.ph[p];
            if (b < ph.c0 || b >= ph.c1) continue;
            const int ci = b - ph.c0;
            // ---- gather
            const u64* src = a.bufs[ph.gbuf] + (size_t)l * a.bsize[ph.gbuf] + (ph.gshared ? 0 : (size_t)ci * ph.gper);
            float s = 0.f;
            bool ok = false;
            if (SENT) {      // wave 0 polls ONE granule (the last of the region) with s_sleep, the other waves wait at the barrier
                if (tid < 64) {
                    for (unsigned spin = 0;; spin++) {
                        const u64 g = ld_g(src + ph.gper - 1);
                        if ((unsigned)(g >> 32) == a.epoch) break;
                        if ((spin & 63) == 63 && (__builtin_amdgcn_s_memrealtime() - t_start > 20000000ull || ld_g((const u64*)a.abort_flag) != 0)) {
                            if (tid == 0) { atomicExch(a.abort_flag, 1u + p); fail = 1; }
                            break;
                        }
                        __builtin_amdgcn_s_sleep(2);
                    }
                }
                __syncthreads();
                if (fail) return;
            }
            for (unsigned spin = 0; !ok; spin++) {
                s = 0.f;
                bool all = true;
                for (int i = tid; i < ph.gper; i += 256) {
                    const u64 g = ld_g(src + i);
                    all &= (unsigned)(g >> 32) == a.epoch;
                    s += __builtin_bit_cast(float, (unsigned)g);
                }
                ok = __all(all);
                if (!ok && (spin & 63) == 63) {
                    if (__builtin_amdgcn_s_memrealtime() - t_start > 20000000ull || ld_g((const u64*)a.abort_flag) != 0) {   // 200 ms
                        if ((tid & 63) == 0) { atomicExch(a.abort_flag, 1u + p); fail = 1; }
                        break;
                    }
                }
            }
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            if ((tid & 63) == 0) red[tid >> 6] = s;
            __syncthreads();
            if (fail) return;
            const float tot = red[0] + red[1] + red[2] + red[3];
            __syncthreads();
            // ---- "compute": consume the prefetched weights (checksum), then request the next layer's
            float v = tot * 1e-3f + carry;
            if (WL > 0) {
                unsigned acc = 0;
#pragma unroll
                for (int i = 0; i < WL; i++) acc ^= wreg[i][0] ^ wreg[i][1] ^ wreg[i][2] ^ wreg[i][3];
                v += (float)(acc & 1);
            }
            // ---- publish
            u64* dst = a.bufs[ph.pbuf] + (size_t)(ph.pbuf == 0 ? l + 1 : l) * a.bsize[ph.pbuf] + (size_t)ci * ph.pper;
            for (int i = tid; i < ph.pper; i += 256) st_g(dst + i, ((u64)a.epoch << 32) | __builtin_bit_cast(unsigned, v + 1.f));
            if (WL > 0 && l + 1 < a.layers) {
#pragma unroll
                for (int i = 0; i < WL; i++) wreg[i] = __builtin_nontemporal_load(a.w + (l + 1) * a.w_per_layer16 + ((long)b * WL + i) * 256 + tid);
            }
            carry = v * 1e-3f;
        }
    }
    if (tid == 0) a.out[b] = carry;
}

__global__ void k_fill(u64* p, int n, unsigned epoch) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = ((u64)epoch << 32) | __builtin_bit_cast(unsigned, 1.0f);
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int G = prop.multiProcessorCount;
    printf("CUs: %d\n", G);
    if (G < 256) { printf("needs 256 CUs\n"); return 0; }
    const int layers = 24;
    // buffers: 0 = x (4480: x_mid 896 + 4 down partials), 1 = q/k/v (1152), 2 = attention partials (8 x 462), 3 = x_mid (896), 4 = h (4864)
    const int bsize[NPH] = {4480, 1152, 3696, 896, 4864};
    Args a{};
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int i = 0; i < NPH; i++) { a.bsize[i] = bsize[i]; CK(hipMalloc(&a.bufs[i], (size_t)(layers + 1) * bsize[i] * 8)); CK(hipMemset(a.bufs[i], 0, (size_t)(layers + 1) * bsize[i] * 8)); }
    CK(hipMalloc(&a.abort_flag, 64)); CK(hipMalloc(&a.out, G * 4));
    const size_t wbytes = (size_t)layers * 32 * 1024 * 1024;
    u32x4* w; CK(hipMalloc(&w, wbytes)); CK(hipMemset(w, 1, wbytes));
    a.w = w; a.w_per_layer16 = 32 * 1024 * 1024 / 16;
    a.layers = layers;
    //            consumers   gather                      publish
    a.ph[0] = Phase{0, 36,    0, 4480, 1,                 1, 32};     // Q: 36 units read all of x, write 32 of q/k/v each
    a.ph[1] = Phase{244, 252, 1, 1152, 1,                 2, 462};    // A: 8 units read q/k/v, write 462 each
    a.ph[2] = Phase{36, 92,   2, 3696, 1,                 3, 16};     // O: 56 units read all partials, write 16 of x_mid each
    a.ph[3] = Phase{92, 244,  3, 896, 1,                  4, 32};     // G: 152 units read x_mid, write 32 of h each
    a.ph[4] = Phase{0, 92,    4, 3648, 1,                 0, 48};     // D: 92 blocks (2-3 units each) read 3 x 1216 of h, write 48 (92 x 48 = 4416 of 4480)
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    unsigned epoch_ctr = 100;
    bool small = false;
    auto run = [&](const char* name, int wl, bool sent) -> int {
        float best = 1e9f;
        for (int rep = 0; rep < 6; rep++) {
            a.epoch = ++epoch_ctr;
            CK(hipMemsetAsync(a.abort_flag, 0, 64, s));
            hipLaunchKernelGGL(k_fill, dim3((4480 + 255) / 256), dim3(256), 0, s, a.bufs[0], 4480, a.epoch);       // layer 0 input
            // the D phase writes only 4416 of 4480: pre-fill the tail of every layer's x buffer
            for (int l = 1; l <= layers; l++) hipLaunchKernelGGL(k_fill, dim3(1), dim3(256), 0, s, a.bufs[0] + (size_t)l * 4480 + 4416, 64, a.epoch);
            (void)small;
            a.wloads = wl;
            CK(hipEventRecord(e0, s));
            if (wl == 0 && !sent) hipLaunchKernelGGL((k_chain<0, false>), dim3(G), dim3(256), 0, s, a);
            else if (wl == 0) hipLaunchKernelGGL((k_chain<0, true>), dim3(G), dim3(256), 0, s, a);
            else if (wl == 28 && !sent) hipLaunchKernelGGL((k_chain<28, false>), dim3(G), dim3(256), 0, s, a);
            else hipLaunchKernelGGL((k_chain<28, true>), dim3(G), dim3(256), 0, s, a);
            CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned ab; CK(hipMemcpy(&ab, a.abort_flag, 4, hipMemcpyDeviceToHost));
            if (ab) { printf("%s: ABORT (timeout) in phase %u\n", name, ab - 1); return 0; }
            if (rep > 0 && ms < best) best = ms;
        }
        printf("%-70s %8.1f us per launch = %6.2f us per layer (%d layers)\n", name, best * 1e3, best * 1e3 / layers, layers);
        return 0;
    };
    if (run("chain only, every waiting CU sweeps its whole region", 0, false)) return 1;
    if (run("chain only, sentinel poll (one granule, s_sleep) then sweep", 0, true)) return 1;
    if (run("chain + ~29 MB/layer weight prefetch, full sweeps", 28, false)) return 1;
    if (run("chain + ~29 MB/layer weight prefetch, sentinel", 28, true)) return 1;
    // latency floor: the same chain with tiny regions (64 granules gathered, 1-2 published per consumer)
    const Phase keep[NPH] = {a.ph[0], a.ph[1], a.ph[2], a.ph[3], a.ph[4]};
    for (int i = 0; i < NPH; i++) { a.ph[i].gper = 64; }
    a.ph[0].pper = 2; a.ph[1].pper = 8; a.ph[2].pper = 2; a.ph[3].pper = 1; a.ph[4].pper = 1;      // 36x2=72>=64, 8x8=64, 56x2, 152x1, 92x1
    small = true;
    if (run("tiny regions (64 granules), sentinel: latency floor of 5 hops", 0, true)) return 1;
    for (int i = 0; i < NPH; i++) a.ph[i] = keep[i];
    return 0;
}
