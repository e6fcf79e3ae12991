// Calibration: cost per kernel of a dependent chain replayed from a hipGraph (what bounds the batch-1 decode step).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/chain.hip -o /tmp/chain && /tmp/chain
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_empty(float* p) { if (p && threadIdx.x == 9999) p[0] = 1.f; }
// stream `bytes` once (16 B per lane, all loads of a thread in flight), reduce, one store per block
template <int NLD>
__global__ __launch_bounds__(1024) void k_stream(const u32x4* w, long n16_per_block, const float* dep, float* out) {
    const u32x4* p = w + (long)blockIdx.x * n16_per_block + threadIdx.x;
    u32x4 v[NLD];
#pragma unroll
    for (int i = 0; i < NLD; i++) v[i] = __builtin_nontemporal_load(p + (long)i * blockDim.x);
    float x = dep[threadIdx.x & 63];
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < NLD; i++) acc ^= v[i][0] ^ v[i][1] ^ v[i][2] ^ v[i][3];
    __shared__ float red[16];
    float s = x + (float)(acc & 1);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { float t = 0.f; for (int i = 0; i < (int)blockDim.x / 64; i++) t += red[i]; out[blockIdx.x] = t; }
}
int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int NK = 120, REP = 50;
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const size_t big = (size_t)24 * 18 * 1024 * 1024;          // 24 "layers" x 18 MB: larger than the 256 MiB Infinity Cache
    u32x4* w; CK(hipMalloc(&w, big)); CK(hipMemset(w, 1, big));
    float *a, *b; CK(hipMalloc(&a, 1 << 20)); CK(hipMalloc(&b, 1 << 20)); CK(hipMemset(a, 0, 1 << 20)); CK(hipMemset(b, 0, 1 << 20));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto launch) -> int {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < NK; i++) launch(i);
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 5; i++) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < REP; i++) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-58s %7.2f us per kernel (%d-kernel graph: %.1f us per replay)\n", name, ms * 1e3 / (REP * NK), NK, ms * 1e3 / REP);
        return 0;
    };
    if (run("empty kernel, 1 block x 64", [&](int) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, (float*)nullptr); })) return 1;
    if (run("empty kernel, 304 blocks x 512", [&](int) { hipLaunchKernelGGL(k_empty, dim3(304), dim3(512), 0, s, (float*)nullptr); })) return 1;
    // 2 MB over 36 blocks (qkv-like), 17.4 MB over 304 blocks (gate/up-like), 8.7 MB over 224 blocks x 256 thr equivalent
    if (run("stream  2.0 MB, 36 blocks x 512 (dependent chain)", [&](int i) {
            hipLaunchKernelGGL((k_stream<7>), dim3(36), dim3(512), 0, s, w + (size_t)(i % 24) * (18 << 16), 7L * 512, (i & 1) ? a : b, (i & 1) ? b : a); })) return 1;
    if (run("stream 17.4 MB, 304 blocks x 512 (dependent chain)", [&](int i) {
            hipLaunchKernelGGL((k_stream<7>), dim3(304), dim3(512), 0, s, w + (size_t)(i % 24) * (18 << 16), 7L * 512, (i & 1) ? a : b, (i & 1) ? b : a); })) return 1;
    if (run("stream 17.4 MB, 152 blocks x 512, 14 loads per thread", [&](int i) {
            hipLaunchKernelGGL((k_stream<14>), dim3(152), dim3(512), 0, s, w + (size_t)(i % 24) * (18 << 16), 14L * 512, (i & 1) ? a : b, (i & 1) ? b : a); })) return 1;
    if (run("stream  1.6 MB, 56 blocks x 256", [&](int i) {
            hipLaunchKernelGGL((k_stream<7>), dim3(56), dim3(256), 0, s, w + (size_t)(i % 24) * (18 << 16), 7L * 256, (i & 1) ? a : b, (i & 1) ? b : a); })) return 1;
    // fat blocks: what can ONE workgroup pull when only a few run (fused per-head attention + O-slice block: 230-690 KB per block)?
    if (run("stream 344 KB per block, 14 blocks x 1024 (21 loads per thread)", [&](int i) {
            hipLaunchKernelGGL((k_stream<21>), dim3(14), dim3(1024), 0, s, w + (size_t)(i % 16) * (18 << 16), 21L * 1024, (i & 1) ? a : b, (i & 1) ? b : a); })) return 1;
    if (run("stream 229 KB per block, 28 blocks x 1024 (14 loads per thread)", [&](int i) {
            hipLaunchKernelGGL((k_stream<14>), dim3(28), dim3(1024), 0, s, w + (size_t)(i % 16) * (18 << 16), 14L * 1024, (i & 1) ? a : b, (i & 1) ? b : a); })) return 1;
    if (run("stream 688 KB per block, 14 blocks x 1024 (42 loads per thread)", [&](int i) {
            hipLaunchKernelGGL((k_stream<42>), dim3(14), dim3(1024), 0, s, w + (size_t)(i % 16) * (18 << 16), 42L * 1024, (i & 1) ? a : b, (i & 1) ? b : a); })) return 1;
    if (run("stream 459 KB per block, 42 blocks x 1024 (28 loads per thread)", [&](int i) {
            hipLaunchKernelGGL((k_stream<28>), dim3(42), dim3(1024), 0, s, w + (size_t)(i % 16) * (18 << 16), 28L * 1024, (i & 1) ? a : b, (i & 1) ? b : a); })) return 1;
    if (run("stream 344 KB per block, 76 blocks x 1024 (21 loads per thread, 26 MB)", [&](int i) {
            hipLaunchKernelGGL((k_stream<21>), dim3(76), dim3(1024), 0, s, w + (size_t)(i % 8) * (36 << 16), 21L * 1024, (i & 1) ? a : b, (i & 1) ? b : a); })) return 1;
    return 0;
}
