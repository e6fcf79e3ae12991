// What does a 16-byte sc1 buffer load return for two 8-byte granules?  hipcc --offload-arch=gfx950 -O3 tools/micro/ld2.hip -o tools/micro/ld2
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned long long u64;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__global__ void k(u64* g, unsigned* out, int aux_sel) {
    const int t = threadIdx.x;
    g[t * 2] = ((u64)7 << 32) | (unsigned)(100 + t * 2);
    g[t * 2 + 1] = ((u64)7 << 32) | (unsigned)(101 + t * 2);
    __syncthreads();
    __threadfence();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, 64 * 16, 0x00020000);
    u32x4 x;
    if (aux_sel == 0) x = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, t * 16, 0, 0));
    else x = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, t * 16, 0, 16));
    for (int i = 0; i < 4; i++) out[t * 4 + i] = x[i];
}
int main() {
    u64* g; unsigned* o; hipMalloc(&g, 64 * 16); hipMalloc(&o, 64 * 16);
    unsigned h[256];
    for (int a = 0; a < 2; a++) {
        hipMemset(g, 0, 64 * 16);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, g, o, a);
        hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
        printf("aux %d: lane0 %u %u %u %u  lane1 %u %u %u %u  lane5 %u %u %u %u\n", a ? 16 : 0, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[20], h[21], h[22], h[23]);
    }
    return 0;
}
