// checks: rows4_max / rows4_sum (permlane swaps), mfma 16x16x32 bf16 and 16x16x16 bf16_1k lane maps with exact small integers
// hipcc --offload-arch=gfx950 -O3 -I cosyvoice2-eu_amd/csrc tools/micro/rows4.hip -o /tmp/rows4 && /tmp/rows4
#include "common.h"
#include <vector>
thread_local std::string g_cv2_err;
int cv2_fail(const char*, ...) { return -1; }
__global__ void k(float* out) {
    const int l = threadIdx.x;
    const float v = (float)((l * 37) % 101);
    out[l] = rows4_max(v);
    out[64 + l] = rows4_sum(v);
    // 16x16x32: A[i][k] = i + 100 k (small ints exact in bf16? keep < 256): A[i][k] = (i + 3 k) % 7, B[k][j] = (k + 5 j) % 5
    const int c = l & 15, g = l >> 4;
    bf16x8 a, b;
    for (int j = 0; j < 8; j++) { const int kk = 8 * g + j; a[j] = (__bf16)(float)((c + 3 * kk) % 7); b[j] = (__bf16)(float)((kk + 5 * c) % 5); }
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; r++) out[128 + l * 4 + r] = acc[r];      // claimed: row 4 g + r, col c
    s16x4 a4, b4;
    for (int j = 0; j < 4; j++) {
        const int kk = 4 * g + j;
        a4[j] = (short)(__builtin_bit_cast(unsigned, (float)((c + 3 * kk) % 7)) >> 16);
        b4[j] = (short)(__builtin_bit_cast(unsigned, (float)((kk + 5 * c) % 5)) >> 16);
    }
    f32x4 acc2 = {0, 0, 0, 0};
    acc2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc2, 0, 0, 0);
    for (int r = 0; r < 4; r++) out[384 + l * 4 + r] = acc2[r];
}
int main() {
    float* d; hipMalloc(&d, 640 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    std::vector<float> h(640); hipMemcpy(h.data(), d, 640 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++) {
        float mx = 0, sm = 0;
        for (int r = 0; r < 4; r++) { const int q = (l & 15) + 16 * r; const float v = (float)((q * 37) % 101); mx = v > mx ? v : mx; sm += v; }
        if (h[l] != mx || h[64 + l] != sm) { if (bad++ < 5) printf("rows4 lane %d: max %g (want %g) sum %g (want %g)\n", l, h[l], mx, h[64 + l], sm); }
    }
    for (int l = 0; l < 64; l++) for (int r = 0; r < 4; r++) {
        const int row = 4 * (l >> 4) + r, col = l & 15;
        float w32 = 0, w16 = 0;
        for (int kk = 0; kk < 32; kk++) w32 += (float)((row + 3 * kk) % 7) * (float)((kk + 5 * col) % 5);
        for (int kk = 0; kk < 16; kk++) w16 += (float)((row + 3 * kk) % 7) * (float)((kk + 5 * col) % 5);
        if (h[128 + l * 4 + r] != w32) { if (bad++ < 10) printf("mfma32 lane %d reg %d: %g want %g\n", l, r, h[128 + l * 4 + r], w32); }
        if (h[384 + l * 4 + r] != w16) { if (bad++ < 10) printf("mfma16 lane %d reg %d: %g want %g\n", l, r, h[384 + l * 4 + r], w16); }
    }
    printf(bad ? "FAIL %d\n" : "all ok\n", bad);
    return 0;
}
