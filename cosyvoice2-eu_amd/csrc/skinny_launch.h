// Host entry of the skinny weight-streaming GEMM (defined in llm.hip), shared with the flow time-embedding MLP.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
// out[r][n] = sum_k W[n][k] x[r][k] (+ bias[n]); rows <= 32, W packed, n % 16 == 0, k % 32 == 0, k <= 1024
int skinny_gemm_launch(const uint16_t* w, const float* bias, const float* x, float* out, int rows, int n, int k, hipStream_t s);
