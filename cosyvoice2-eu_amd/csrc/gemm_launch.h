// Host-side launch helpers of k_gemm (gemm.h), shared by flow.hip and the LLM prefill in llm.hip.
#pragma once
#include "gemm.h"

// a GEMM whose epilogue only stores bf16: k_gemm's EPI = 1 (CV2_GEMM_PLAIN=0, diagnostics: the general epilogue everywhere)
static bool gemm_plain(const GemmArgs& a) {
    static const bool off = getenv("CV2_GEMM_PLAIN") && getenv("CV2_GEMM_PLAIN")[0] == '0';
    return !off && !a.bias && !a.ln1_g && !a.ln2_g && a.act == ACT_NONE && !a.rowadd && !a.res && !a.out_f32 && a.out_bf16 && !a.kvc;
}
template <int BM, int BN, int WM, int WN, bool SPLITA = false, int NSTAGE = 2, bool PLAIN_OK = false>
static int gemm_go(const GemmArgs& a, int batch, bool packed, hipStream_t s) {
    constexpr size_t sm = gemm_smem_bytes<BM, BN, SPLITA, NSTAGE>();
    dim3 grid(a.N / BN, a.M / BM, batch), block(WM * WN * 64);
    if (PLAIN_OK && packed && gemm_plain(a)) {
        static std::atomic<bool> once{false};
        if (!once) { CV2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm<BM, BN, WM, WN, true, SPLITA, NSTAGE, PLAIN_OK ? 1 : 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm)); once = true; }
        hipLaunchKernelGGL((k_gemm<BM, BN, WM, WN, true, SPLITA, NSTAGE, PLAIN_OK ? 1 : 0>), grid, block, sm, s, a);
    } else if (packed) {
        static std::atomic<bool> once{false};    // idempotent attribute call: a second thread repeats it rather than launch before it is in place
        if (!once) { CV2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm<BM, BN, WM, WN, true, SPLITA, NSTAGE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm)); once = true; }
        hipLaunchKernelGGL((k_gemm<BM, BN, WM, WN, true, SPLITA, NSTAGE>), grid, block, sm, s, a);
    } else if (!SPLITA) {
        static std::atomic<bool> once{false};    // idempotent attribute call: a second thread repeats it rather than launch before it is in place
        if (!once) { CV2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm<BM, BN, WM, WN, false, false, NSTAGE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm)); once = true; }
        hipLaunchKernelGGL((k_gemm<BM, BN, WM, WN, false, false, NSTAGE>), grid, block, sm, s, a);
    } else {
        return cv2_fail("gemm: the split-A kernels need packed weights");
    }
    CV2_LAUNCH_CHECK();
    return 0;
}

template <int KS_T>
static int gemm_go_panel_k(const GemmArgs& a, int batch, hipStream_t s) {
    constexpr int CH = 8;
    constexpr size_t panel = (size_t)KS_T * 1024, ctile = (size_t)16 * (256 + 4) * 4;
    constexpr size_t sm = (panel > ctile ? panel : ctile) + 5 * 1024;      // + the five epilogue vectors
    static std::atomic<bool> once{false};    // idempotent attribute call: a second thread repeats it rather than launch before it is in place
    if (!once) { CV2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_panel<CH, KS_T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm)); once = true; }
    hipLaunchKernelGGL((k_gemm_panel<CH, KS_T>), dim3(a.N / 256, a.M / 16, batch), dim3(1024), sm, s, a);
    CV2_LAUNCH_CHECK();
    return 0;
}
// the K values of the flow estimator's N = 256 layers (256 .. 1536); anything else stays on k_gemm
static bool gemm_panel_has(int K) { return K == 256 || K == 512 || K == 768 || K == 1024 || K == 1536; }
static int gemm_go_panel(const GemmArgs& a, int batch, hipStream_t s) {
    switch (a.K) {
        case 256: return gemm_go_panel_k<8>(a, batch, s);
        case 512: return gemm_go_panel_k<16>(a, batch, s);
        case 768: return gemm_go_panel_k<24>(a, batch, s);
        case 1024: return gemm_go_panel_k<32>(a, batch, s);
        default: return gemm_go_panel_k<48>(a, batch, s);
    }
}

// many rows: 64-row blocks (M is a multiple of 128)
static int tail_rows_go(const TailArgs& a, int M, hipStream_t s) {
    constexpr size_t sm = tail_rows_smem<4>();
    static std::atomic<bool> once{false};    // idempotent attribute call: a second thread repeats it rather than launch before it is in place
    if (!once) { CV2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tail_rows<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm)); once = true; }
    hipLaunchKernelGGL(k_tail_rows<4>, dim3(1, M / 64, 1), dim3(1024), sm, s, a);
    CV2_LAUNCH_CHECK();
    return 0;
}
// round 5: 8 waves x (32 columns x 64 rows), one weight stream through the block's GEMMs; qkv: the next block's QKV projection chained on
static int tail_rows2_go(const TailArgs& a, int M, bool qkv, hipStream_t s) {
    constexpr size_t sm = tail_rows2_smem();
    static std::atomic<bool> once{false};
    if (!once) {
        CV2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tail_rows2<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
        CV2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tail_rows2<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
        once = true;
    }
    if (qkv) hipLaunchKernelGGL(k_tail_rows2<true>, dim3(1, M / 64, 1), dim3(512), sm, s, a);
    else hipLaunchKernelGGL(k_tail_rows2<false>, dim3(1, M / 64, 1), dim3(512), sm, s, a);
    CV2_LAUNCH_CHECK();
    return 0;
}
static int tail_panel_go(const TailArgs& a, int M, hipStream_t s) {
    constexpr size_t sm = tail_panel_smem();
    static std::atomic<bool> once{false};    // idempotent attribute call: a second thread repeats it rather than launch before it is in place
    if (!once) {
        CV2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tail_panel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
        CV2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tail_panel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
        CV2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tail_panel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
        once = true;
    }
    // fewer panels than CUs: split the feed-forward's hidden columns over 2 or 4 workgroups per panel (needs the scratch of TailArgs)
    static const int split_env = getenv("CV2_FLOW_TAIL_SPLIT") ? atoi(getenv("CV2_FLOW_TAIL_SPLIT")) : -1;      // A/B switch (diagnostics): 1 / 2 / 4
    const int panels = (M - a.row0) / 16;
    int S = a.part ? (panels <= 64 ? 4 : (panels <= 128 ? 2 : 1)) : 1;
    // the scratch holds 128 panels x 4 parts and 128 tickets: a forced split only where it fits, else the default for this panel count
    if (split_env > 0 && a.part && (split_env == 1 || ((split_env == 2 || split_env == 4) && panels <= 128))) S = split_env;
    if (S == 4) hipLaunchKernelGGL(k_tail_panel<4>, dim3(4, panels, 1), dim3(1024), sm, s, a);
    else if (S == 2) hipLaunchKernelGGL(k_tail_panel<2>, dim3(2, panels, 1), dim3(1024), sm, s, a);
    else hipLaunchKernelGGL(k_tail_panel<1>, dim3(1, panels, 1), dim3(1024), sm, s, a);
    CV2_LAUNCH_CHECK();
    return 0;
}

// cfg 0: 128x128; cfg 1: 64x256 / 32x256 (whole rows of N == 256); cfg 2: 128x64; cfg 4: 128x128 with the hi/lo operand split
static int gemm_launch_cfg(const GemmArgs& a, int cfg, int batch, bool packed, hipStream_t s) {
    CV2_CHECK(a.K % 64 == 0 && a.K > 0, "gemm: K=%d must be a positive multiple of 64", a.K);
    CV2_CHECK(a.M % 128 == 0 && a.M > 0, "gemm: M=%d must be a positive multiple of 128", a.M);
    if (cfg == 0) { CV2_CHECK(a.N % 128 == 0, "gemm cfg0: N=%d %% 128", a.N); 
        // fewer blocks than ~1.5 per CU: 16 waves per block spread the LDS-DMA issue cost (phase stamps: the K loop runs at ~2x the per-CU
        // vector-memory bound of 64 B/clk whatever the number of stages in flight); otherwise 8 waves and two blocks per CU
        if ((long)(a.N / 128) * (a.M / 128) * batch < 200) return gemm_go<64, 128, 2, 4, false, 2, true>(a, batch, packed, s);      // < 1 block per CU: halve the tile, the row epilogue (GELU, V^T) is VALU-bound
        if ((long)(a.N / 128) * (a.M / 128) * batch < 400) return gemm_go<128, 128, 4, 4, false, 2, true>(a, batch, packed, s);
        return gemm_go<128, 128, 2, 4, false, 2, true>(a, batch, packed, s);
    }
    if (cfg == 1) {
        CV2_CHECK(a.N % 256 == 0, "gemm cfg1: N=%d %% 256", a.N);
        CV2_CHECK((!a.ln1_g && !a.ln2_g) || a.N == 256, "gemm cfg1: LayerNorm epilogue needs N == 256");
        // few rows (one utterance): 32-row tiles double the blocks that share the latency-bound K loop and the row epilogue
        if ((long)(a.M / 64) * (a.N / 256) * batch < 200) {
            if (packed && gemm_panel_has(a.K) && !a.vt && !a.A_lo) return gemm_go_panel(a, batch, s);     // row-panel kernel
            return gemm_go<32, 256, 2, 8>(a, batch, packed, s);
        }
        return gemm_go<64, 256, 2, 4>(a, batch, packed, s);
    }
    if (cfg == 4) {
        CV2_CHECK(a.N % 128 == 0 && a.A_lo, "gemm cfg4: N=%d %% 128, A_lo required", a.N);
        // one utterance's prefill (3 row tiles of 128): the N = 896 / 1152 layers give 21-27 blocks; 64-row tiles double them
        if ((long)(a.N / 128) * (a.M / 128) * batch < 128) return gemm_go<64, 128, 2, 4, true>(a, batch, packed, s);
        return gemm_go<128, 128, 2, 4, true>(a, batch, packed, s);
    }
    CV2_CHECK(a.N % 64 == 0, "gemm cfg2: N=%d %% 64", a.N);
    return gemm_go<128, 64, 2, 4>(a, batch, packed, s);
}

static GemmArgs gemm_args(const uint16_t* A, long lda, long a_off, const uint16_t* W, int M, int N, int K) {
    GemmArgs a{};
    a.A = A; a.lda = lda; a.a_row_off = a_off; a.W = W; a.M = M; a.N = N; a.K = K; a.M_valid = M;
    a.out_scale = 1.f; a.n_store = N; a.ln2_scale = 1.f; a.act_slope = 0.01f;
    return a;
}

