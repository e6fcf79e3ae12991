// Prompt features on the device (the part that feeds this build's path): the 24 kHz mel spectrogram
// of the prompt (`feat_extractor` of cosyvoice2.yaml:152-160 = matcha.utils.audio.mel_spectrogram, third_party/Matcha-TTS/
// matcha/utils/audio.py:45-82) and the 16 kHz -> 24 kHz resampling in front of it (cli/frontend.py:497
// torchaudio.transforms.Resample).  The speech tokenizer and the speaker encoder are ONNX graphs that do not exist offline.
//
// Both kernels are table-driven: window, twiddles, mel filterbank and the polyphase resampling kernel are computed by the host
// (cv2amd/prompt.py) in float64 and handed over as device arrays, so the arithmetic here is sums of products only.
#include "common.h"
#include "../../include/cv2_amd.h"

// One block per frame.  LDS: twiddle table [n_fft] (cos, sin; fp64) + windowed frame [n_fft] (fp64) + magnitudes [n_bins].
// The DFT is evaluated directly (n_fft = 1920 is not a power of two; 961 x 1920 MACs per frame, 1.8 GFLOP per 10 s prompt):
// in fp64 throughout (window and twiddles are fp64 tables: with fp32 tables the quiet bins of a loud frame carry ~1e-3 relative
// error), the phase index k j mod n_fft kept exact in integers; the result is rounded to fp32 once.
__global__ __launch_bounds__(256) void k_melspec(cv2_melspec_cfg c, const float* __restrict__ wav, long n, float* __restrict__ out, int n_frames) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double2* tw = reinterpret_cast<double2*>(smem);                    // [n_fft]
    double* x = reinterpret_cast<double*>(tw + c.n_fft);               // [n_fft]
    float* mag = reinterpret_cast<float*>(x + c.n_fft);                // [n_bins]
    const int f = blockIdx.x, tid = threadIdx.x;
    const long pad = (c.n_fft - c.hop) / 2;
    for (int j = tid; j < c.n_fft; j += 256) {
        long i = (long)f * c.hop + j - pad;                            // reflect padding (torch.nn.functional.pad mode='reflect')
        if (i < 0) i = -i;
        if (i >= n) i = 2 * (n - 1) - i;
        x[j] = (double)wav[i] * c.window[j];
        tw[j] = reinterpret_cast<const double2*>(c.twiddle)[j];
    }
    __syncthreads();
    for (int k = tid; k < c.n_bins; k += 256) {
        double re = 0.0, im = 0.0;
        int idx = 0;
        for (int j = 0; j < c.n_fft; j++) {
            const double2 w = tw[idx];
            const double v = x[j];
            re += v * w.x;
            im -= v * w.y;
            idx += k;
            if (idx >= c.n_fft) idx -= c.n_fft;
        }
        const float r = (float)re, i2 = (float)im;
        mag[k] = sqrtf(r * r + i2 * i2 + 1e-9f);                       // audio.py:77: sqrt(spec.pow(2).sum(-1) + 1e-9) in fp32
    }
    __syncthreads();
    for (int m = tid; m < c.n_mels; m += 256) {
        float s = 0.f;
        const float* fb = c.mel_fb + (size_t)m * c.n_bins;
        for (int k = c.fb_lo[m]; k < c.fb_hi[m]; k++) s = fmaf(fb[k], mag[k], s);
        out[(size_t)f * c.n_mels + m] = logf(fmaxf(s, c.clamp_min));   // audio.py:24: log(clamp(x, min=1e-5))
    }
}

extern "C" int cv2_melspec(const cv2_melspec_cfg* c, const float* wav, int64_t n, float* out, int32_t n_frames, void* stream) {
    CV2_CHECK(c && wav && out, "cv2_melspec: null argument");
    CV2_CHECK(c->n_fft >= 16 && c->n_fft <= 2048 && c->hop >= 1 && c->hop <= c->n_fft && (c->n_fft - c->hop) % 2 == 0,
              "cv2_melspec: n_fft=%d hop=%d unsupported", c->n_fft, c->hop);
    CV2_CHECK(c->n_bins == c->n_fft / 2 + 1 && c->n_mels >= 1 && c->n_mels <= 256, "cv2_melspec: n_bins=%d n_mels=%d", c->n_bins, c->n_mels);
    CV2_CHECK(c->window && c->twiddle && c->mel_fb && c->fb_lo && c->fb_hi, "cv2_melspec: null table");
    const long pad = (c->n_fft - c->hop) / 2;
    CV2_CHECK(n > pad, "cv2_melspec: %lld samples are not more than the reflect padding %ld", (long long)n, pad);
    const long want = 1 + (n + 2 * pad - c->n_fft) / c->hop;
    CV2_CHECK(n + 2 * pad >= c->n_fft && n_frames == want, "cv2_melspec: n_frames=%d, %lld samples give %ld", n_frames, (long long)n, want);
    const size_t sm = (size_t)c->n_fft * 24 + (size_t)c->n_bins * 4;
    hipLaunchKernelGGL(k_melspec, dim3(n_frames), dim3(256), sm, (hipStream_t)stream, *c, wav, (long)n, out, (int)n_frames);
    CV2_LAUNCH_CHECK();
    return 0;
}

// Polyphase resampling by up / down (torchaudio.functional.resample's convolution, functional.py `_apply_sinc_resample_kernel`):
// out[i * up + p] = sum_j kernel[p][j] * xpad[i * down + j], xpad = zeros(pad_left) ++ in ++ zeros.
__global__ __launch_bounds__(256) void k_resample(const float* __restrict__ in, long n_in, const float* __restrict__ kernel, int up, int down,
                                                  int klen, int pad_left, float* __restrict__ out, long n_out) {
    const long o = (long)blockIdx.x * 256 + threadIdx.x;
    if (o >= n_out) return;
    const long i = o / up;
    const int p = (int)(o - i * up);
    const float* kp = kernel + (size_t)p * klen;
    const long base = i * down - pad_left;
    float s = 0.f;
    for (int j = 0; j < klen; j++) {
        const long q = base + j;
        if (q >= 0 && q < n_in) s = fmaf(kp[j], in[q], s);
    }
    out[o] = s;
}

extern "C" int cv2_resample(const float* in, int64_t n_in, const float* kernel, int32_t up, int32_t down, int32_t klen, int32_t pad_left,
                            float* out, int64_t n_out, void* stream) {
    CV2_CHECK(in && kernel && out, "cv2_resample: null argument");
    CV2_CHECK(up >= 1 && down >= 1 && klen >= 1 && klen <= 4096 && pad_left >= 0 && n_in >= 1, "cv2_resample: bad shape");
    CV2_CHECK(n_out >= 1 && n_out <= (n_in * up + down - 1) / down, "cv2_resample: n_out=%lld exceeds ceil(%lld * %d / %d)", (long long)n_out,
              (long long)n_in, up, down);
    hipLaunchKernelGGL(k_resample, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, (long)n_in, kernel, (int)up,
                       (int)down, (int)klen, (int)pad_left, out, (long)n_out);
    CV2_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------------------
// The two feature extractors in front of the ONNX prompt models (cli/frontend.py:262-283), same table-driven scheme:
//   whisper.log_mel_spectrogram(speech, n_mels=128)   (openai-whisper audio.py: hann(400) STFT, hop 160, center / reflect, |X|^2 without the
//                                                      last frame, Slaney mel filters, log10, clamp to max - 8, (x + 4) / 4)
//   torchaudio.compliance.kaldi.fbank(speech, num_mel_bins=80, dither=0, sample_frequency=16000)
//                                                     (snip_edges frames of 400 at stride 160, DC removal, pre-emphasis 0.97, povey window,
//                                                      zero-padded 512-point FFT, |X|^2, kaldi mel banks 20 Hz .. Nyquist, log(max(., eps)))
// One block per frame: frame -> (DC removal) -> (pre-emphasis) -> window -> DFT by its definition in fp64 (win <= n_fft, the frame is
// zero-padded to n_fft) -> power -> sparse mel filters -> log.  Output time-major [frames][n_mels].
__global__ __launch_bounds__(256) void k_framefeat(cv2_framefeat_cfg c, const float* __restrict__ wav, long n, float* __restrict__ out, int n_frames) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double2* tw = reinterpret_cast<double2*>(smem);                    // [n_fft]
    double* x = reinterpret_cast<double*>(tw + c.n_fft);               // [win]
    double* y = x + c.win;                                             // [win]
    float* pw = reinterpret_cast<float*>(y + c.win);                   // [n_bins]
    __shared__ double red[4];
    const int f = blockIdx.x, tid = threadIdx.x;
    for (int j = tid; j < c.n_fft; j += 256) tw[j] = reinterpret_cast<const double2*>(c.twiddle)[j];
    double s = 0.0;
    for (int j = tid; j < c.win; j += 256) {
        long i = (long)f * c.hop + j - (c.center ? c.n_fft / 2 : 0);
        if (i < 0) i = -i;                                             // reflect padding of torch.stft(center=True)
        if (i >= n) i = 2 * (n - 1) - i;
        const double v = (double)wav[i];
        x[j] = v;
        s += v;
    }
    if (c.remove_dc) {                                                 // kaldi.py _get_window: strided_input -= row mean
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if ((tid & 63) == 0) red[tid >> 6] = s;
    }
    __syncthreads();
    const double mean = c.remove_dc ? ((red[0] + red[1]) + (red[2] + red[3])) / (double)c.win : 0.0;
    for (int j = tid; j < c.win; j += 256) {
        const double v = x[j] - mean, p = x[j > 0 ? j - 1 : 0] - mean;  // pre-emphasis with the first sample replicated (kaldi.py: pad mode 'replicate')
        y[j] = (v - (double)c.preemph * p) * c.window[j];
    }
    __syncthreads();
    for (int k = tid; k < c.n_bins; k += 256) {
        double re = 0.0, im = 0.0;
        int idx = 0;
        for (int j = 0; j < c.win; j++) {
            const double2 w = tw[idx];
            const double v = y[j];
            re += v * w.x;
            im -= v * w.y;
            idx += k;
            if (idx >= c.n_fft) idx -= c.n_fft;
        }
        const float r = (float)re, i2 = (float)im;
        pw[k] = r * r + i2 * i2;                                       // both references square the fp32 spectrum
    }
    __syncthreads();
    for (int m = tid; m < c.n_mels; m += 256) {
        float acc = 0.f;
        const float* fb = c.mel_fb + (size_t)m * c.n_bins;
        for (int k = c.fb_lo[m]; k < c.fb_hi[m]; k++) acc = fmaf(fb[k], pw[k], acc);
        acc = fmaxf(acc, c.floor);
        out[(size_t)f * c.n_mels + m] = c.log10 ? log10f(acc) : logf(acc);
    }
}

extern "C" int cv2_framefeat(const cv2_framefeat_cfg* c, const float* wav, int64_t n, float* out, int32_t n_frames, void* stream) {
    CV2_CHECK(c && wav && out, "cv2_framefeat: null argument");
    CV2_CHECK(c->n_fft >= 16 && c->n_fft <= 2048 && c->win >= 1 && c->win <= c->n_fft && c->hop >= 1, "cv2_framefeat: n_fft=%d win=%d hop=%d unsupported",
              c->n_fft, c->win, c->hop);
    CV2_CHECK(c->n_bins >= 1 && c->n_bins <= c->n_fft / 2 + 1 && c->n_mels >= 1 && c->n_mels <= 256, "cv2_framefeat: n_bins=%d n_mels=%d", c->n_bins, c->n_mels);
    CV2_CHECK(c->window && c->twiddle && c->mel_fb && c->fb_lo && c->fb_hi, "cv2_framefeat: null table");
    long want;
    if (c->center) {                                                   // torch.stft(center=True) minus the last frame (whisper audio.py: stft[..., :-1])
        CV2_CHECK(n > c->n_fft / 2, "cv2_framefeat: %lld samples are not more than the reflect padding %d", (long long)n, c->n_fft / 2);
        want = n / c->hop;
    } else {                                                           // kaldi snip_edges
        want = n < c->win ? 0 : 1 + (n - c->win) / c->hop;
    }
    CV2_CHECK(n_frames >= 1 && n_frames == want, "cv2_framefeat: n_frames=%d, %lld samples give %ld", n_frames, (long long)n, want);
    const size_t sm = (size_t)c->n_fft * 16 + (size_t)c->win * 16 + (size_t)c->n_bins * 4;
    hipLaunchKernelGGL(k_framefeat, dim3(n_frames), dim3(256), sm, (hipStream_t)stream, *c, wav, (long)n, out, (int)n_frames);
    CV2_LAUNCH_CHECK();
    return 0;
}

// whisper audio.py: log_spec = maximum(log_spec, log_spec.max() - 8.0); log_spec = (log_spec + 4.0) / 4.0 ; in [T][M] -> out [M][T]
__global__ __launch_bounds__(1024) void k_whisper_post(const float* __restrict__ in, float* __restrict__ out, int T, int M) {
    __shared__ float red[16];
    float mx = -INFINITY;
    for (long i = threadIdx.x; i < (long)T * M; i += 1024) mx = fmaxf(mx, in[i]);
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = red[0];
    for (int w = 1; w < 16; w++) mx = fmaxf(mx, red[w]);
    for (long i = threadIdx.x; i < (long)T * M; i += 1024) {
        const int t = (int)(i / M), m = (int)(i - (long)t * M);
        out[(size_t)m * T + t] = (fmaxf(in[i], mx - 8.0f) + 4.0f) / 4.0f;
    }
}
extern "C" int cv2_whisper_post(const float* in, float* out, int32_t n_frames, int32_t n_mels, void* stream) {
    CV2_CHECK(in && out && n_frames >= 1 && n_mels >= 1, "cv2_whisper_post: bad argument");
    hipLaunchKernelGGL(k_whisper_post, dim3(1), dim3(1024), 0, (hipStream_t)stream, in, out, (int)n_frames, (int)n_mels);
    CV2_LAUNCH_CHECK();
    return 0;
}

// cli/frontend.py:278: feat = feat - feat.mean(dim=0, keepdim=True) ; x [T][M] in place, one block per column
__global__ __launch_bounds__(256) void k_sub_col_mean(float* __restrict__ x, int T, int M) {
    __shared__ float red[4];
    const int m = blockIdx.x;
    float s = 0.f;
    for (int t = threadIdx.x; t < T; t += 256) s += x[(size_t)t * M + m];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float mean = ((red[0] + red[1]) + (red[2] + red[3])) / (float)T;
    for (int t = threadIdx.x; t < T; t += 256) x[(size_t)t * M + m] -= mean;
}
extern "C" int cv2_sub_col_mean(float* x, int32_t n_rows, int32_t n_cols, void* stream) {
    CV2_CHECK(x && n_rows >= 1 && n_cols >= 1, "cv2_sub_col_mean: bad argument");
    hipLaunchKernelGGL(k_sub_col_mean, dim3(n_cols), dim3(256), 0, (hipStream_t)stream, x, (int)n_rows, (int)n_cols);
    CV2_LAUNCH_CHECK();
    return 0;
}
