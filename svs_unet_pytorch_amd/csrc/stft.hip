// Windowed STFT / inverse STFT on the GPU (gfx950): one 1024-point FFT per frame, held in LDS.
// Replaces librosa.stft / librosa.magphase / librosa.istft (reference data.py:79-80,100-101,159) and
// torch.istft (train.py:51-58):  periodic Hann window of n_fft samples, centred frames with zero
// padding, frames = 1 + n_samples / hop; inverse = irfft, window, overlap-add, divide by the window
// sum-of-squares, trim n_fft/2 from both ends.
//
// FFT: radix-4 Stockham autosort, 5 passes over 1024 complex points, 256 threads = 256 butterflies per
// pass, twiddles from a 1024-entry LDS table built with sincospif.  The frame is real, so this does 2x
// the minimum arithmetic; the transform is HBM/launch-bound anyway (6.6 MFLOP per 128-frame tile).
// Bound: HBM.  Algorithmic bytes per frame: hop*4 in, 513*4 (+513*8 with phase) out.
#include "common.h"

#define NFFT 1024
#define NBIN 513

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// forward DFT of buf0 (1024 complex) -> result left in the returned buffer.  tw[k] = exp(-2*pi*i*k/1024).
__device__ float2* fft1024(float2* buf0, float2* buf1, const float2* tw) {
  const int j = threadIdx.x;   // butterfly index, 0..255
  float2* in = buf0;
  float2* out = buf1;
#pragma unroll
  for (int Ns = 1; Ns < NFFT; Ns <<= 2) {
    const int k = j & (Ns - 1);
    const int tstep = k * (NFFT / (4 * Ns));
    float2 v0 = in[j];
    float2 v1 = cmul(in[j + 256], tw[tstep]);
    float2 v2 = cmul(in[j + 512], tw[2 * tstep]);
    float2 v3 = cmul(in[j + 768], tw[3 * tstep]);
    const float2 a0 = make_float2(v0.x + v2.x, v0.y + v2.y);
    const float2 a2 = make_float2(v0.x - v2.x, v0.y - v2.y);
    const float2 a1 = make_float2(v1.x + v3.x, v1.y + v3.y);
    const float2 d = make_float2(v1.x - v3.x, v1.y - v3.y);
    const float2 a3 = make_float2(d.y, -d.x);                      // (v1 - v3) * (-i)
    const int j0 = ((j - k) << 2) + k;                             // (j / Ns) * 4Ns + k
    out[j0] = make_float2(a0.x + a1.x, a0.y + a1.y);
    out[j0 + Ns] = make_float2(a2.x + a3.x, a2.y + a3.y);
    out[j0 + 2 * Ns] = make_float2(a0.x - a1.x, a0.y - a1.y);
    out[j0 + 3 * Ns] = make_float2(a2.x - a3.x, a2.y - a3.y);
    __syncthreads();
    float2* t = in; in = out; out = t;
  }
  return in;
}

__device__ __forceinline__ void build_twiddles(float2* tw) {
  for (int k = threadIdx.x; k < NFFT; k += 256) {
    float s, c;
    sincospif((float)k * (2.0f / NFFT), &s, &c);
    tw[k] = make_float2(c, -s);
  }
}

__global__ __launch_bounds__(256) void stft_fwd_kernel(const float* __restrict__ y, long n_samples, int hop, int T,
                                                       float* __restrict__ mag, float* __restrict__ phase) {
  __shared__ float2 b0[NFFT], b1[NFFT], tw[NFFT];
  build_twiddles(tw);
  __syncthreads();
  for (int t = blockIdx.x; t < T; t += gridDim.x) {
    const long start = (long)t * hop - NFFT / 2;
    for (int n = threadIdx.x; n < NFFT; n += 256) {
      const long s = start + n;
      const float w = 0.5f - 0.5f * tw[n].x;                       // periodic Hann
      const float v = (s >= 0 && s < n_samples) ? y[s] * w : 0.f;
      b0[n] = make_float2(v, 0.f);
    }
    __syncthreads();
    const float2* r = fft1024(b0, b1, tw);
    for (int f = threadIdx.x; f < NBIN; f += 256) {
      const float2 d = r[f];
      const float m = sqrtf(d.x * d.x + d.y * d.y);
      mag[(long)f * T + t] = m;
      if (phase) {
        float2 ph = (m == 0.f) ? make_float2(1.f, 0.f) : make_float2(d.x / m, d.y / m);
        *(float2*)(phase + 2 * ((long)f * T + t)) = ph;
      }
    }
    __syncthreads();
  }
}

// one frame: irfft(mag*phase) * window -> frames[t][1024]
__global__ __launch_bounds__(256) void istft_frames_kernel(const float* __restrict__ mag, const float* __restrict__ phase,
                                                           int phase_is_angle, int T, float* __restrict__ frames) {
  __shared__ float2 b0[NFFT], b1[NFFT], tw[NFFT];
  build_twiddles(tw);
  __syncthreads();
  for (int t = blockIdx.x; t < T; t += gridDim.x) {
    for (int k = threadIdx.x; k < NBIN; k += 256) {
      const long idx = (long)k * T + t;
      const float m = mag[idx];
      float re, im;
      if (phase_is_angle) {
        float s, c;
        sincosf(phase[idx], &s, &c);
        re = m * c; im = m * s;
      } else {
        const float2 ph = *(const float2*)(phase + 2 * idx);
        re = m * ph.x; im = m * ph.y;
      }
      if (k == 0 || k == NFFT / 2) im = 0.f;                       // c2r ignores these
      b0[k] = make_float2(re, -im);                                // conj(X[k])
      if (k > 0 && k < NFFT / 2) b0[NFFT - k] = make_float2(re, im);   // conj(X[N-k]) = X[k]
    }
    __syncthreads();
    const float2* r = fft1024(b0, b1, tw);
    for (int n = threadIdx.x; n < NFFT; n += 256) {
      const float w = 0.5f - 0.5f * tw[n].x;
      frames[(long)t * NFFT + n] = r[n].x * (1.0f / NFFT) * w;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void istft_ola_kernel(const float* __restrict__ frames, int hop, int T, long n_out,
                                                        float* __restrict__ y) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n_out; i += (long)gridDim.x * 256) {
    const long n = i + NFFT / 2;
    long t1 = n / hop;
    if (t1 > T - 1) t1 = T - 1;
    long t0 = n - (NFFT - 1);
    t0 = t0 <= 0 ? 0 : (t0 + hop - 1) / hop;
    float s = 0.f, env = 0.f;
    for (long t = t0; t <= t1; ++t) {
      const int o = (int)(n - t * hop);
      float sn, cs;
      sincospif((float)o * (2.0f / NFFT), &sn, &cs);
      const float w = 0.5f - 0.5f * cs;
      s += frames[t * NFFT + o];
      env += w * w;
    }
    y[i] = env > 1.1754944e-38f ? s / env : s;
  }
}

extern "C" int svs_stft_frames(int64_t n_samples, int hop) { return hop > 0 && n_samples >= 0 ? (int)(1 + n_samples / hop) : -1; }

extern "C" int svs_stft_fwd(const float* y, int64_t n_samples, int n_fft, int hop, float* mag, float* phase, hipStream_t stream) {
  SVS_REQUIRE(y && mag && n_samples > 0 && hop > 0, "svs_stft_fwd: bad arguments");
  SVS_REQUIRE(n_fft == NFFT, "svs_stft_fwd: only n_fft=1024 (reference config.py:47) is built, got %d", n_fft);
  SVS_REQUIRE(!phase || (((uintptr_t)phase) & 7u) == 0, "svs_stft_fwd: phase must be 8-byte aligned");
  const int T = (int)(1 + n_samples / hop);
  int grid = T < 2048 ? T : 2048;
  hipLaunchKernelGGL(stft_fwd_kernel, dim3(grid), dim3(256), 0, stream, y, (long)n_samples, hop, T, mag, phase);
  SVS_CHECK_LAUNCH("stft_fwd");
  return SVS_OK;
}

extern "C" size_t svs_istft_workspace_bytes(int n_fft, int hop, int frames) {
  (void)hop;
  return (size_t)frames * n_fft * sizeof(float);
}

extern "C" int svs_istft(const float* mag, const float* phase, int phase_is_angle, int n_fft, int hop, int frames, float* y,
                         void* ws, size_t ws_bytes, hipStream_t stream) {
  SVS_REQUIRE(mag && phase && y && hop > 0 && frames > 1, "svs_istft: bad arguments (need >= 2 frames)");
  SVS_REQUIRE(n_fft == NFFT, "svs_istft: only n_fft=1024 (reference config.py:47) is built, got %d", n_fft);
  SVS_REQUIRE(hop <= NFFT, "svs_istft: hop %d > n_fft leaves gaps", hop);
  if (!ws || ws_bytes < svs_istft_workspace_bytes(n_fft, hop, frames)) { svs_set_error("svs_istft: workspace too small"); return SVS_ERR_WORKSPACE; }
  int grid = frames < 2048 ? frames : 2048;
  hipLaunchKernelGGL(istft_frames_kernel, dim3(grid), dim3(256), 0, stream, mag, phase, phase_is_angle, frames, (float*)ws);
  SVS_CHECK_LAUNCH("istft_frames");
  const long n_out = (long)hop * (frames - 1);
  long g = (n_out + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(istft_ola_kernel, dim3((unsigned)g), dim3(256), 0, stream, (const float*)ws, hop, frames, n_out, y);
  SVS_CHECK_LAUNCH("istft_ola");
  return SVS_OK;
}
