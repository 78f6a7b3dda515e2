// Windowed STFT / inverse STFT on the GPU (gfx950) as bandwidth kernels.
// Replaces librosa.stft / librosa.magphase / librosa.istft (reference data.py:79-80,100-101,159) and torch.istft
// (train.py:51-58): periodic Hann window of n_fft = 1024 samples, centred frames with zero padding,
// frames = 1 + n_samples / hop; inverse = irfft, window, overlap-add, divide by the window sum-of-squares, trim
// n_fft/2 from both ends.
//
// Structure (both directions): a 512-thread block owns 16 consecutive frames (hops) of one channel.
//   * FFT: one 1024-point complex FFT per WAVE in LDS (fft_wave.h: three register-radix passes, no block barriers),
//     and TWO real frames per transform (frame t in the real part, frame t+1 in the imaginary part; the two spectra
//     separate by Hermitian symmetry) -- half the arithmetic of a complex transform per real frame.
//   * Layout: spectrograms are f-major with time fastest ((513, T) files of data.py:107-109, (tiles, 1, 512, 128)
//     network tiles of inference.py:84, (B, 1, 512, 128) training tiles of train.py:138).  A frame is a COLUMN of
//     them, so the 16 frames of a block go through an LDS transpose and every row is written / read as one 64-byte
//     run (the first version did 513 scattered 4-byte accesses per frame).  `SpecLayout` addresses all three forms,
//     DC row dropped or not, so the forward transform writes network tiles directly and the inverse reads them.
//   * Inverse: the 17 frames that touch the block's 16 hops are overlap-added in LDS (even frames, then odd frames:
//     fixed order, no atomics), divided by the window envelope and written once -- no frame buffer in HBM.
//   * Fused around them: mask application (inference.py:100-107) on the inverse's input, per-block |.|max partials
//     for the normalisations of data.py:84-85,162-164, and -- for the differentiable inverse of train.py:33-60 --
//     the transposed operator (`SINK_DMAG`) that maps d(loss)/d(waveform) back to d(loss)/d(mask logit).
// Bound: HBM.  Algorithmic bytes per frame: hop*4 in, 513*4 (+513*8 with phase) out, and the reverse.
#include "internal.h"
#include "fft_wave.h"

#define NFFT 1024
#define NBIN 513
#define GROUP 16                  // frames (hops) per block
#define SROW (GROUP + 1)          // floats per staged row (17: odd -> conflict-free column access)
#define IROW 19                   // inverse: 17 frames per row, odd pitch

// Address of (channel c, bin k, frame t) of an f-major spectrogram cut into `seg`-frame tiles of `rows` rows each,
// the first of which is bin `first_bin`:  (513, T) file: seg = T, rows = 513, first_bin = 0;
// network tiles (n, 1, 512, 128): seg = 128, rows = 512, first_bin = 1 (DC row dropped, inference.py:68 / train.py:109).
struct SpecLayout {
  long chan_stride; int seg, rows, first_bin, frames_alloc;      // frames_alloc: columns that exist (>= T: tile padding)
  __device__ __forceinline__ long at(int c, int k, int t) const {
    const int tile = t / seg;
    return (long)c * chan_stride + ((long)tile * rows + (k - first_bin)) * seg + (t - tile * seg);
  }
};

__device__ __forceinline__ float hann_at(int m) { return 0.5f - 0.5f * cospif((float)m * (2.0f / NFFT)); }

// window sum-of-squares at padded position q (= sample index + n_fft/2) for T frames of hop `hop`
__device__ __forceinline__ float envelope_at(long q, int hop, int T, const float* win) {
  long t1 = q / hop;
  if (t1 > T - 1) t1 = T - 1;
  long t0 = q - (NFFT - 1);
  t0 = t0 <= 0 ? 0 : (t0 + hop - 1) / hop;
  float env = 0.f;
  for (long t = t0; t <= t1; ++t) { const float w = win[q - t * hop]; env += w * w; }
  return env;
}

// ------------------------------------------------------------------------------------------------
// forward: frames of a real signal -> bins
// ------------------------------------------------------------------------------------------------
enum { SRC_SIGNAL = 0, SRC_ENVDIV = 1 };        // x[s] zero-padded | d_wav[s] / envelope (transpose of the inverse's tail)
enum { SINK_MAGPHASE = 0, SINK_DMAG = 1 };
struct StftArgs {
  const float* y; long n_samples; int channels; int hop; int T;     // T frames per channel
  float* mag; SpecLayout lay;                                       // SINK_MAGPHASE
  float* phase; int phase_mode;                                     // 0 none, 1 frame-major [c][t][513] float2, 2 f-major (513, T) float2
  float* absmax_partial;                                            // [gridDim.y * gridDim.x] or null
  // SINK_DMAG: d_logit[idx] += alpha * d|S| * mix[idx] * mask[idx] * (1 - mask[idx]), all in layout `lay`
  const float* angle; const float* mix; const float* mask; float* d_logit; float alpha;
};

template <int SRC, int SINK>
__global__ __launch_bounds__(512) void stft_fwd_kernel(StftArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int BUF = FftSize<NFFT>::BUF, TW = FftSize<NFFT>::TW;
  float2* const fbuf = (float2*)smem;                       // [8][BUF]
  float2* const tw = fbuf + 8 * BUF;                        // [TW]
  float* const win = (float*)(tw + TW);                     // [NFFT]
  float* const stage = win + NFFT;                          // [513 * 9 * 2] (>= 513 * 17)
  float* const angs = stage + NBIN * 18;                    // SINK_DMAG only: [513 * 17]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = blockIdx.y, t0 = blockIdx.x * GROUP;
  fft_build_twiddles<NFFT>(tw, tid, 512);
  for (int m = tid; m < NFFT; m += 512) win[m] = hann_at(m);
  if (SINK == SINK_DMAG) {
    for (int e = tid; e < NBIN * GROUP; e += 512) {
      const int k = e / GROUP, col = e - k * GROUP, t = t0 + col;
      angs[k * SROW + col] = (k >= p.lay.first_bin && t < p.T) ? p.angle[p.lay.at(c, k, t)] : 0.f;
    }
  }
  __syncthreads();
  float2* const buf = fbuf + wave * BUF;
  const int ta = t0 + 2 * wave, tb = ta + 1;
  // ---- frames ta, tb -> z = a + i b (windowed)
#pragma unroll 4
  for (int r = 0; r < NFFT / 64; ++r) {
    const int m = lane + 64 * r;
    float v[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int t = ta + h;
      const long q = (long)t * p.hop + m;                 // padded position; sample index = q - NFFT/2
      const long s = q - NFFT / 2;
      float x = 0.f;
      if (t < p.T && s >= 0 && s < p.n_samples) {
        x = p.y[(long)c * p.n_samples + s];
        if (SRC == SRC_ENVDIV) {
          const float env = envelope_at(q, p.hop, p.T, win);
          if (env > 1.1754944e-38f) x /= env;
        }
      }
      v[h] = x * win[m];
    }
    buf[fft_pad(m)] = float2{v[0], v[1]};
  }
  fft_wave<NFFT>(buf, tw, lane);
  // ---- separate the two spectra, bins k = 0 .. 512
  float vmax = 0.f;
#pragma unroll
  for (int r = 0; r < 9; ++r) {
    const int k = lane + 64 * r;
    if (k > NFFT / 2) continue;
    const float2 zk = buf[fft_pad(k)], zn = buf[fft_pad((NFFT - k) & (NFFT - 1))];
    const float2 A = float2{0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y)};
    const float2 B = float2{0.5f * (zk.y + zn.y), 0.5f * (zn.x - zk.x)};
    if (SINK == SINK_MAGPHASE) {
      const float ma = sqrtf(A.x * A.x + A.y * A.y), mb = sqrtf(B.x * B.x + B.y * B.y);
      stage[k * SROW + 2 * wave] = ma;
      stage[k * SROW + 2 * wave + 1] = mb;
      if (ta < p.T) vmax = fmaxf(vmax, ma);
      if (tb < p.T) vmax = fmaxf(vmax, mb);
      if (p.phase_mode == 1) {                             // unit phasors, 1 + 0i where the bin is exactly zero (librosa.magphase)
        float2* ph = (float2*)p.phase;
        if (ta < p.T) ph[((long)c * p.T + ta) * NBIN + k] = ma == 0.f ? float2{1.f, 0.f} : float2{A.x / ma, A.y / ma};
        if (tb < p.T) ph[((long)c * p.T + tb) * NBIN + k] = mb == 0.f ? float2{1.f, 0.f} : float2{B.x / mb, B.y / mb};
      }
    } else {
      // transpose of irfft (1/N, bins 1..511 count twice, imaginary parts of DC / Nyquist are ignored by irfft), then
      // d|S| = Re(conj(e^{i phi}) G)
      const float ck = (k == 0 || k == NFFT / 2) ? 1.0f / NFFT : 2.0f / NFFT;
      float sa, ca, sb, cb;
      sincosf(angs[k * SROW + 2 * wave], &sa, &ca);
      sincosf(angs[k * SROW + 2 * wave + 1], &sb, &cb);
      const bool edge = (k == 0 || k == NFFT / 2);
      stage[k * SROW + 2 * wave] = ck * (A.x * ca + (edge ? 0.f : A.y * sa));
      stage[k * SROW + 2 * wave + 1] = ck * (B.x * cb + (edge ? 0.f : B.y * sb));
    }
  }
  __syncthreads();
  // ---- rows out: 16 consecutive frames of a bin = one 64-byte run
  for (int e = tid; e < NBIN * GROUP; e += 512) {
    const int k = e / GROUP, col = e - k * GROUP, t = t0 + col;
    if (k < p.lay.first_bin || t >= p.lay.frames_alloc) continue;
    const float v = t < p.T ? stage[k * SROW + col] : 0.f;       // tile padding beyond the last frame is written as zeros
    const long idx = p.lay.at(c, k, t);
    if (SINK == SINK_MAGPHASE) p.mag[idx] = v;
    else if (t < p.T) { const float m = p.mask[idx]; p.d_logit[idx] += p.alpha * v * p.mix[idx] * m * (1.f - m); }
  }
  if (SINK == SINK_MAGPHASE && p.phase_mode == 2) {
    // f-major phasors (the .npy layout of data.py:108-109): two more staged rounds of 8 frames each, as float2
    float2* const st2 = (float2*)stage;                     // [513][9]
    float2* ph = (float2*)p.phase;
    for (int half = 0; half < 2; ++half) {
      __syncthreads();
      if ((wave >> 2) == half) {
#pragma unroll
        for (int r = 0; r < 9; ++r) {
          const int k = lane + 64 * r;
          if (k > NFFT / 2) continue;
          const float2 zk = buf[fft_pad(k)], zn = buf[fft_pad((NFFT - k) & (NFFT - 1))];
          const float2 A = float2{0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y)};
          const float2 B = float2{0.5f * (zk.y + zn.y), 0.5f * (zn.x - zk.x)};
          const float ma = sqrtf(A.x * A.x + A.y * A.y), mb = sqrtf(B.x * B.x + B.y * B.y);
          st2[k * 9 + 2 * (wave & 3)] = ma == 0.f ? float2{1.f, 0.f} : float2{A.x / ma, A.y / ma};
          st2[k * 9 + 2 * (wave & 3) + 1] = mb == 0.f ? float2{1.f, 0.f} : float2{B.x / mb, B.y / mb};
        }
      }
      __syncthreads();
      for (int e = tid; e < NBIN * 8; e += 512) {
        const int k = e >> 3, col = e & 7, t = t0 + 8 * half + col;
        if (t < p.T) ph[((long)c * NBIN + k) * p.T + t] = st2[k * 9 + col];
      }
    }
  }
  if (p.absmax_partial) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
    __syncthreads();
    if (lane == 0) stage[wave] = vmax;
    __syncthreads();
    if (tid == 0) {
      float m = stage[0];
      for (int w = 1; w < 8; ++w) m = fmaxf(m, stage[w]);
      p.absmax_partial[(long)blockIdx.y * gridDim.x + blockIdx.x] = m;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// inverse: bins -> overlap-added signal
// ------------------------------------------------------------------------------------------------
struct IstftArgs {
  const float* mag; SpecLayout lay;
  const float* mask; int invert;                  // optional: |S| = mag * mask (or * (1 - mask)), inference.py:100-107
  const float* phase; int phase_mode;             // 1 frame-major phasors [c][t][513] float2, 3 angles in layout `lay`
  int channels, T, hop;
  float* y; long n_out;                           // (channels, hop * (T - 1))
  float* absmax_partial;
};

__global__ __launch_bounds__(512) void istft_kernel(IstftArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int BUF = FftSize<NFFT>::BUF, TW = FftSize<NFFT>::TW;
  float2* const fbuf = (float2*)smem;                       // [8][BUF]
  float2* const tw = fbuf + 8 * BUF;
  float* const win = (float*)(tw + TW);
  float* const mags = win + NFFT;                           // [513 * 19]; later the overlap-add buffer [16 * hop <= 16 * 1024]
  float* const angs = mags + NBIN * IROW;                   // [513 * 19] (phase_mode 3)
  float* const ola = mags;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = blockIdx.y, t0 = blockIdx.x * GROUP;        // owns padded samples [hop * t0, hop * (t0 + 16)): frames t0-1 .. t0+15
  const int NF = GROUP + 1;
  fft_build_twiddles<NFFT>(tw, tid, 512);
  for (int m = tid; m < NFFT; m += 512) win[m] = hann_at(m);
  for (int e = tid; e < NBIN * NF; e += 512) {
    const int k = e / NF, f = e - k * NF, t = t0 - 1 + f;
    float m = 0.f, a = 0.f;
    if (k >= p.lay.first_bin && t >= 0 && t < p.T) {
      const long idx = p.lay.at(c, k, t);
      m = p.mag[idx];
      if (p.mask) { const float mk = p.mask[idx]; m *= p.invert ? 1.f - mk : mk; }
      if (p.phase_mode == 3) a = p.phase[idx];
    }
    mags[k * IROW + f] = m;
    if (p.phase_mode == 3) angs[k * IROW + f] = a;
  }
  __syncthreads();
  float2* const buf = fbuf + wave * BUF;
  // spectrum value of local frame f, bin k (imaginary parts of DC / Nyquist dropped, as irfft does)
  auto spec = [&](int f, int k) -> float2 {
    const int t = t0 - 1 + f;
    const float m = mags[k * IROW + f];
    float2 s;
    if (p.phase_mode == 3) { float sn, cs; sincosf(angs[k * IROW + f], &sn, &cs); s = float2{m * cs, m * sn}; }
    else if (t >= 0 && t < p.T) { const float2 ph = ((const float2*)p.phase)[((long)c * p.T + t) * NBIN + k]; s = float2{m * ph.x, m * ph.y}; }
    else s = float2{0.f, 0.f};
    if (k == 0 || k == NFFT / 2) s.y = 0.f;
    return s;
  };
  // conj(Z) with Z = Sa + i Sb (Hermitian-extended): the forward transform of conj(Z) is conj(ifft(Z)) = a - i b
  auto build = [&](int fa, bool has_b) {
#pragma unroll
    for (int r = 0; r < 9; ++r) {
      const int k = lane + 64 * r;
      if (k > NFFT / 2) continue;
      const float2 sa = spec(fa, k);
      const float2 sb = has_b ? spec(fa + 1, k) : float2{0.f, 0.f};
      buf[fft_pad(k)] = float2{sa.x - sb.y, -(sa.y + sb.x)};
      if (k > 0 && k < NFFT / 2) buf[fft_pad(NFFT - k)] = float2{sa.x + sb.y, -(sb.x - sa.y)};
    }
  };
  build(2 * wave, true);
  // wave 0 also owns the 17th frame (local f = 16): its inputs are fetched now, before the staging area becomes the
  // overlap-add buffer
  float2 extra[9];
  if (wave == 0) {
#pragma unroll
    for (int r = 0; r < 9; ++r) { const int k = lane + 64 * r; extra[r] = k <= NFFT / 2 ? spec(GROUP, k) : float2{0.f, 0.f}; }
  }
  fft_wave<NFFT>(buf, tw, lane);
  __syncthreads();                                          // every wave is done with the staged inputs
  const int span = GROUP * p.hop;
  for (int e = tid; e < span; e += 512) ola[e] = 0.f;
  __syncthreads();
  auto add_frame = [&](int f, bool imag) {                  // rel = hop * (f - 1) + m
    const int base = p.hop * (f - 1);
#pragma unroll 4
    for (int r = 0; r < NFFT / 64; ++r) {
      const int m = lane + 64 * r, rel = base + m;
      if (rel >= 0 && rel < span) {
        const float2 z = buf[fft_pad(m)];
        ola[rel] += (imag ? -z.y : z.x) * (win[m] * (1.0f / NFFT));
      }
    }
  };
  add_frame(2 * wave, false);                               // even local frames: pairwise disjoint
  __syncthreads();
  add_frame(2 * wave + 1, true);                            // odd local frames
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int r = 0; r < 9; ++r) {
      const int k = lane + 64 * r;
      if (k > NFFT / 2) continue;
      buf[fft_pad(k)] = float2{extra[r].x, -extra[r].y};
      if (k > 0 && k < NFFT / 2) buf[fft_pad(NFFT - k)] = float2{extra[r].x, extra[r].y};
    }
    fft_wave<NFFT>(buf, tw, lane);
    add_frame(GROUP, false);                                // overlaps only frame 15, which is complete
  }
  __syncthreads();
  float vmax = 0.f;
  for (int e = tid; e < span; e += 512) {
    const long q = (long)p.hop * t0 + e;                    // padded position
    const long i = q - NFFT / 2;
    if (i < 0 || i >= p.n_out) continue;
    const float env = envelope_at(q, p.hop, p.T, win);
    const float s = ola[e];
    const float v = env > 1.1754944e-38f ? s / env : s;
    p.y[(long)c * p.n_out + i] = v;
    vmax = fmaxf(vmax, fabsf(v));
  }
  if (p.absmax_partial) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
    __syncthreads();
    if (lane == 0) ola[wave] = vmax;
    __syncthreads();
    if (tid == 0) {
      float m = ola[0];
      for (int w = 1; w < 8; ++w) m = fmaxf(m, ola[w]);
      p.absmax_partial[(long)blockIdx.y * gridDim.x + blockIdx.x] = m;
    }
  }
}

// (R, C) float2 matrix -> (C, R): f-major phasors of a .npy file <-> the frame-major form the kernels stream
__global__ __launch_bounds__(256) void transpose_c64_kernel(const float2* __restrict__ in, float2* __restrict__ out, int R, int C) {
  __shared__ float2 tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int i = threadIdx.x; i < 1024; i += 256) {
    const int r = r0 + (i >> 5), cc = c0 + (i & 31);
    if (r < R && cc < C) tile[i >> 5][i & 31] = in[(long)r * C + cc];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 256) {
    const int cc = c0 + (i >> 5), r = r0 + (i & 31);
    if (r < R && cc < C) out[(long)cc * R + r] = tile[i & 31][i >> 5];
  }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static size_t fwd_lds_bytes(bool dmag) {
  return (size_t)8 * FftSize<NFFT>::BUF * 8 + FftSize<NFFT>::TW * 8 + NFFT * 4 + NBIN * 18 * 4 + (dmag ? NBIN * SROW * 4 : 0);
}
static size_t inv_lds_bytes() {
  const size_t stage = (size_t)NBIN * IROW * 4 * 2, olab = (size_t)GROUP * NFFT * 4;
  return (size_t)8 * FftSize<NFFT>::BUF * 8 + FftSize<NFFT>::TW * 8 + NFFT * 4 + (stage > olab ? stage : olab);
}
template <class K>
static int allow_lds(K kernel, size_t bytes) {
  SVS_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return SVS_OK;
}
static int check_layout(const char* who, int seg, int rows, int first_bin, int frames_alloc, int T) {
  SVS_REQUIRE(seg > 0 && (first_bin == 0 || first_bin == 1) && rows == NBIN - first_bin && frames_alloc >= T,
              "%s: bad spectrogram layout (seg=%d rows=%d first_bin=%d frames_alloc=%d T=%d)", who, seg, rows, first_bin, frames_alloc, T);
  return SVS_OK;
}

extern "C" int svs_stft_frames(int64_t n_samples, int hop) { return hop > 0 && n_samples >= 0 ? (int)(1 + n_samples / hop) : -1; }

extern "C" int svs_stft_tiles(const float* y, int64_t n_samples, int channels, int n_fft, int hop, float* mag, int64_t chan_stride,
                              int seg, int rows, int first_bin, int frames_alloc, float* phase, int phase_mode,
                              float* absmax_partial, hipStream_t stream) {
  SVS_REQUIRE(y && mag && n_samples > 0 && channels > 0 && hop > 0, "svs_stft_tiles: bad arguments");
  SVS_REQUIRE(n_fft == NFFT, "svs_stft_tiles: only n_fft=1024 (reference config.py:47) is built, got %d", n_fft);
  SVS_REQUIRE(phase_mode >= 0 && phase_mode <= 2 && (phase_mode == 0 || (phase && (((uintptr_t)phase) & 7u) == 0)), "svs_stft_tiles: bad phase arguments");
  const int T = (int)(1 + n_samples / hop);
  int rc = check_layout("svs_stft_tiles", seg, rows, first_bin, frames_alloc, T);
  if (rc) return rc;
  StftArgs a{};
  a.y = y; a.n_samples = n_samples; a.channels = channels; a.hop = hop; a.T = T;
  a.mag = mag; a.lay = SpecLayout{chan_stride, seg, rows, first_bin, frames_alloc};
  a.phase = phase; a.phase_mode = phase_mode; a.absmax_partial = absmax_partial;
  const size_t lds = fwd_lds_bytes(false);
  if ((rc = allow_lds(stft_fwd_kernel<SRC_SIGNAL, SINK_MAGPHASE>, lds))) return rc;
  dim3 grid((unsigned)((frames_alloc + GROUP - 1) / GROUP), (unsigned)channels);
  hipLaunchKernelGGL((stft_fwd_kernel<SRC_SIGNAL, SINK_MAGPHASE>), grid, dim3(512), lds, stream, a);
  SVS_CHECK_LAUNCH("stft_fwd");
  return SVS_OK;
}
extern "C" int svs_stft_groups(int frames_alloc) { return (frames_alloc + GROUP - 1) / GROUP; }

extern "C" int svs_stft_fwd(const float* y, int64_t n_samples, int n_fft, int hop, float* mag, float* phase, hipStream_t stream) {
  SVS_REQUIRE(y && mag && n_samples > 0 && hop > 0, "svs_stft_fwd: bad arguments");
  const int T = (int)(1 + n_samples / hop);
  return svs_stft_tiles(y, n_samples, 1, n_fft, hop, mag, (int64_t)NBIN * T, T, NBIN, 0, T, phase, phase ? 2 : 0, nullptr, stream);
}

extern "C" int svs_istft_tiles(const float* mag, int64_t chan_stride, int seg, int rows, int first_bin, const float* mask, int invert,
                               const float* phase, int phase_mode, int channels, int n_fft, int hop, int frames, float* y,
                               float* absmax_partial, hipStream_t stream) {
  SVS_REQUIRE(mag && phase && y && hop > 0 && frames > 1 && channels > 0, "svs_istft_tiles: bad arguments (need >= 2 frames)");
  SVS_REQUIRE(n_fft == NFFT, "svs_istft_tiles: only n_fft=1024 (reference config.py:47) is built, got %d", n_fft);
  SVS_REQUIRE(hop <= NFFT && hop >= NFFT / 2, "svs_istft_tiles: hop %d outside [n_fft/2, n_fft] (a sample may be covered by at most two frames)", hop);
  SVS_REQUIRE(phase_mode == 1 || phase_mode == 3, "svs_istft_tiles: phase_mode must be 1 (frame-major phasors) or 3 (angles)");
  int rc = check_layout("svs_istft_tiles", seg, rows, first_bin, frames, frames);
  if (rc) return rc;
  IstftArgs a{};
  a.mag = mag; a.lay = SpecLayout{chan_stride, seg, rows, first_bin, frames};
  a.mask = mask; a.invert = invert; a.phase = phase; a.phase_mode = phase_mode;
  a.channels = channels; a.T = frames; a.hop = hop; a.y = y; a.n_out = (long)hop * (frames - 1);
  a.absmax_partial = absmax_partial;
  const size_t lds = inv_lds_bytes();
  if ((rc = allow_lds(istft_kernel, lds))) return rc;
  // padded length n_fft + hop * (T - 1): the last samples belong to group floor((padded - 1) / (16 hop))
  const long padded = NFFT + (long)hop * (frames - 1);
  dim3 grid((unsigned)((padded + (long)GROUP * hop - 1) / ((long)GROUP * hop)), (unsigned)channels);
  hipLaunchKernelGGL(istft_kernel, grid, dim3(512), lds, stream, a);
  SVS_CHECK_LAUNCH("istft");
  return SVS_OK;
}
extern "C" int svs_istft_groups(int hop, int frames) { return (int)((NFFT + (long)hop * (frames - 1) + (long)GROUP * hop - 1) / ((long)GROUP * hop)); }

extern "C" size_t svs_istft_workspace_bytes(int n_fft, int hop, int frames) {
  (void)n_fft; (void)hop;
  return (size_t)frames * NBIN * 8 + 256;           // frame-major copy of f-major phasors
}

extern "C" int svs_transpose_c64(const float* in, float* out, int rows, int cols, hipStream_t stream) {
  SVS_REQUIRE(in && out && rows > 0 && cols > 0, "svs_transpose_c64: bad arguments");
  dim3 grid((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32));
  hipLaunchKernelGGL(transpose_c64_kernel, grid, dim3(256), 0, stream, (const float2*)in, (float2*)out, rows, cols);
  SVS_CHECK_LAUNCH("transpose_c64");
  return SVS_OK;
}

extern "C" int svs_istft(const float* mag, const float* phase, int phase_is_angle, int n_fft, int hop, int frames, float* y,
                         void* ws, size_t ws_bytes, hipStream_t stream) {
  SVS_REQUIRE(mag && phase && y && hop > 0 && frames > 1, "svs_istft: bad arguments (need >= 2 frames)");
  if (phase_is_angle)
    return svs_istft_tiles(mag, (int64_t)NBIN * frames, frames, NBIN, 0, nullptr, 0, phase, 3, 1, n_fft, hop, frames, y, nullptr, stream);
  if (!ws || ws_bytes < svs_istft_workspace_bytes(n_fft, hop, frames) || !svs_aligned16(ws)) { svs_set_error("svs_istft: workspace too small"); return SVS_ERR_WORKSPACE; }
  int rc = svs_transpose_c64(phase, (float*)ws, NBIN, frames, stream);
  if (rc) return rc;
  return svs_istft_tiles(mag, (int64_t)NBIN * frames, frames, NBIN, 0, nullptr, 0, (const float*)ws, 1, 1, n_fft, hop, frames, y, nullptr, stream);
}

// Transpose of the differentiable inverse of train.py:33-60 (`specific_istft`), fused with the mask's chain rule:
//   d_logit[b, f, t] += alpha * dL/d|S|[b, f+1, t] * mix * mask * (1 - mask),   |S| = mask * mix (train.py:275,288)
// d_wav: (B, hop * (T - 1)); angle / mix / mask / d_logit: (B, 1, 512, T) training tiles.
extern "C" int svs_istft_bwd_mask(const float* d_wav, const float* angle, const float* mix, const float* mask, float* d_logit,
                                  float alpha, int B, int n_fft, int hop, int frames, hipStream_t stream) {
  SVS_REQUIRE(d_wav && angle && mix && mask && d_logit && B > 0 && frames > 1 && hop > 0, "svs_istft_bwd_mask: bad arguments");
  SVS_REQUIRE(n_fft == NFFT, "svs_istft_bwd_mask: only n_fft=1024 is built, got %d", n_fft);
  StftArgs a{};
  a.y = d_wav; a.n_samples = (long)hop * (frames - 1); a.channels = B; a.hop = hop; a.T = frames;
  a.lay = SpecLayout{(long)(NBIN - 1) * frames, frames, NBIN - 1, 1, frames};
  a.angle = angle; a.mix = mix; a.mask = mask; a.d_logit = d_logit; a.alpha = alpha;
  const size_t lds = fwd_lds_bytes(true);
  int rc = allow_lds(stft_fwd_kernel<SRC_ENVDIV, SINK_DMAG>, lds);
  if (rc) return rc;
  dim3 grid((unsigned)((frames + GROUP - 1) / GROUP), (unsigned)B);
  hipLaunchKernelGGL((stft_fwd_kernel<SRC_ENVDIV, SINK_DMAG>), grid, dim3(512), lds, stream, a);
  SVS_CHECK_LAUNCH("istft_bwd");
  return SVS_OK;
}
