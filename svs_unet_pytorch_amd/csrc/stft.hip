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
#define GROUP 16                  // forward: frames per block (8 waves x 2)
#define SROW (GROUP + 1)          // floats per staged angle row (odd -> conflict-free column access)
#define IGROUP 15                 // inverse: hops per step; they touch 16 frames = 8 waves x 2 (one of 16 transforms is halo)

// Address of (channel c, bin k, frame t) of an f-major spectrogram cut into `seg`-frame tiles of `rows` rows each,
// the first of which is bin `first_bin`:  (513, T) file: seg = T, rows = 513, first_bin = 0;
// network tiles (n, 1, 512, 128): seg = 128, rows = 512, first_bin = 1 (DC row dropped, inference.py:68 / train.py:109).
struct SpecLayout {
  long chan_stride; int seg, rows, first_bin, frames_alloc;      // frames_alloc: columns that exist (>= T: tile padding)
  __device__ __forceinline__ long at(int c, int k, int t) const {
    const int tile = t / seg;
    return (long)c * chan_stride + ((long)tile * rows + (k - first_bin)) * seg + (t - tile * seg);
  }
  // the same as col(c, t) + k * seg: one division per (thread, group) instead of one per element
  __device__ __forceinline__ long col(int c, int t) const {
    const int tile = t / seg;
    return (long)c * chan_stride + ((long)tile * rows - first_bin) * seg + (t - tile * seg);
  }
};

__device__ __forceinline__ float hann_at(int m) { return 0.5f - 0.5f * cospif((float)m * (2.0f / NFFT)); }
// the same value from the FFT's pass-2 twiddle table (tw2[256 + k] = exp(-2 pi i k / 1024), k < 256: exact sincospi
// entries), by symmetry: an LDS read and three selects instead of a ~40-instruction cospif
__device__ __forceinline__ float hann_tw(const float2* tw2, int m) {
  const int r = m <= NFFT / 2 ? m : NFFT - m;
  const float c = r < 256 ? tw2[256 + r].x : (r == 256 ? 0.f : -tw2[256 + (512 - r)].x);
  return 0.5f - 0.5f * c;
}
// one-instruction form (v_cos_f32 takes revolutions; absolute error ~1e-6): used where the window also divides out again
// (the inverse's overlap-add / envelope), not for the forward magnitudes
__device__ __forceinline__ float hann_fast(int m) { return 0.5f - 0.5f * __builtin_amdgcn_cosf((float)m * (1.0f / NFFT)); }

// window sum-of-squares at padded position q (= sample index + n_fft/2) for T frames of hop `hop`
__device__ __forceinline__ float envelope_at(int q, int hop, int T) {      // (32-bit: two 64-bit divisions per sample were most of the caller)
  if (q < 0) return 0.f;
  int t1 = (int)((unsigned)q / (unsigned)hop);
  if (t1 > T - 1) t1 = T - 1;
  int t0 = q - (NFFT - 1);
  t0 = t0 <= 0 ? 0 : (int)((unsigned)(t0 + hop - 1) / (unsigned)hop);
  float env = 0.f;
  for (int t = t0; t <= t1; ++t) { const float w = hann_fast(q - t * hop); env += w * w; }
  return env;
}

// The transposes between "a frame = a column" (what a transform works on) and "a row = consecutive frames" (what HBM
// holds) go through the waves' own FFT buffers: wave w keeps the two values (frame 2w, frame 2w+1) of bin k as ONE
// float2 at raw element k + w of its buffer (the + w skews the waves by one bank pair, so the 16 values of a row -- two
// from each of the 8 buffers, BUF * 2 floats apart = 0 mod 32 banks -- come from 16 different banks).
__device__ __forceinline__ int xpose_at(int wave, int k) { return wave * FftSize<NFFT>::BUF + k + wave; }

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
  const float2* twiddles;                                           // svs_fft_twiddles(NFFT): the device-wide table
};

template <int SRC, int SINK>
__global__ __launch_bounds__(512) void stft_fwd_kernel(StftArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int BUF = FftSize<NFFT>::BUF, TW = FftSize<NFFT>::TW;
  float2* const fbuf = (float2*)smem;                       // [8][BUF]
  float2* const tw = fbuf + 8 * BUF;                        // [TW]
  float* const angs = (float*)(tw + TW);                    // SINK_DMAG only: [513 * 17]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = blockIdx.y, t0 = blockIdx.x * GROUP;
  if (SINK == SINK_DMAG) {
    const int col = tid & (GROUP - 1), t = t0 + col;
    const long cbase = t < p.T ? p.lay.col(c, t) : 0;
    for (int k = tid >> 4; k < NBIN; k += 512 / GROUP)
      angs[k * SROW + col] = (k >= p.lay.first_bin && t < p.T) ? p.angle[cbase + (long)k * p.lay.seg] : 0.f;
  }
  float2* const buf = fbuf + wave * BUF;
  const int ta = t0 + 2 * wave, tb = ta + 1;
  // ---- frames ta, tb -> z = a + i b (windowed); 4 consecutive samples per lane and step.  The loads are issued first and
  // stay in flight while the twiddle table is built.
  const float* ysig = p.y + (long)c * p.n_samples;
  const bool vec_ok = ((p.n_samples | p.hop) & 3) == 0 && ((uintptr_t)p.y & 15u) == 0;
  float v[NFFT / 256][2][4];
#pragma unroll
  for (int r = 0; r < NFFT / 256; ++r) {
    const int m0 = 4 * lane + 256 * r;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int t = ta + h;
      const long q0 = (long)t * p.hop + m0, s0 = q0 - NFFT / 2;
      if (t < p.T && vec_ok && s0 >= 0 && s0 + 3 < p.n_samples) {
        const f32x4 x = *(const f32x4*)(ysig + s0);
        v[r][h][0] = x[0]; v[r][h][1] = x[1]; v[r][h][2] = x[2]; v[r][h][3] = x[3];
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) { const long s = s0 + j; v[r][h][j] = (t < p.T && s >= 0 && s < p.n_samples) ? ysig[s] : 0.f; }
      }
      if (SRC == SRC_ENVDIV) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float env = envelope_at((int)(q0 + j), p.hop, p.T); if (env > 1.1754944e-38f) v[r][h][j] *= __builtin_amdgcn_rcpf(env); }   // (as the inverse itself divides)
      }
    }
  }
  fft_load_twiddles<NFFT>(tw, p.twiddles, tid, 512);
  __syncthreads();                                          // twiddles (and staged angles) are complete
  const float2* const tw2 = tw + FftSize<NFFT>::TW1;
#pragma unroll
  for (int r = 0; r < NFFT / 256; ++r) {
    const int m0 = 4 * lane + 256 * r;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float w = hann_tw(tw2, m0 + j); buf[fft_pad(m0 + j)] = float2{v[r][0][j] * w, v[r][1][j] * w}; }
  }
  fft_wave<NFFT>(buf, tw, lane);
  // ---- separate the two spectra, bins k = 0 .. 512, into registers (every Z is read before the buffer is reused)
  float2 A[9], B[9];
#pragma unroll
  for (int r = 0; r < 9; ++r) {
    const int k = lane + 64 * r;
    A[r] = B[r] = float2{0.f, 0.f};
    if (k > NFFT / 2) continue;
    const float2 zk = buf[fft_pad(k)], zn = buf[fft_pad((NFFT - k) & (NFFT - 1))];
    A[r] = float2{0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y)};
    B[r] = float2{0.5f * (zk.y + zn.y), 0.5f * (zn.x - zk.x)};
  }
  fft_wave_sync();                                          // every lane has read Z before the buffer is rewritten below
  float vmax = 0.f;
#pragma unroll
  for (int r = 0; r < 9; ++r) {
    const int k = lane + 64 * r;
    if (k > NFFT / 2) continue;
    float2 out;
    if (SINK == SINK_MAGPHASE) {
      // v_sqrt_f32 / v_rcp_f32 (1 ulp) instead of the ~12-instruction IEEE sequences: the kernel is VALU-issue-bound
      const float ma = __builtin_amdgcn_sqrtf(A[r].x * A[r].x + A[r].y * A[r].y), mb = __builtin_amdgcn_sqrtf(B[r].x * B[r].x + B[r].y * B[r].y);
      out = float2{ma, mb};
      if (ta < p.T) vmax = fmaxf(vmax, ma);
      if (tb < p.T) vmax = fmaxf(vmax, mb);
      if (p.phase_mode == 1) {                             // unit phasors, 1 + 0i where the bin is exactly zero (librosa.magphase)
        float2* ph = (float2*)p.phase;
        const float ia = __builtin_amdgcn_rcpf(ma), ib = __builtin_amdgcn_rcpf(mb);
        if (ta < p.T) ph[((long)c * p.T + ta) * NBIN + k] = ma == 0.f ? float2{1.f, 0.f} : float2{A[r].x * ia, A[r].y * ia};
        if (tb < p.T) ph[((long)c * p.T + tb) * NBIN + k] = mb == 0.f ? float2{1.f, 0.f} : float2{B[r].x * ib, B[r].y * ib};
      }
    } else {
      // transpose of irfft (1/N, bins 1..511 count twice, imaginary parts of DC / Nyquist are ignored by irfft), then
      // d|S| = Re(conj(e^{i phi}) G)
      const bool edge = (k == 0 || k == NFFT / 2);
      const float ck = edge ? 1.0f / NFFT : 2.0f / NFFT;
      // (the same one-instruction phasors as the inverse it is the transpose of)
      const float ra = angs[k * SROW + 2 * wave] * 0.15915494309189535f, rb = angs[k * SROW + 2 * wave + 1] * 0.15915494309189535f;
      const float sa = __builtin_amdgcn_sinf(ra), ca = __builtin_amdgcn_cosf(ra), sb = __builtin_amdgcn_sinf(rb), cb = __builtin_amdgcn_cosf(rb);
      out = float2{ck * (A[r].x * ca + (edge ? 0.f : A[r].y * sa)), ck * (B[r].x * cb + (edge ? 0.f : B[r].y * sb))};
    }
    fbuf[xpose_at(wave, k)] = out;
  }
  __syncthreads();
  // ---- rows out: 16 consecutive frames of a bin = one 64-byte run
  const float* const xp = (const float*)fbuf;
  {
    const int col = tid & (GROUP - 1), t = t0 + col;        // a thread keeps its frame: one tile division per thread
    if (t < p.lay.frames_alloc) {
      const long cbase = p.lay.col(c, t);
      for (int k = tid >> 4; k < NBIN; k += 512 / GROUP) {
        if (k < p.lay.first_bin) continue;
        const float v = t < p.T ? xp[2 * xpose_at(col >> 1, k) + (col & 1)] : 0.f;   // tile padding beyond the last frame is written as zeros
        const long idx = cbase + (long)k * p.lay.seg;
        if (SINK == SINK_MAGPHASE) p.mag[idx] = v;
        else if (t < p.T) { const float m = p.mask[idx]; p.d_logit[idx] += p.alpha * v * p.mix[idx] * m * (1.f - m); }
      }
    }
  }
  if (SINK == SINK_MAGPHASE && p.phase_mode == 2) {
    // f-major phasors (the .npy layout of data.py:108-109): a second transposed round, phasor of frame 2w at raw
    // element k + w, of frame 2w+1 at 544 + k + w of the wave's buffer
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 9; ++r) {
      const int k = lane + 64 * r;
      if (k > NFFT / 2) continue;
      const float ma = sqrtf(A[r].x * A[r].x + A[r].y * A[r].y), mb = sqrtf(B[r].x * B[r].x + B[r].y * B[r].y);
      fbuf[xpose_at(wave, k)] = ma == 0.f ? float2{1.f, 0.f} : float2{A[r].x / ma, A[r].y / ma};
      fbuf[xpose_at(wave, k) + 544] = mb == 0.f ? float2{1.f, 0.f} : float2{B[r].x / mb, B[r].y / mb};
    }
    __syncthreads();
    float2* ph = (float2*)p.phase;
    for (int e = tid; e < NBIN * GROUP; e += 512) {
      const int k = e / GROUP, col = e - k * GROUP, t = t0 + col;
      if (t < p.T) ph[((long)c * NBIN + k) * p.T + t] = fbuf[xpose_at(col >> 1, k) + 544 * (col & 1)];
    }
  }
  if (p.absmax_partial) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
    __syncthreads();
    float* red = (float*)fbuf;
    if (lane == 0) red[wave] = vmax;
    __syncthreads();
    if (tid == 0) {
      float m = red[0];
      for (int w = 1; w < 8; ++w) m = fmaxf(m, red[w]);
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
  const float2* twiddles;                         // svs_fft_twiddles(NFFT)
};

// A 512-thread block (8 waves) owns the padded samples [hop * t0, hop * (t0 + 15)) of one channel: exactly the 16 frames
// t0-1 .. t0+14 touch them (hop >= n_fft / 2), two per wave.  Inputs are transposed into the waves' buffers (magnitudes as
// raw floats 2(k + w) + (f & 1), angles 1088 floats higher), each wave builds the Hermitian spectrum of Sa + i Sb,
// transforms, and the output threads add the (at most two) frames that cover a sample straight from the buffers and divide
// by the window envelope of the same frames.  No integer division in any per-element loop (a thread keeps its frame /
// walks its samples incrementally); the Hann window comes from the twiddle table.  79,872 B of LDS and <= 128 VGPRs: two
// blocks per CU, so one block's loads overlap the other's transform.
// (Measured alternatives, 240 s stereo: one block per CU with per-element divisions 0.29 ms; persistent blocks with register
// prefetch of the next group, 239 VGPRs, one block per CU 0.165 ms.)
template <int PMODE>                                        // phase_mode as a template argument: only its registers exist
__global__ __launch_bounds__(512, 2) void istft_kernel(IstftArgs p, int ngroups) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int BUF = FftSize<NFFT>::BUF, TW = FftSize<NFFT>::TW, NW = 8, NF = 2 * NW;
  constexpr int NIT = (NBIN * NF + 511) / 512;              // staged values per thread (17: 513 * 16 / 512)
  float2* const fbuf = (float2*)smem;                       // [8][BUF]
  float2* const tw = fbuf + NW * BUF;
  float* const raw = (float*)fbuf;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = blockIdx.x, c = g / ngroups, t0 = (g - c * ngroups) * IGROUP;
  float2* const buf = fbuf + wave * BUF;
  // 32-bit byte offsets into buffer descriptors (one VGPR per address in flight; the host checks that the views stay below
  // 2 GiB); an out-of-range element is pointed past num_records and reads as zero
  constexpr unsigned OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t rmag = __builtin_amdgcn_make_buffer_rsrc((void*)p.mag, 0, OOB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rmask = __builtin_amdgcn_make_buffer_rsrc((void*)(p.mask ? p.mask : p.mag), 0, OOB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rph = __builtin_amdgcn_make_buffer_rsrc((void*)p.phase, 0, OOB, 0x00020000);
  const bool has_mask = p.mask != nullptr;
  // ---- loads first (in flight while the twiddles are built): this thread's frame column, and this wave's phasors
  float sm[NIT], sa[PMODE == 3 ? NIT : 1];
  {
    const int tf = t0 - 1 + (tid & (NF - 1));               // a thread keeps its frame: one tile division per thread
    const bool tok = tf >= 0 && tf < p.T;
    const unsigned cbase = tok ? (unsigned)p.lay.col(c, tf) : 0u;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int k = (tid >> 4) + (512 / NF) * it;
      const bool ok = tok && k < NBIN && k >= p.lay.first_bin;
      const unsigned off = ok ? (cbase + (unsigned)(k * p.lay.seg)) * 4u : OOB;
      float m = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rmag, (int)off, 0, 0));
      if (has_mask) {
        const float mk = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rmask, (int)off, 0, 0));
        m *= p.invert ? 1.f - mk : mk;
      }
      sm[it] = m;
      if (PMODE == 3) sa[PMODE == 3 ? it : 0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rph, (int)off, 0, 0));
    }
  }
  float2 pa[PMODE == 1 ? 9 : 1], pb[PMODE == 1 ? 9 : 1];
  const int ta = t0 - 1 + 2 * wave;
  if (PMODE == 1) {
    struct F2 { float x, y; };                              // (bit_cast: whatever 8-byte type the builtin returns)
#pragma unroll
    for (int r = 0; r < 9; ++r) {
      const int k = lane + 64 * r;
      const bool oka = k <= NFFT / 2 && ta >= 0 && ta < p.T, okb = k <= NFFT / 2 && ta + 1 >= 0 && ta + 1 < p.T;
      const unsigned oa = oka ? (unsigned)(((long)c * p.T + ta) * NBIN + k) * 8u : OOB;
      const unsigned ob = okb ? (unsigned)(((long)c * p.T + ta + 1) * NBIN + k) * 8u : OOB;
      const F2 va = __builtin_bit_cast(F2, __builtin_amdgcn_raw_buffer_load_b64(rph, (int)oa, 0, 0));
      const F2 vb = __builtin_bit_cast(F2, __builtin_amdgcn_raw_buffer_load_b64(rph, (int)ob, 0, 0));
      pa[PMODE == 1 ? r : 0] = float2{va.x, va.y};
      pb[PMODE == 1 ? r : 0] = float2{vb.x, vb.y};
    }
  }
  fft_load_twiddles<NFFT>(tw, p.twiddles, tid, 512);
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int k = (tid >> 4) + (512 / NF) * it, f = tid & (NF - 1);
    if (k < NBIN) {
      const int w = f >> 1, at = 2 * (w * BUF + k + w) + (f & 1);
      raw[at] = sm[it];
      if (PMODE == 3) raw[at + BUF] = sa[PMODE == 3 ? it : 0];       // BUF floats = half a buffer higher
    }
  }
  __syncthreads();
  // spectra of this wave's two frames into registers (imaginary parts of DC / Nyquist dropped, as irfft does)
  float2 Sa[9], Sb[9];
#pragma unroll
  for (int r = 0; r < 9; ++r) {
    const int k = lane + 64 * r;
    Sa[r] = Sb[r] = float2{0.f, 0.f};
    if (k > NFFT / 2) continue;
    const float2 m = buf[k + wave];
    float2 qa = pa[PMODE == 1 ? r : 0], qb = pb[PMODE == 1 ? r : 0];
    if (PMODE == 3) {
      const float2 a = buf[k + wave + BUF / 2];
      // v_sin_f32 / v_cos_f32 (argument in revolutions; absolute error ~1e-6, the phasor multiplies a magnitude): the library
      // sincosf with its large-argument reduction was a third of this kernel's instructions
      const float ra = a.x * 0.15915494309189535f, rb = a.y * 0.15915494309189535f;
      qa = float2{__builtin_amdgcn_cosf(ra), __builtin_amdgcn_sinf(ra)};
      qb = float2{__builtin_amdgcn_cosf(rb), __builtin_amdgcn_sinf(rb)};
    }
    const bool edge = (k == 0 || k == NFFT / 2);
    Sa[r] = float2{m.x * qa.x, edge ? 0.f : m.x * qa.y};
    Sb[r] = float2{m.y * qb.x, edge ? 0.f : m.y * qb.y};
  }
  // conj(Z) with Z = Sa + i Sb (Hermitian-extended): the forward transform of conj(Z) is conj(ifft(Z)) = a - i b
  fft_wave_sync();                                          // every lane has read its staged values before the buffer is refilled
#pragma unroll
  for (int r = 0; r < 9; ++r) {
    const int k = lane + 64 * r;
    if (k > NFFT / 2) continue;
    buf[fft_pad(k)] = float2{Sa[r].x - Sb[r].y, -(Sa[r].y + Sb[r].x)};
    if (k > 0 && k < NFFT / 2) buf[fft_pad(NFFT - k)] = float2{Sa[r].x + Sb[r].y, -(Sb[r].x - Sa[r].y)};
  }
  fft_wave<NFFT>(buf, tw, lane);
  __syncthreads();
  // ---- output: sample rel = hop * (f - 1) + m of local frame f; at most frames f_hi (m < hop) and f_hi - 1 (m + hop < n_fft)
  float vmax = 0.f;
  {
    // A thread keeps its position m inside the hop and walks the 15 hops: the two window values (and, away from the ends of
    // the signal, the envelope) are computed once per thread, and which frames exist is uniform across the block per hop.
    float* const yc = p.y + (long)c * p.n_out + ((long)p.hop * t0 - NFFT / 2);
    const long ilo = NFFT / 2 - (long)p.hop * t0, ihi = p.n_out + ilo;       // valid e: ilo <= e < ihi
    auto emit = [&](int m, int fh_lo, int fh_hi) __attribute__((always_inline)) {
      const bool two = m + p.hop < NFFT;                    // frame fh - 1 still covers this position
      const float w1 = hann_fast(m) * (1.0f / NFFT), w0 = two ? hann_fast(m + p.hop) * (1.0f / NFFT) : 0.f;
      const float e1 = w1 * w1 * (float)(NFFT * NFFT), e0 = w0 * w0 * (float)(NFFT * NFFT);
      const int a1 = fft_pad(m), a0 = fft_pad(two ? m + p.hop : 0);
#pragma unroll 4
      for (int fh = fh_lo; fh <= fh_hi; ++fh) {
        const int e = (fh - 1) * p.hop + m;
        if (e < ilo || e >= ihi) continue;
        const int th = t0 - 1 + fh;                         // global index of local frame fh; fh - 1 is th - 1
        const float2 z1 = fbuf[(fh >> 1) * BUF + a1], z0 = fbuf[((fh - 1) >> 1) * BUF + a0];
        const float s = ((fh & 1) ? -z1.y : z1.x) * w1 + ((fh & 1) ? z0.x : -z0.y) * w0;     // (frames outside [0, T) hold zeros)
        const float env = ((th >= 0 && th < p.T) ? e1 : 0.f) + ((th - 1 >= 0 && th - 1 < p.T) ? e0 : 0.f);
        const float v = env > 1.1754944e-38f ? s * __builtin_amdgcn_rcpf(env) : s;
        yc[e] = v;
        vmax = fmaxf(vmax, fabsf(v));
      }
    };
    // positions 0 .. 511: one per thread, all 15 hops.  The remaining hop - 512 positions (256 at hop 768) would leave half
    // the block idle for a second full pass: instead position 512 + j is shared by threads j and j + hop - 512 (if it exists),
    // which take the lower and the upper half of the hops
    if (tid < p.hop) emit(tid, 1, IGROUP);
    const int rest = p.hop - 512;
    if (rest > 0) {
      if (2 * rest <= 512) {
        if (tid < rest) emit(512 + tid, 1, (IGROUP + 1) / 2);
        else if (tid < 2 * rest) emit(512 + tid - rest, (IGROUP + 1) / 2 + 1, IGROUP);
      } else {
        for (int m = 512 + tid; m < p.hop; m += 512) emit(m, 1, IGROUP);
      }
    }
  }
  if (p.absmax_partial) {                                   // partial[c][group]: max |y| of this block
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
    __syncthreads();
    if (lane == 0) raw[wave] = vmax;
    __syncthreads();
    if (tid == 0) {
      float m = raw[0];
      for (int w = 1; w < NW; ++w) m = fmaxf(m, raw[w]);
      p.absmax_partial[g] = m;
    }
  }
}

// General overlap-add for ANY 0 < hop <= n_fft (data.py:24-25 lets --hop_size override the config, and config.py:14-25 records
// runs at HOP_SIZE = 256: four frames per sample).  A block owns G hops = G * hop padded samples of one channel and walks the
// frames that touch them, floor((n_fft - 1) / hop) + G of them, in rounds of 16 (two per wave, the same transform as above);
// after each round every thread adds the round's frames into ITS OWN positions of an LDS accumulator (windowed sample and
// squared window side by side: fixed frame order, no atomics), and after the last round divides and stores.  Slower than the
// two-frames-per-sample kernel (one block per CU, divisions per position and round); that one keeps hop >= n_fft / 2.
template <int PMODE>
__global__ __launch_bounds__(512) void istft_general_kernel(IstftArgs p, int ngroups, int G, int rounds) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int BUF = FftSize<NFFT>::BUF, TW = FftSize<NFFT>::TW, NW = 8, NF = 2 * NW;
  constexpr int NIT = (NBIN * NF + 511) / 512;
  float2* const fbuf = (float2*)smem;                       // [8][BUF]
  float2* const tw = fbuf + NW * BUF;
  float2* const acc = tw + TW;                              // [G * hop]: (sum of windowed samples, sum of squared windows)
  float* const raw = (float*)fbuf;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = blockIdx.x, c = g / ngroups, t0 = (g - c * ngroups) * G;      // padded samples [hop * t0, hop * (t0 + G))
  const int seg_len = G * p.hop, halo = (NFFT - 1) / p.hop;
  float2* const buf = fbuf + wave * BUF;
  constexpr unsigned OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t rmag = __builtin_amdgcn_make_buffer_rsrc((void*)p.mag, 0, OOB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rmask = __builtin_amdgcn_make_buffer_rsrc((void*)(p.mask ? p.mask : p.mag), 0, OOB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rph = __builtin_amdgcn_make_buffer_rsrc((void*)p.phase, 0, OOB, 0x00020000);
  const bool has_mask = p.mask != nullptr;
  fft_load_twiddles<NFFT>(tw, p.twiddles, tid, 512);
  for (int i = tid; i < seg_len; i += 512) acc[i] = float2{0.f, 0.f};
  for (int r = 0; r < rounds; ++r) {
    const int tb = t0 - halo + NF * r;                      // first frame of this round
    __syncthreads();                                        // the previous round's buffers have been consumed (and, r = 0: twiddles)
    // ---- stage: this thread's frame column (16 frames x 513 bins through the waves' buffers, as in istft_kernel)
    {
      const int tf = tb + (tid & (NF - 1));
      const bool tok = tf >= 0 && tf < p.T;
      const unsigned cbase = tok ? (unsigned)p.lay.col(c, tf) : 0u;
      for (int it = 0; it < NIT; ++it) {
        const int k = (tid >> 4) + (512 / NF) * it;
        if (k >= NBIN) continue;
        const bool ok = tok && k >= p.lay.first_bin;
        const unsigned off = ok ? (cbase + (unsigned)(k * p.lay.seg)) * 4u : OOB;
        float m = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rmag, (int)off, 0, 0));
        if (has_mask) {
          const float mk = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rmask, (int)off, 0, 0));
          m *= p.invert ? 1.f - mk : mk;
        }
        const int f = tid & (NF - 1), w = f >> 1, at = 2 * (w * BUF + k + w) + (f & 1);
        raw[at] = m;
        if (PMODE == 3) raw[at + BUF] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rph, (int)off, 0, 0));
      }
    }
    __syncthreads();
    // ---- spectra of this wave's two frames, Hermitian-extended conj(Sa + i Sb), one transform
    const int ta = tb + 2 * wave;
    float2 Sa[9], Sb[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      const int k = lane + 64 * q;
      Sa[q] = Sb[q] = float2{0.f, 0.f};
      if (k > NFFT / 2) continue;
      const float2 m = buf[k + wave];
      float2 qa, qb;
      if (PMODE == 3) {
        const float2 a = buf[k + wave + BUF / 2];
        const float ra = a.x * 0.15915494309189535f, rb = a.y * 0.15915494309189535f;
        qa = float2{__builtin_amdgcn_cosf(ra), __builtin_amdgcn_sinf(ra)};
        qb = float2{__builtin_amdgcn_cosf(rb), __builtin_amdgcn_sinf(rb)};
      } else {
        struct F2 { float x, y; };
        const bool oka = ta >= 0 && ta < p.T, okb = ta + 1 >= 0 && ta + 1 < p.T;
        const unsigned oa = oka ? (unsigned)(((long)c * p.T + ta) * NBIN + k) * 8u : OOB;
        const unsigned ob = okb ? (unsigned)(((long)c * p.T + ta + 1) * NBIN + k) * 8u : OOB;
        const F2 va = __builtin_bit_cast(F2, __builtin_amdgcn_raw_buffer_load_b64(rph, (int)oa, 0, 0));
        const F2 vb = __builtin_bit_cast(F2, __builtin_amdgcn_raw_buffer_load_b64(rph, (int)ob, 0, 0));
        qa = float2{va.x, va.y};
        qb = float2{vb.x, vb.y};
      }
      const bool edge = (k == 0 || k == NFFT / 2);
      Sa[q] = float2{m.x * qa.x, edge ? 0.f : m.x * qa.y};
      Sb[q] = float2{m.y * qb.x, edge ? 0.f : m.y * qb.y};
    }
    fft_wave_sync();                                        // every lane has read its staged values before the buffer is refilled
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      const int k = lane + 64 * q;
      if (k > NFFT / 2) continue;
      buf[fft_pad(k)] = float2{Sa[q].x - Sb[q].y, -(Sa[q].y + Sb[q].x)};
      if (k > 0 && k < NFFT / 2) buf[fft_pad(NFFT - k)] = float2{Sa[q].x + Sb[q].y, -(Sb[q].x - Sa[q].y)};
    }
    fft_wave<NFFT>(buf, tw, lane);
    __syncthreads();
    // ---- accumulate: position i (padded sample hop * t0 + i) takes every frame of this round that covers it, ascending
    for (int i = tid; i < seg_len; i += 512) {
      const int q = p.hop * t0 + i;
      int lo = q - (NFFT - 1);
      lo = lo <= 0 ? 0 : (int)((unsigned)(lo + p.hop - 1) / (unsigned)p.hop);
      int hi = (int)((unsigned)q / (unsigned)p.hop);
      if (hi > p.T - 1) hi = p.T - 1;
      if (lo < tb) lo = tb;
      if (hi > tb + NF - 1) hi = tb + NF - 1;
      float2 a = acc[i];
      for (int t = lo; t <= hi; ++t) {
        const int f = t - tb, m = q - t * p.hop;
        const float2 z = fbuf[(f >> 1) * BUF + fft_pad(m)];
        const float w = hann_fast(m);
        a.x += ((f & 1) ? -z.y : z.x) * (w * (1.0f / NFFT));
        a.y += w * w;
      }
      acc[i] = a;
    }
  }
  // ---- divide by the window envelope, store, |.|max
  float vmax = 0.f;
  for (int i = tid; i < seg_len; i += 512) {
    const long e = (long)p.hop * t0 + i - NFFT / 2;
    if (e < 0 || e >= p.n_out) continue;
    const float2 a = acc[i];
    const float v = a.y > 1.1754944e-38f ? a.x * __builtin_amdgcn_rcpf(a.y) : a.x;
    p.y[(long)c * p.n_out + e] = v;
    vmax = fmaxf(vmax, fabsf(v));
  }
  if (p.absmax_partial) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
    __syncthreads();
    if (lane == 0) raw[wave] = vmax;
    __syncthreads();
    if (tid == 0) {
      float m = raw[0];
      for (int w = 1; w < NW; ++w) m = fmaxf(m, raw[w]);
      p.absmax_partial[g] = m;
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
static size_t fwd_lds_bytes(bool dmag) {       // 79,872 B: two blocks per CU
  return (size_t)8 * FftSize<NFFT>::BUF * 8 + FftSize<NFFT>::TW * 8 + (dmag ? NBIN * SROW * 4 : 0);
}
static size_t inv_lds_bytes() { return (size_t)8 * FftSize<NFFT>::BUF * 8 + FftSize<NFFT>::TW * 8; }    // 79,872 B: two blocks per CU
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
  if ((rc = svs_fft_twiddles(NFFT, stream, &a.twiddles))) return rc;
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

// hops per block and rounds of 16 frames of the general kernel: G + floor((n_fft - 1) / hop) frames touch G hops; one round
// while the halo leaves at least 4 hops of it (hop >= 86), else enough rounds that at least half of every round is new hops
static void istft_general_plan(int hop, int* G, int* rounds) {
  const int halo = (NFFT - 1) / hop;
  *rounds = halo <= 12 ? 1 : (halo + 7) / 8;
  *G = 16 * *rounds - halo;
}
// padded length n_fft + hop * (T - 1): the last samples belong to group floor((padded - 1) / (G hop)), G = 15 hops per block
// of the two-frames-per-sample kernel (hop >= n_fft / 2), istft_general_plan's below it
static int svs_istft_groups_per_channel(int hop, int frames) {
  int G = IGROUP, rounds;
  if (hop < NFFT / 2) istft_general_plan(hop, &G, &rounds);
  return (int)((NFFT + (long)hop * (frames - 1) + (long)G * hop - 1) / ((long)G * hop));
}
// blocks per channel of svs_istft_tiles = absmax partials per channel (layout [channel][group])
extern "C" int svs_istft_groups(int hop, int frames, int channels) { (void)channels; return svs_istft_groups_per_channel(hop, frames); }

extern "C" int svs_istft_tiles(const float* mag, int64_t chan_stride, int seg, int rows, int first_bin, const float* mask, int invert,
                               const float* phase, int phase_mode, int channels, int n_fft, int hop, int frames, float* y,
                               float* absmax_partial, hipStream_t stream) {
  SVS_REQUIRE(mag && phase && y && hop > 0 && frames > 1 && channels > 0, "svs_istft_tiles: bad arguments (need >= 2 frames)");
  SVS_REQUIRE(n_fft == NFFT, "svs_istft_tiles: only n_fft=1024 (reference config.py:47) is built, got %d", n_fft);
  SVS_REQUIRE(hop <= NFFT, "svs_istft_tiles: hop %d > n_fft leaves samples that no frame covers", hop);
  SVS_REQUIRE(phase_mode == 1 || phase_mode == 3, "svs_istft_tiles: phase_mode must be 1 (frame-major phasors) or 3 (angles)");
  SVS_REQUIRE((long)channels * (chan_stride > (long)frames * NBIN ? chan_stride : (long)frames * NBIN) * 8 < (1L << 31),
              "svs_istft_tiles: a spectrogram view of more than 2 GiB needs 64-bit offsets; split the channels");
  int rc = check_layout("svs_istft_tiles", seg, rows, first_bin, frames, frames);
  if (rc) return rc;
  IstftArgs a{};
  a.mag = mag; a.lay = SpecLayout{chan_stride, seg, rows, first_bin, frames};
  a.mask = mask; a.invert = invert; a.phase = phase; a.phase_mode = phase_mode;
  a.channels = channels; a.T = frames; a.hop = hop; a.y = y; a.n_out = (long)hop * (frames - 1);
  a.absmax_partial = absmax_partial;
  if ((rc = svs_fft_twiddles(NFFT, stream, &a.twiddles))) return rc;
  const int ngroups = svs_istft_groups_per_channel(hop, frames);
  const long total = (long)ngroups * channels;
  const dim3 grid((unsigned)total);
  if (hop < NFFT / 2) {                          // more than two frames per sample: the general overlap-add
    int G, rounds;
    istft_general_plan(hop, &G, &rounds);
    const size_t lds = inv_lds_bytes() + (size_t)G * hop * 8;
    if ((rc = phase_mode == 1 ? allow_lds(istft_general_kernel<1>, lds) : allow_lds(istft_general_kernel<3>, lds))) return rc;
    if (phase_mode == 1) hipLaunchKernelGGL(istft_general_kernel<1>, grid, dim3(512), lds, stream, a, ngroups, G, rounds);
    else hipLaunchKernelGGL(istft_general_kernel<3>, grid, dim3(512), lds, stream, a, ngroups, G, rounds);
    SVS_CHECK_LAUNCH("istft_general");
    return SVS_OK;
  }
  const size_t lds = inv_lds_bytes();
  if ((rc = phase_mode == 1 ? allow_lds(istft_kernel<1>, lds) : allow_lds(istft_kernel<3>, lds))) return rc;
  if (phase_mode == 1) hipLaunchKernelGGL(istft_kernel<1>, grid, dim3(512), lds, stream, a, ngroups);
  else hipLaunchKernelGGL(istft_kernel<3>, grid, dim3(512), lds, stream, a, ngroups);
  SVS_CHECK_LAUNCH("istft");
  return SVS_OK;
}
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
  if ((rc = svs_fft_twiddles(NFFT, stream, &a.twiddles))) return rc;
  dim3 grid((unsigned)((frames + GROUP - 1) / GROUP), (unsigned)B);
  hipLaunchKernelGGL((stft_fwd_kernel<SRC_ENVDIV, SINK_DMAG>), grid, dim3(512), lds, stream, a);
  SVS_CHECK_LAUNCH("istft_bwd");
  return SVS_OK;
}
