// One complex FFT per WAVE (64 lanes), data in LDS, gfx950.  N = 512 / 1024 / 2048.
//
// Stockham autosort in three register-radix passes (512 = 8*8*8, 1024 = 16*16*4, 2048 = 16*16*8): a lane reads the
// R inputs of each of its N/(64 R) butterflies into registers (conflict-free ds_read_b64: consecutive lanes read
// consecutive points), multiplies by the pass twiddles, does the radix-R DFT in registers and writes the R outputs
// back.  A pass is "all reads, then all writes" of ONE wave, so it runs IN PLACE in a single buffer with no barrier:
// LDS operations of a wave execute in program order, and an instruction covers all 64 lanes at once.  Two LDS round
// trips per transform (the five-pass radix-4 form it replaces made five block-wide ones with barriers).
//
//   * buffer index: logical element i lives at i + (i >> 4) (one pad element per 16): the strided writes of a pass
//     (stride R or R*Ns elements) then hit 16 different bank pairs per 16-lane group;
//   * twiddles: one small table per pass, [t][k] with k fastest (tw[t * Ns + k] = exp(-2 pi i k t / (Ns R))), so the
//     lanes of a pass read consecutive entries (conflict-free) and every entry is an exact sincospi value;
//   * direction: forward e^{-i}; the inverse is conj(FFT(conj(x))) (unnormalised) -- the callers conjugate on
//     load / store, which is free.
//
// The header also compiles for the host (tools/fft_host_check.cpp emulates a wave lane by lane) so that the index
// arithmetic is checked against a direct DFT without a GPU.
#pragma once

#ifndef FFT_HD
#define FFT_HD __host__ __device__ __forceinline__
#endif

template <int N> struct FftPlan;
template <> struct FftPlan<512> { static constexpr int R0 = 8, R1 = 8, R2 = 8; };
template <> struct FftPlan<1024> { static constexpr int R0 = 16, R1 = 16, R2 = 4; };
template <> struct FftPlan<2048> { static constexpr int R0 = 16, R1 = 16, R2 = 8; };

FFT_HD int fft_pad(int i) { return i + (i >> 4); }
template <int N> struct FftSize {
  static constexpr int BUF = N + N / 16;                                     // float2 elements of one wave's buffer
  static constexpr int NS1 = FftPlan<N>::R0;                                 // sub-transform length before pass 1 / 2
  static constexpr int NS2 = FftPlan<N>::R0 * FftPlan<N>::R1;
  static constexpr int TW1 = NS1 * FftPlan<N>::R1;                           // entries of the pass-1 table
  static constexpr int TW2 = NS2 * FftPlan<N>::R2;                           // entries of the pass-2 table (= N)
  static constexpr int TW = TW1 + TW2;
};

FFT_HD float2 fft_cmul(float2 a, float2 b) { return float2{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }

// entry e (0 .. FftSize<N>::TW - 1) of the twiddle tables, as an angle in units of pi: tw = (cos(pi a), sin(pi a))
template <int N>
FFT_HD float fft_twiddle_angle(int e) {
  using S = FftSize<N>;
  if (e < S::TW1) {
    const int t = e / S::NS1, k = e - t * S::NS1;
    return -2.0f * (float)(k * t) / (float)(S::NS1 * FftPlan<N>::R1);
  }
  e -= S::TW1;
  const int t = e / S::NS2, k = e - t * S::NS2;
  return -2.0f * (float)(k * t) / (float)(S::NS2 * FftPlan<N>::R2);           // k*t < N*R2 <= 2^14: exact in float
}

// radix-R DFT of v in place (decimation in frequency): afterwards X[t] = v[bitrev_R(t)]
template <int R>
FFT_HD void fft_dft(float2 (&v)[R]) {
  constexpr float C16[8] = {1.0f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f,
                            0.0f, -0.38268343236508977f, -0.70710678118654752f, -0.92387953251128674f};
  constexpr float S16[8] = {0.0f, 0.38268343236508977f, 0.70710678118654752f, 0.92387953251128674f,
                            1.0f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f};
#pragma unroll
  for (int half = R / 2; half >= 1; half >>= 1) {
#pragma unroll
    for (int base = 0; base < R; base += 2 * half) {
#pragma unroll
      for (int i = 0; i < half; ++i) {
        const float2 a = v[base + i], b = v[base + i + half];
        v[base + i] = float2{a.x + b.x, a.y + b.y};
        const float2 d = float2{a.x - b.x, a.y - b.y};
        const int k = i * (8 / half);                    // twiddle exp(-2 pi i * i / (2 half)) = exp(-2 pi i k / 16)
        if (k == 0) v[base + i + half] = d;
        else if (k == 4) v[base + i + half] = float2{d.y, -d.x};
        else v[base + i + half] = float2{d.x * C16[k] + d.y * S16[k], d.y * C16[k] - d.x * S16[k]};
      }
    }
  }
}
template <int R> FFT_HD constexpr int fft_bitrev(int t) {
  int r = 0;
  for (int b = 1; b < R; b <<= 1) { r = (r << 1) | (t & 1); t >>= 1; }
  return r;
}

// One pass of one lane, split so that a host emulation can run "all loads of all lanes, then all stores".
template <int N, int R, int NS>
struct FftPass {
  static constexpr int NB = N / R / 64;                  // butterflies per lane
  float2 v[NB][R];
  // Padded indices as (one lane-dependent base) + (compile-time constant), so that every access of a pass is one base register
  // and an immediate offset: pad(a + c) = pad(a) + pad(c) whenever c is a multiple of 16 (the shift cannot carry), which the
  // compiler does not derive by itself from i + (i >> 4) -- it built a separate address for most of the 16 accesses.
  static_assert((N / R) % 64 == 0, "pass stride");
  FFT_HD void load(const float2* buf, int lane) {
    const float2* base = buf + fft_pad(lane);
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int t = 0; t < R; ++t) v[b][t] = base[fft_pad(64 * b + t * (N / R))];
  }
  FFT_HD void compute(const float2* tw, int lane) {      // tw: this pass's table ([t][k]); unused when NS == 1
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (NS > 1) {
        const int k = (lane + 64 * b) & (NS - 1);
#pragma unroll
        for (int t = 1; t < R; ++t) v[b][t] = fft_cmul(v[b][t], tw[t * NS + k]);
      }
      fft_dft<R>(v[b]);
    }
  }
  FFT_HD void store(float2* buf, int lane) {
    if constexpr (NS == 1 && (R == 16 || R == 8)) {
      // element (lane + 64 b) R + t: the low four bits of lane R are 0 (R = 16) or 0 / 8 (R = 8), so adding t < R never carries
      // into the padding shift: pad = pad(lane R) + t + pad(64 b R)
      float2* base = buf + fft_pad(lane * R);
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int t = 0; t < R; ++t) base[t + fft_pad(64 * b * R)] = v[b][fft_bitrev<R>(t)];
    } else if constexpr (NS % 16 == 0 && NS <= 64) {
      // k = lane & (NS - 1) is the same for every b; element (lane - k) R + k + (64 b R + t NS), the bracket a multiple of 16
      const int k = lane & (NS - 1);
      float2* base = buf + fft_pad((lane - k) * R + k);
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int t = 0; t < R; ++t) base[fft_pad(64 * b * R + t * NS)] = v[b][fft_bitrev<R>(t)];
    } else if constexpr (NS % 64 == 0 && NS >= 64 * NB) {
      // lane + 64 b < NS: k = j, element lane + (64 b + t NS)
      float2* base = buf + fft_pad(lane);
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int t = 0; t < R; ++t) base[fft_pad(64 * b + t * NS)] = v[b][fft_bitrev<R>(t)];
    } else {
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int j = lane + 64 * b, k = j & (NS - 1);
        const int j0 = (j - k) * R + k;
#pragma unroll
        for (int t = 0; t < R; ++t) buf[fft_pad(j0 + t * NS)] = v[b][fft_bitrev<R>(t)];
      }
    }
  }
};

#if defined(__HIPCC__)
// Forward DFT of the N points in `buf` (logical order, padded index), in place, by the calling wave.
// tw: FftSize<N>::TW entries built with fft_build_twiddles.  The caller orders its own LDS accesses around the call
// (same wave: program order suffices; other waves must not touch this buffer).
// Between a pass's stores and the next pass's loads (lanes read what OTHER lanes wrote) the hardware needs nothing -- the LDS
// operations of one wave execute in program order -- but the compiler must not move a load above a store it can prove
// independent for the single thread: a wave-level scheduling barrier (no instruction) pins the order.
//
// NOTE for every file that includes this header: build it with -fno-slp-vectorize (svs_unet_pytorch_amd/build.py).  The SLP
// vectoriser turns complex multiplies into packed-fp32 instructions with op_sel swizzles, and on gfx950 a v_pk_{add,mul,fma}_f32
// whose op_sel takes the HIGH half of a source for the low result returns garbage while a bf16 MFMA of any other wave is
// executing on the CU (whole frames of garbage as soon as another stream or process ran the bf16 network; tools/attic/stress_victims.py,
// tools/check_isa.py).  Without the packed forms the transforms are also 5-10 % faster.
__device__ __forceinline__ void fft_wave_sync() { __builtin_amdgcn_wave_barrier(); }
template <int N>
__device__ __forceinline__ void fft_wave(float2* buf, const float2* tw, int lane) {
  using P = FftPlan<N>;
  using S = FftSize<N>;
  fft_wave_sync();                           // the caller's fill
  { FftPass<N, P::R0, 1> p; p.load(buf, lane); p.compute(nullptr, lane); p.store(buf, lane); }
  fft_wave_sync();
  { FftPass<N, P::R1, S::NS1> p; p.load(buf, lane); p.compute(tw, lane); p.store(buf, lane); }
  fft_wave_sync();
  { FftPass<N, P::R2, S::NS2> p; p.load(buf, lane); p.compute(tw + S::TW1, lane); p.store(buf, lane); }
  fft_wave_sync();                           // before the caller reads other lanes' outputs
}
template <int N>
__device__ __forceinline__ void fft_build_twiddles(float2* tw, int tid, int nthreads) {
  for (int e = tid; e < FftSize<N>::TW; e += nthreads) {
    float s, c;
    sincospif(fft_twiddle_angle<N>(e), &s, &c);
    tw[e] = float2{c, s};
  }
}
// the block's copy of the device-wide table (fft_tables.hip: the same values, built once): two entries per 16-byte load
template <int N>
__device__ __forceinline__ void fft_load_twiddles(float2* tw, const float2* __restrict__ table, int tid, int nthreads) {
  static_assert(FftSize<N>::TW % 2 == 0, "table length");
  typedef float fft_f4 __attribute__((ext_vector_type(4)));
  for (int e = tid; e < FftSize<N>::TW / 2; e += nthreads) ((fft_f4*)tw)[e] = ((const fft_f4*)table)[e];
}
#endif
