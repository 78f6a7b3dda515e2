// The single-channel ends of the U-Net (bandwidth-bound, no MFMA), gfx950:
//   conv_c1_kernel    1 -> COUT channels, 5x5 stride-2 gather.  conv1 forward (reference model.py:47-51)
//                     and deconv6 backward-data (model.py:109).
//   deconv_to1_kernel CIN -> 1 channel transposed conv + sigmoid.  deconv6 forward (model.py:198-200).
//   wgrad_c1_kernel   dw[cs][tap] = sum_pix S[pix][cs] * L[window(pix, tap)], L single-channel.
//                     conv1 / deconv6 weight gradients.
// All three give each pixel's channel vector to COUT/4 (CIN/4) adjacent lanes as float4, so the wide
// side is read/written in contiguous 16-byte pieces; the single-channel image is read through L1.
// Bound: HBM.  Algorithmic bytes: the NHWC tensor once + the 1-channel image once.
#include "internal.h"

struct C1Args {
  const float* x; int B, H, W;            // (B,H,W) single channel
  const float* w;                         // [COUT][25]  (torch (COUT,1,5,5) == gather packing with C=1)
  const float* bias; const float* scale; const float* shift; float slope;
  float* y; long ldy; int Ho, Wo; int accumulate;
  long half;                              // floats between channel COUT/2-1 and COUT/2 of a pixel minus ... see chan_off()
};

// Address of channel group `sub` (4 channels) of pixel `pix` in a view whose two channel halves may live in
// different places ("planar" level-1 buffers: the decoder half and the skip half of the 32-channel level-1
// tensor are two dense 16-channel planes, so that each half is read and written in full 64-byte pieces by
// its 16-channel producers/consumers).  half == 0: ordinary interleaved view.
__device__ __forceinline__ long chan_off(long pix, long ld, int sub, int subs_per_half, long half) {
  return pix * ld + (half ? (sub >= subs_per_half ? half + (sub - subs_per_half) * 4 : sub * 4) : sub * 4);
}

// One thread = one output pixel, all COUT channels: the 25 input samples are loaded once per pixel (the
// earlier 4-channels-per-thread mapping re-issued them COUT/4 times and was bound by load issue), the
// weights are wave-uniform and come through the scalar cache, and the COUT results go through an LDS
// transpose so that the global stores are contiguous 16-byte pieces of each pixel's channel vector.
template <int COUT>
__global__ __launch_bounds__(256) void conv_c1_kernel(C1Args p) {
  constexpr int G = COUT / 4;
  constexpr int LD = COUT + 4;
  __shared__ __attribute__((aligned(16))) float tile[256 * LD];
  const float* __restrict__ w = p.w;
  const long P = (long)p.B * p.Ho * p.Wo;
  const long pix0 = (long)blockIdx.x * 256;
  const long pix = pix0 + threadIdx.x;
  if (pix < P) {
    const unsigned upix = (unsigned)pix, utmp = upix / (unsigned)p.Wo;          // (P < 2^31 checked on the host: 32-bit divisions)
    const int ow = (int)(upix - utmp * (unsigned)p.Wo);
    const unsigned ub = utmp / (unsigned)p.Ho;
    const int oh = (int)(utmp - ub * (unsigned)p.Ho);
    // the 25 samples as buffer loads: taps outside the image get an out-of-range offset and read as zero, so there is
    // no branch around any load (25 exec-masked regions before) and all of them are in flight together
    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, OOB, 0x00020000);
    const int img = (int)ub * p.H * p.W;                                        // (whole input < 2 GiB checked on the host)
    float xin[25];
#pragma unroll
    for (int kh = 0; kh < 5; ++kh) {
      const int ih = 2 * oh - 2 + kh;
      const bool okh = (unsigned)ih < (unsigned)p.H;
#pragma unroll
      for (int kw = 0; kw < 5; ++kw) {
        const int iw = 2 * ow - 2 + kw;
        const unsigned vo = (okh && (unsigned)iw < (unsigned)p.W) ? (unsigned)(img + ih * p.W + iw) * 4u : OOB;
        xin[kh * 5 + kw] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, (int)vo, 0, 0));
      }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      f32x4 acc;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int n = g * 4 + k;
        float a = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int tap = 0; tap < 25; ++tap) a += xin[tap] * w[n * 25 + tap];
        if (p.scale) {
          a = a * p.scale[n] + p.shift[n];
          a = a > 0.f ? a : a * p.slope;
        }
        acc[k] = a;
      }
      *(f32x4*)(&tile[threadIdx.x * LD + g * 4]) = acc;
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < G; ++r) {
    const int id = threadIdx.x + 256 * r;
    const int pp = id / G, sub = id - pp * G;
    if (pix0 + pp < P) {
      f32x4 v = *(const f32x4*)(&tile[pp * LD + sub * 4]);
      float* dst = p.y + chan_off(pix0 + pp, p.ldy, sub, G / 2, p.half);
      if (p.accumulate) v += *(const f32x4*)dst;
      *(f32x4*)dst = v;
    }
  }
}

// Tiled form of the same convolution.  The thread-per-pixel kernel above multiplies 25 x COUT values per thread behind scalar
// weight loads (COUT x 25 weights do not fit the scalar registers: 25-50 dependent s_load rounds per wave) with one wave's worth
// of work in flight per pixel row: 2.5-2.9 TB/s.  Here a persistent block owns runs of 4 x 64-output tiles: the (11 x 131)-sample
// window of a tile is staged ONCE in LDS (requested one tile ahead), a thread is (pixel, group of 4 channels) with its 25 x 4
// weights in registers for the whole launch, reads each window row of its pixel as three 8-byte LDS loads (the 4 or 8 lanes of
// a pixel read the same address: broadcast; consecutive pixels are 2 floats apart: conflict-free) and stores its float4
// directly -- the COUT/4 lanes of a pixel write one contiguous 64 / 128-byte run, so there is no LDS transpose of the output.
template <int COUT>
__global__ __launch_bounds__(256, COUT == 32 ? 3 : 2) void conv_c1_tiled_kernel(C1Args p, int tiles_h, int tiles_w) {   // 166 / 188 registers: three / two blocks per CU (the 16-channel form spills at three)
  constexpr int G = COUT / 4, TOH = 4, TOW = 64;
  constexpr int WH = 2 * TOH + 3, WWD = 2 * TOW + 4;          // 11 x 132 floats (131 used; even pitch keeps the 8-byte reads aligned)
  constexpr int NW = WH * WWD, NST = (NW + 255) / 256;
  constexpr int PASSES = TOH * TOW * G / 256;
  constexpr unsigned OOB = 0x80000000u;
  __shared__ __attribute__((aligned(16))) float win[NW];
  const int t = threadIdx.x, cg = t % G, pl = t / G;
  f32x4 wr[25];
#pragma unroll
  for (int tap = 0; tap < 25; ++tap)
    wr[tap] = (f32x4){p.w[(cg * 4 + 0) * 25 + tap], p.w[(cg * 4 + 1) * 25 + tap], p.w[(cg * 4 + 2) * 25 + tap], p.w[(cg * 4 + 3) * 25 + tap]};
  f32x4 b4 = {0.f, 0.f, 0.f, 0.f}, sc4 = {1.f, 1.f, 1.f, 1.f}, sh4 = {0.f, 0.f, 0.f, 0.f};
  if (p.bias) b4 = *(const f32x4*)(p.bias + cg * 4);
  if (p.scale) { sc4 = *(const f32x4*)(p.scale + cg * 4); sh4 = *(const f32x4*)(p.shift + cg * 4); }
  const int ntiles = p.B * tiles_h * tiles_w;
  const int tile_lo = (int)((long)ntiles * blockIdx.x / gridDim.x), tile_hi = (int)((long)ntiles * (blockIdx.x + 1) / gridDim.x);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, OOB, 0x00020000);
  float stage[NST];
  auto fetch = [&](int tile) {            // the tile's input window, zero outside the image (out-of-range offset: no branches)
    const int ow0 = (tile % tiles_w) * TOW, oh0 = ((tile / tiles_w) % tiles_h) * TOH;
    const int img = (tile / (tiles_w * tiles_h)) * p.H * p.W;                    // (whole input < 2 GiB: checked on the host)
#pragma unroll
    for (int k = 0; k < NST; ++k) {
      const int e = t + 256 * k, lh = e / WWD, lw = e - lh * WWD;
      const int ih = 2 * oh0 - 2 + lh, iw = 2 * ow0 - 2 + lw;
      const bool ok = e < NW && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
      stage[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, ok ? (int)((unsigned)(img + ih * p.W + iw) * 4u) : (int)OOB, 0, 0));
    }
  };
  if (tile_lo < tile_hi) fetch(tile_lo);
  for (int tile = tile_lo; tile < tile_hi; ++tile) {
    const int ow0 = (tile % tiles_w) * TOW, oh0 = ((tile / tiles_w) % tiles_h) * TOH;
    const long b = tile / (tiles_w * tiles_h);
    __syncthreads();                                            // the previous tile's readers are done
#pragma unroll
    for (int k = 0; k < NST; ++k) { const int e = t + 256 * k; if (e < NW) win[e] = stage[k]; }
    __syncthreads();
    if (tile + 1 < tile_hi) fetch(tile + 1);                    // in flight while this tile is computed
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      const int px = ps * (256 / G) + pl, lr = px / TOW, lc = px - lr * TOW;
      f32x4 acc = b4;
#pragma unroll
      for (int kh = 0; kh < 5; ++kh) {
        const float2* row = (const float2*)(&win[(2 * lr + kh) * WWD + 2 * lc]);
        const float2 a = row[0], c = row[1], e = row[2];
        // the odd samples sit in the HIGH half of their loaded register pair; multiplied as they stand, the packed FMA takes
        // them with op_sel:[1,..] -- the form that returns garbage beside a bf16 MFMA on gfx950 (tools/check_isa.py, DESIGN
        // section 5).  A copy into a register of its own makes it a plain low-half broadcast.
        float ay = a.y, cy = c.y;
        asm volatile("v_mov_b32 %0, %1" : "=v"(ay) : "v"(a.y));
        asm volatile("v_mov_b32 %0, %1" : "=v"(cy) : "v"(c.y));
        acc += wr[kh * 5 + 0] * a.x; acc += wr[kh * 5 + 1] * ay; acc += wr[kh * 5 + 2] * c.x; acc += wr[kh * 5 + 3] * cy;
        acc += wr[kh * 5 + 4] * e.x;
      }
      if (p.scale) {
        acc = acc * sc4 + sh4;
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = acc[k] > 0.f ? acc[k] : acc[k] * p.slope;
      }
      const int oh = oh0 + lr, ow = ow0 + lc;
      if (oh < p.Ho && ow < p.Wo) {
        float* dst = p.y + chan_off((b * p.Ho + oh) * p.Wo + ow, p.ldy, cg, G / 2, p.half);
        if (p.accumulate) acc += *(const f32x4*)dst;
        *(f32x4*)dst = acc;
      }
    }
  }
}

int svs_conv_c1_run(const float* x, int B, int H, int W, const float* w, const float* bias, const float* scale,
                    const float* shift, float slope, float* y, long ldy, int N, int accumulate, hipStream_t stream,
                    const char* who, long half) {
  SVS_REQUIRE(x && w && y, "%s: null pointer", who);
  SVS_REQUIRE(N == 16 || N == 32, "%s: single-channel conv supports N=16/32, got %d", who, N);
  SVS_REQUIRE(ldy >= (half ? N / 2 : N) && ldy % 4 == 0 && svs_aligned16(y), "%s: bad output view", who);
  C1Args a{x, B, H, W, w, bias, scale, shift, slope, y, ldy, svs_conv_out(H), svs_conv_out(W), accumulate, half};
  const long total = (long)B * a.Ho * a.Wo;
  SVS_REQUIRE(total < (1L << 31), "%s: %ld output pixels need 64-bit indices; split the batch", who, total);
  SVS_REQUIRE((long)B * H * W * 4 < (1L << 31), "%s: the input must span < 2 GiB (32-bit buffer offsets); split the batch", who);
  // same-device A/B at batch 64 (tools/ab_c1_tiled.py): 32 channels (deconv6 backward-data) 50.0 -> 44.4 us, 16 channels
  // (conv1 forward) 25.9 -> 29.9 us -- with 36 registers the thread-per-pixel form runs 8 waves per SIMD there
  const long tiled = svs_tune(SVS_TUNE_CONV_C1_TILED);       // 0: never, 2: always (A/B runs, tests)
  if (tiled == 2 || (tiled != 0 && N == 32)) {
    const int tiles_h = (a.Ho + 3) / 4, tiles_w = (a.Wo + 63) / 64;
    const long ntiles = (long)B * tiles_h * tiles_w;
    SVS_REQUIRE(ntiles < (1L << 31), "%s: too many tiles", who);
    const long cap = N == 32 ? 768 : 512;                   // persistent: every block resident (three / two per CU), weights loaded once per block
    const int grid = (int)(ntiles < cap ? ntiles : cap);
    if (N == 16) hipLaunchKernelGGL(conv_c1_tiled_kernel<16>, dim3(grid), dim3(256), 0, stream, a, tiles_h, tiles_w);
    else hipLaunchKernelGGL(conv_c1_tiled_kernel<32>, dim3(grid), dim3(256), 0, stream, a, tiles_h, tiles_w);
    SVS_CHECK_LAUNCH("conv_c1_tiled");
    return SVS_OK;
  }
  const int grid = (int)((total + 255) / 256);
  if (N == 16) hipLaunchKernelGGL(conv_c1_kernel<16>, dim3(grid), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(conv_c1_kernel<32>, dim3(grid), dim3(256), 0, stream, a);
  SVS_CHECK_LAUNCH("conv_c1");
  return SVS_OK;
}

// -------------------------------------------------------------------------------------------------
struct To1Args {
  const float* x; long ldx; int B, H, W;   // (B,H,W,CIN)
  const float* w;                          // torch (CIN,1,5,5) = [c][25]
  const float* bias;                       // device scalar or null
  float* y; int Ho, Wo; int sigmoid;
  long half;                               // see chan_off()
};

typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {           // cross-lane move on the VALU (no LDS round trip)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// One block = a tile of TH x TW input pixels (= 2TH x 2TW outputs) of one image.  The tile plus a one-pixel halo
// is staged ONCE in LDS (zero-filled outside the image), so every input vector is fetched from HBM once instead
// of by each of the 9 output quads that use it (PMC: 431 MB per launch before, ~150 MB algorithmic); each lane
// keeps the 25 x 4 weights of its channel group in registers and combines 9 LDS vectors into its quad's 4
// partial outputs, which the CIN/4 lanes of a quad then sum with shuffles.
template <int CIN>
__global__ __launch_bounds__(256) void deconv_to1_kernel(To1Args p) {
  constexpr int G = CIN / 4;                 // lanes per quad
  constexpr int TH = 8, TW = 32;             // quads per tile
  constexpr int LP = CIN + 4;                // floats per staged pixel (pad keeps 16-byte alignment, breaks bank stride)
  __shared__ __attribute__((aligned(16))) float tile_lds[(TH + 2) * (TW + 2) * LP];
  const int t = threadIdx.x;
  const int tiles_w = (p.W + TW - 1) / TW, tiles_h = (p.H + TH - 1) / TH;
  const int ntiles = p.B * tiles_h * tiles_w;
  // this lane's weights: w[c][tap] for c = 4*cg .. 4*cg+3  (torch (CIN,1,5,5)); loaded once per (persistent) block
  const int cg = t % G;
  f32x4 wr[25];
#pragma unroll
  for (int tap = 0; tap < 25; ++tap)
    wr[tap] = (f32x4){p.w[(cg * 4 + 0) * 25 + tap], p.w[(cg * 4 + 1) * 25 + tap], p.w[(cg * 4 + 2) * 25 + tap], p.w[(cg * 4 + 3) * 25 + tap]};
  const float bias = p.bias ? p.bias[0] : 0.f;
  constexpr int NPX = (TH + 2) * (TW + 2);
  constexpr int NST = (NPX * G + 255) / 256;
  constexpr unsigned OOB = 0x80000000u;                        // >= num_records: the load returns zeros
  // a block walks a contiguous run of tiles, so the halo rows it shares with the previous tile are still in its XCD's L2
  const int tile_lo = (int)((long)ntiles * blockIdx.x / gridDim.x), tile_hi = (int)((long)ntiles * (blockIdx.x + 1) / gridDim.x);
  const unsigned coff = (unsigned)chan_off(0, 0, cg, G / 2, p.half) * 4u;
  f32x4 stage[NST];
  auto fetch = [&](int tile) {                                 // global -> registers: one tile plus halo, zero outside the image
    const int tw0 = (tile % tiles_w) * TW;
    const int th0 = ((tile / tiles_w) % tiles_h) * TH;
    const int b = tile / (tiles_w * tiles_h);
    // buffer loads: out-of-image pixels get the out-of-range offset and read as zero, with no branch around the load
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (long)b * p.H * p.W * p.ldx), 0, OOB, 0x00020000);
#pragma unroll
    for (int k = 0; k < NST; ++k) {
      const int px = t / G + k * (256 / G);
      const int lw = px % (TW + 2), lh = px / (TW + 2);
      const int ih = th0 - 1 + lh, iw = tw0 - 1 + lw;
      const bool ok = px < NPX && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
      const unsigned vo = ok ? (unsigned)((ih * p.W + iw) * (int)p.ldx) * 4u + coff : OOB;
      stage[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)vo, 0, 0));
    }
  };
  if (tile_lo < tile_hi) fetch(tile_lo);
  for (int tile = tile_lo; tile < tile_hi; ++tile) {
  const int tw0 = (tile % tiles_w) * TW;
  const int th0 = ((tile / tiles_w) % tiles_h) * TH;
  const long b = tile / (tiles_w * tiles_h);
  __syncthreads();                                             // the previous tile's readers are done
#pragma unroll
  for (int k = 0; k < NST; ++k) {
    const int px = t / G + k * (256 / G);
    if (px < NPX) *(f32x4*)(&tile_lds[px * LP + cg * 4]) = stage[k];
  }
  __syncthreads();
  if (tile + 1 < tile_hi) fetch(tile + 1);                     // in flight while this tile is computed
  for (int qd = t / G; qd < TH * TW; qd += 256 / G) {          // every lane of a wave runs the same trip count
    const int lc = qd % TW, la = qd / TW;
    f32x2 acc[2][2] = {{{0.f, 0.f}, {0.f, 0.f}}, {{0.f, 0.f}, {0.f, 0.f}}};   // channel pairs: one v_pk_fma_f32 per two MACs
#pragma unroll
    for (int dh = -1; dh <= 1; ++dh)
#pragma unroll
      for (int dw = -1; dw <= 1; ++dw) {
        const f32x4 v = *(const f32x4*)(&tile_lds[((la + 1 + dh) * (TW + 2) + lc + 1 + dw) * LP + cg * 4]);
        const f32x2 vlo = {v[0], v[1]}, vhi = {v[2], v[3]};
        const int th = 1 - dh, tw = 1 - dw;
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
          const int kh = ph + 2 * th;
          if (kh > 4) continue;
#pragma unroll
          for (int pw = 0; pw < 2; ++pw) {
            const int kw = pw + 2 * tw;
            if (kw > 4) continue;
            const f32x4 w4 = wr[kh * 5 + kw];
            acc[ph][pw] += vlo * (f32x2){w4[0], w4[1]};
            acc[ph][pw] += vhi * (f32x2){w4[2], w4[3]};
          }
        }
      }
    float o[2][2];
#pragma unroll
    for (int ph = 0; ph < 2; ++ph)
#pragma unroll
      for (int pw = 0; pw < 2; ++pw) {
        float r = acc[ph][pw][0] + acc[ph][pw][1];
        if (G >= 8) r += dpp_f32<0x141>(r);                    // row_half_mirror: lane i <-> 7-i within 8
        r += dpp_f32<0xB1>(r);                                 // quad_perm [1,0,3,2]
        r += dpp_f32<0x4E>(r);                                 // quad_perm [2,3,0,1]
        o[ph][pw] = r;
      }
    const int a = th0 + la, c = tw0 + lc;
    if (cg < 4 && a < p.H && c < p.W) {
      const int ph = cg >> 1, pw = cg & 1;
      float v = (cg == 0) ? o[0][0] : (cg == 1) ? o[0][1] : (cg == 2) ? o[1][0] : o[1][1];
      const int oh = 2 * a + ph, ow = 2 * c + pw;
      if (oh < p.Ho && ow < p.Wo) {
        v += bias;
        if (p.sigmoid) v = 1.f / (1.f + __expf(-v));
        p.y[(b * p.Ho + oh) * p.Wo + ow] = v;
      }
    }
  }
  }
}

int svs_deconv_to1_run(const float* x, long ldx, int B, int H, int W, int C, const float* w, const float* bias,
                       float* y, int Ho, int Wo, int apply_sigmoid, hipStream_t stream, const char* who, long half) {
  SVS_REQUIRE(x && w && y, "%s: null pointer", who);
  SVS_REQUIRE(C == 32 || C == 16, "%s: supports C=16/32, got %d", who, C);
  SVS_REQUIRE((Ho == 2 * H || Ho == 2 * H - 1) && (Wo == 2 * W || Wo == 2 * W - 1), "%s: output %dx%d unreachable from %dx%d", who, Ho, Wo, H, W);
  SVS_REQUIRE(ldx >= (half ? C / 2 : C) && ldx % 4 == 0 && svs_aligned16(x), "%s: bad input view", who);
  To1Args a{x, ldx, B, H, W, w, bias, y, Ho, Wo, apply_sigmoid, half};
  const long grid_l = (long)B * ((H + 7) / 8) * ((W + 31) / 32);
  SVS_REQUIRE(grid_l < (1L << 31), "%s: too many tiles", who);
  SVS_REQUIRE(((long)H * W * ldx + half) * 4 < (1L << 31), "%s: one image of the input view must span < 2 GiB", who);
  const int grid = (int)(grid_l < 512 ? grid_l : 512);     // persistent: 2 blocks per CU, weights loaded once per block
  if (C == 32) hipLaunchKernelGGL(deconv_to1_kernel<32>, dim3(grid), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(deconv_to1_kernel<16>, dim3(grid), dim3(256), 0, stream, a);
  SVS_CHECK_LAUNCH("deconv_to1");
  return SVS_OK;
}

// -------------------------------------------------------------------------------------------------
struct WgC1Args {
  const float* s; long lds; int B, Hs, Ws;   // (B,Hs,Ws,CS)
  const float* l; int Hl, Wl;                // (B,Hl,Wl) single channel
  float* slab;                               // [gridDim.x][CS*25]
  long pix_per_block;
  long half;                                 // see chan_off()
};

template <int CS>
__global__ __launch_bounds__(256) void wgrad_c1_kernel(WgC1Args p) {
  constexpr int G = CS / 4;
  constexpr int PL = 256 / G;                // pixels per sweep
  __shared__ float red[4][25 * CS];
  const int t = threadIdx.x;
  const int cg = t % G, pl = t / G;
  const long P = (long)p.B * p.Hs * p.Ws;
  const long p0 = (long)blockIdx.x * p.pix_per_block;
  long p1 = p0 + p.pix_per_block;
  if (p1 > P) p1 = P;
  f32x4 acc[25];
#pragma unroll
  for (int k = 0; k < 25; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (long pix = p0 + pl; pix < p1; pix += PL) {
    const int j = (int)(pix % p.Ws);
    const long tmp = pix / p.Ws;
    const int i = (int)(tmp % p.Hs);
    const long b = tmp / p.Hs;
    const f32x4 s4 = *(const f32x4*)(p.s + chan_off(pix, p.lds, cg, G / 2, p.half));
    const float* img = p.l + b * p.Hl * p.Wl;
    // The G lanes of a pixel need the same 25 window values.  Each loads only taps cg, cg + G, ... (the kernel was bound
    // by the 25 x G-fold redundant loads: 2 TB/s) and the group shares them with lane permutes.
    constexpr int NM = (25 + G - 1) / G;
    float mine[NM];
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      const int tap = cg + G * m;
      const int kh = tap / 5, kw = tap - kh * 5;
      const int ih = 2 * i - 2 + kh, iw = 2 * j - 2 + kw;
      float lv = 0.f;
      if (tap < 25 && (unsigned)ih < (unsigned)p.Hl && (unsigned)iw < (unsigned)p.Wl) lv = img[(long)ih * p.Wl + iw];
      mine[m] = lv;
    }
    const int group = (t & 63) & ~(G - 1);
#pragma unroll
    for (int tap = 0; tap < 25; ++tap) {
      const float lv = __shfl(mine[tap / G], group | (tap % G), 64);
      acc[tap] += s4 * lv;
    }
  }
  // lanes with equal cg inside a wave, then the 4 waves through LDS (fixed order: reproducible)
#pragma unroll
  for (int k = 0; k < 25; ++k)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float v = acc[k][c];
#pragma unroll
      for (int s = G; s < 64; s <<= 1) v += __shfl_xor(v, s, 64);
      acc[k][c] = v;
    }
  const int lane = t & 63, wave = t >> 6;
  if (lane < G) {
#pragma unroll
    for (int k = 0; k < 25; ++k)
#pragma unroll
      for (int c = 0; c < 4; ++c) red[wave][(lane * 4 + c) * 25 + k] = acc[k][c];
  }
  __syncthreads();
  float* out = p.slab + (long)blockIdx.x * (25 * CS);
  for (int i = t; i < 25 * CS; i += 256) out[i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
}

// The same gradient on the MFMA: dw[cs][tap] = sum_pixels S[pixel][cs] * L[window(pixel, tap)] is a GEMM with M = CS,
// N = 25 taps (two 16-column tiles, the last 7 columns unused) and K = pixels.  Each WAVE walks its own run of 4 x 16-pixel
// tiles: the tile's S vectors go through a private LDS region (pixel-major, pitch LS, so that the transposed MFMA
// A-operand read is conflict-free) and so does the 11 x 35 single-channel window; A = S^T (16 cs x 4 pixels), B = window
// values gathered per (pixel, tap) with a per-lane tap offset.  No block barriers; the accumulators live across all tiles
// of a wave and each wave writes one slab.  The VALU kernel above needs 25 x CS multiply-adds and, even with shared
// loads, 25 lane permutes per pixel and lane; here a pixel costs CS/16 x 2 / 4 MFMAs.
template <int CS>
__global__ __launch_bounds__(256) void wgrad_c1_mfma_kernel(WgC1Args p, int ntiles, int tiles_per_wave) {
  constexpr int TH = 4, TW = 16, MT = CS / 16;
  constexpr int LS = (CS == 32) ? 48 : 16;           // pixel pitch: four pixels x 16 cs -> 64 different banks
  constexpr int LH = 2 * TH + 3, LW = 2 * TW + 3, LWP = LW + 1;
  constexpr int NSP = TH * TW * (CS / 4) / 64;       // b128 pieces of S per lane and tile
  constexpr int NLP = (LH * LW + 63) / 64;           // window floats per lane and tile
  __shared__ __attribute__((aligned(16))) float Ssm[4][TH * TW * LS];
  __shared__ float Lsm[4][LH * LWP];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lrow = lane & 15, q = lane >> 4;
  float* const sS = Ssm[wave];
  float* const sL = Lsm[wave];
  const int tiles_w = (p.Ws + TW - 1) / TW, tiles_h = (p.Hs + TH - 1) / TH;
  const int gw = blockIdx.x * 4 + wave;
  const int tile_lo = gw * tiles_per_wave;
  const int tile_hi = min(tile_lo + tiles_per_wave, ntiles);

  f32x4 sreg[NSP];
  float lreg[NLP];
  auto fetch = [&](int tile) {
    const int tw0 = (tile % tiles_w) * TW, th0 = ((tile / tiles_w) % tiles_h) * TH;
    const long b = tile / (tiles_w * tiles_h);
#pragma unroll
    for (int r = 0; r < NSP; ++r) {
      const int e = lane + 64 * r;
      const int c4 = e % (CS / 4), px = e / (CS / 4);
      const int sh = th0 + px / TW, sw = tw0 + px % TW;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (sh < p.Hs && sw < p.Ws) v = *(const f32x4*)(p.s + chan_off((b * p.Hs + sh) * p.Ws + sw, p.lds, c4, CS / 8, p.half));
      sreg[r] = v;
    }
    const float* img = p.l + b * p.Hl * p.Wl;
#pragma unroll
    for (int r = 0; r < NLP; ++r) {
      const int e = lane + 64 * r;
      const int wr = e / LW, wc = e % LW;
      const int ih = 2 * th0 - 2 + wr, iw = 2 * tw0 - 2 + wc;
      float v = 0.f;
      if (e < LH * LW && (unsigned)ih < (unsigned)p.Hl && (unsigned)iw < (unsigned)p.Wl) v = img[(long)ih * p.Wl + iw];
      lreg[r] = v;
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int r = 0; r < NSP; ++r) {
      const int e = lane + 64 * r;
      *(f32x4*)(&sS[(e / (CS / 4)) * LS + (e % (CS / 4)) * 4]) = sreg[r];
    }
#pragma unroll
    for (int r = 0; r < NLP; ++r) {
      const int e = lane + 64 * r;
      if (e < LH * LW) sL[(e / LW) * LWP + e % LW] = lreg[r];
    }
  };
  f32x4 acc[MT][2];
#pragma unroll
  for (int i = 0; i < MT; ++i) { acc[i][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[i][1] = acc[i][0]; }
  // B operand: column lrow of tap tile nt is tap nt*16 + lrow (clamped: columns 25..31 are never stored); k = pixel q of
  // the k-step -> window offset of (tap) + two columns per pixel
  int boff[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int tap = min(nt * 16 + lrow, 24);
    boff[nt] = (tap / 5) * LWP + tap % 5 + 2 * q;
  }
  if (tile_lo < tile_hi) fetch(tile_lo);
  for (int tile = tile_lo; tile < tile_hi; ++tile) {
    stage();                                            // (one wave: LDS operations execute in program order)
    if (tile + 1 < tile_hi) fetch(tile + 1);
#pragma unroll
    for (int ks = 0; ks < TH * TW / 4; ++ks) {
      const int sh = ks / 4, sg = ks % 4;              // pixel (sh, 4 sg + q)
      float bv[2];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) bv[nt] = sL[boff[nt] + 2 * sh * LWP + 8 * sg];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const float a = sS[(sh * TW + sg * 4 + q) * LS + i * 16 + lrow];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[nt], acc[i][nt], 0, 0, 0);
      }
    }
  }
  // acc[i][nt][r]: row cs = 16 i + 4 q + r, column tap = 16 nt + lrow
  float* out = p.slab + (long)gw * (25 * CS);
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int tap = nt * 16 + lrow;
      if (tap < 25) {
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(i * 16 + 4 * q + r) * 25 + tap] = acc[i][nt][r];
      }
    }
}

// out[g][i] = sum of slabs z in group g (contiguous chunks of `per` slabs), in ascending z (a fixed order: bitwise reproducible).
// One float4 per thread and four slabs' loads in flight per step: the kernel is a pure stream of the slabs (650 MB per train
// step over all layers), and with one 4-byte load per dependent add it ran at 1.1-2.3 TB/s.  grid (ceil(n/4/256), groups); n % 4 == 0.
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ slab, int nslab, int per, long n,
                                                           float* __restrict__ out) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  const int z0 = blockIdx.y * per;
  int z1 = z0 + per;
  if (z1 > nslab) z1 = nslab;
  const float* src = slab + (long)z0 * n + i;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  int z = z0;
  for (; z + 4 <= z1; z += 4, src += 4 * n) {
    const f32x4 a = *(const f32x4*)src, b = *(const f32x4*)(src + n), c = *(const f32x4*)(src + 2 * n), d = *(const f32x4*)(src + 3 * n);
    s += a; s += b; s += c; s += d;                   // (same order as one at a time)
  }
  for (; z < z1; ++z, src += n) s += *(const f32x4*)src;
  *(f32x4*)(out + (long)blockIdx.y * n + i) = s;
}

int svs_reduce_slabs_run(const float* slab, int nslab, int per, int groups, long n, float* out, hipStream_t stream) {
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((n / 4 + 255) / 256), (unsigned)groups), dim3(256), 0, stream, slab, nslab, per, n, out);
  SVS_CHECK_LAUNCH("reduce_slabs");
  return SVS_OK;
}

#define C1_GROUPS 32

static int wgrad_c1_blocks(long P) {
  long nb = (P + 2047) / 2048;
  if (nb > 1024) nb = 1024;
  if (nb < 1) nb = 1;
  return (int)nb;
}

size_t svs_wgrad_c1_workspace(int B, int Hs, int Ws, int Cs) {
  (void)B; (void)Hs; (void)Ws;
  return (size_t)(1024 + C1_GROUPS) * 25 * Cs * sizeof(float);          // up to 1024 slabs (one per block or per wave)
}

int svs_wgrad_c1_run(const float* s, long lds, int B, int Hs, int Ws, int Cs, const float* l, int Hl, int Wl,
                     float* dw, void* ws, size_t ws_bytes, hipStream_t stream, const char* who, long half) {
  SVS_REQUIRE(s && l && dw, "%s: null pointer", who);
  SVS_REQUIRE(Cs == 16 || Cs == 32, "%s: supports Cs=16/32, got %d", who, Cs);
  SVS_REQUIRE(Hs == svs_conv_out(Hl) && Ws == svs_conv_out(Wl), "%s: grid mismatch", who);
  SVS_REQUIRE(lds >= (half ? Cs / 2 : Cs) && lds % 4 == 0 && svs_aligned16(s), "%s: bad view", who);
  const long P = (long)B * Hs * Ws;
  const int nb = wgrad_c1_blocks(P);
  const size_t need = svs_wgrad_c1_workspace(B, Hs, Ws, Cs);
  if (!ws || ws_bytes < need) {
    svs_set_error("%s: workspace too small (%zu < %zu)", who, ws_bytes, need);
    return SVS_ERR_WORKSPACE;
  }
  WgC1Args a{s, lds, B, Hs, Ws, l, Hl, Wl, (float*)ws, (P + nb - 1) / nb, half};
  int nslab = nb;
  const int ntiles = B * ((Hs + 3) / 4) * ((Ws + 15) / 16);
  if (ntiles >= 256 && !svs_tune_flag(SVS_TUNE_WGRAD_C1_VALU)) {          // MFMA kernel: one slab per wave, <= 1024 waves (workspace bound)
    int waves = ntiles < 1024 ? ntiles : 1024;
    waves = (waves + 3) / 4 * 4;
    if (waves > 1024) waves = 1024;
    const int tpw = (ntiles + waves - 1) / waves;
    waves = ((ntiles + tpw - 1) / tpw + 3) / 4 * 4;             // every launched wave writes its slab (possibly all zeros)
    nslab = waves;
    if (Cs == 16) hipLaunchKernelGGL(wgrad_c1_mfma_kernel<16>, dim3(waves / 4), dim3(256), 0, stream, a, ntiles, tpw);
    else hipLaunchKernelGGL(wgrad_c1_mfma_kernel<32>, dim3(waves / 4), dim3(256), 0, stream, a, ntiles, tpw);
  } else if (Cs == 16) hipLaunchKernelGGL(wgrad_c1_kernel<16>, dim3(nb), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(wgrad_c1_kernel<32>, dim3(nb), dim3(256), 0, stream, a);
  SVS_CHECK_LAUNCH("wgrad_c1");
  const long n = 25L * Cs;
  float* tmp = (float*)ws + (size_t)nslab * n;
  const int groups = nslab < C1_GROUPS ? nslab : C1_GROUPS;
  const int per = (nslab + groups - 1) / groups;
  const unsigned gx = (unsigned)((n / 4 + 255) / 256);
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3(gx, (unsigned)groups), dim3(256), 0, stream, (const float*)ws, nslab, per, n, tmp);
  SVS_CHECK_LAUNCH("reduce_slabs");
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3(gx, 1), dim3(256), 0, stream, (const float*)tmp, groups, groups, n, dw);
  SVS_CHECK_LAUNCH("reduce_slabs");
  return SVS_OK;
}
