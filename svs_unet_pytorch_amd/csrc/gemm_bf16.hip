// bf16 eval forward of the U-Net on the bf16 MFMA (v_mfma_f32_16x16x32_bf16, fp32 accumulate), gfx950 -- BASELINE.json
// configs[4] ("bf16 convs on MFMA"); the fp32 path (gemm_conv.hip) stays the parity reference.
//
// Same implicit GEMM as gemm_conv.hip (GATHER = Conv2d forward, PARITY = the four stride-1 sub-convolutions of a
// ConvTranspose2d forward; reference model.py:47-109), with the operand layout chosen for the bf16 instruction:
//   * activations are bf16 NHWC; a K-tile is 32 consecutive input channels of one tap = ONE 64-byte run per pixel, the same
//     run a 16-channel fp32 K-tile is, so staging (16-byte buffer loads, XOR-swizzled 64-byte LDS rows, double buffering,
//     one barrier per K-tile) is identical;
//   * the 16-byte chunk a lane reads back with ds_read_b128 holds k = 8q .. 8q+7 of its row (q = lane >> 4): exactly the
//     operand map of the 16x16x32 instruction (lane l: row l & 15, k = 8 (l >> 4) + j), so one fragment read feeds one MFMA
//     that covers the whole K-tile (the fp32 form needs four 16x16x4 MFMAs per fragment);
//   * eval-mode BatchNorm is folded: the per-channel scale goes into the bf16 weights, the shift and LeakyReLU / ReLU into
//     the epilogue, which rounds to bf16 once (v_cvt_pk_bf16_f32) and stores 32-byte runs.
// Levels 2..5 are interleaved ([decoder half | skip half] per pixel); level 1 is PLANAR (a 16-channel decoder plane and a
// 16-channel skip plane, 32 bytes per pixel each), so that conv1 / deconv5 write and conv2 reads whole HBM bursts instead of
// 32 bytes of every 64.  Bound: HBM / launch at streaming batch sizes (bf16 MFMA peak ~2.5 PFLOP/s: 16x the fp32 rate).
#include <type_traits>

#include "internal.h"

typedef unsigned short u16;
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ u16 to_bf16(float v) { return __builtin_bit_cast(u16, (__bf16)v); }
__device__ __forceinline__ float from_bf16(u16 v) { return __builtin_bit_cast(float, (unsigned)v << 16); }

struct ConvBf16Args {
  const u16* x; long ldx;          // bf16 NHWC view, ld in elements
  int B, H, W, C;                  // C % 32 == 0
  const u16* wp;                   // packed bf16 weights, BatchNorm scale folded in: gather [n][tap][c], parity [class][n][th][tw][c]
  const float* shift;              // [N] fp32 (folded BatchNorm shift incl. conv bias)
  float slope;                     // LeakyReLU slope (0 = ReLU)
  u16* y; long ldy;
  int Ho, Wo, N;
  int ksplit; float* slab;         // fp32 partial sums [ksplit][B*Ho*Wo][N] when ksplit > 1
};

template <int N_, int I_ = 0, class F>
__device__ __forceinline__ void bf_static_for(F&& f) {          // f(integral_constant<int, I>) for I = 0 .. N-1, unrolled
  if constexpr (I_ < N_) {
    f(std::integral_constant<int, I_>{});
    bf_static_for<N_, I_ + 1>(f);
  }
}

enum { BF_GATHER = 0, BF_PARITY = 1 };
__device__ __forceinline__ int swz16(int row, int chunk) { return chunk ^ ((row >> 1) & 3); }

template <int MODE, int BM, int BN, int WM, int WN, int KB = 1>      // KB: K-tiles (32 channels of one tap each) per barrier
__global__ __launch_bounds__(256) void conv_gemm_bf16_kernel(ConvBf16Args p) {
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
  constexpr int RA = (BM + 63) / 64, RB = (BN + 63) / 64;
  static_assert(WM * WN == 4 && TM >= 1 && TN >= 1, "tile");
  __shared__ __attribute__((aligned(16))) float As[2][KB][BM * 16];     // 64-byte rows (32 bf16)
  __shared__ __attribute__((aligned(16))) float Bs[2][KB][BN * 16];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave / WN, wn = wave % WN, lrow = lane & 15, q = lane >> 4;
  int ph = 0, pw = 0, nth = 5, ntw = 5, Ha, Wa;
  const u16* wp = p.wp;
  if (MODE == BF_PARITY) {
    const int par = blockIdx.z;
    ph = par >> 1; pw = par & 1;
    nth = 3 - ph; ntw = 3 - pw;
    Ha = (p.Ho - ph + 1) >> 1; Wa = (p.Wo - pw + 1) >> 1;
    const int poff = (par == 0) ? 0 : (par == 1) ? 9 : (par == 2) ? 15 : 21;
    wp += (long)poff * p.N * p.C;
  } else { Ha = p.Ho; Wa = p.Wo; }
  const int ntaps = nth * ntw;
  const long M = (long)p.B * Ha * Wa;
  const int ntile_n = p.N / BN;
  const long m0 = (long)(blockIdx.x / ntile_n) * BM;
  const int n0 = (blockIdx.x % ntile_n) * BN;
  if (m0 >= M) return;
  const int cpt = p.C >> 5;                                   // K-tiles (32 channels) per tap
  const int nkt = ntaps * cpt;
  const int kt_begin = (int)((long)nkt * blockIdx.y / p.ksplit), kt_end = (int)((long)nkt * (blockIdx.y + 1) / p.ksplit);
  const long Kw = (long)ntaps * p.C;
  const int chunk = t & 3;
  constexpr unsigned OOB = 0x80000000u;
  unsigned a_voff[RA], a_mask[RA];
#pragma unroll
  for (int r = 0; r < RA; ++r) {
    const int row = (t >> 2) + 64 * r;
    const long m = m0 + row;
    const bool ok = (row < BM) && (m < M);
    const long mm = ok ? m : 0;
    const unsigned um = (unsigned)mm, utmp = um / (unsigned)Wa;      // (M < 2^31, checked on the host: 32-bit divisions)
    const int wq = (int)(um - utmp * (unsigned)Wa);
    const unsigned ub = utmp / (unsigned)Ha;
    const int hq = (int)(utmp - ub * (unsigned)Ha);
    const long b = ub;
    const int h0 = (MODE == BF_GATHER) ? 2 * hq : hq, w0 = (MODE == BF_GATHER) ? 2 * wq : wq;
    a_voff[r] = ok ? (unsigned)((((b * p.H + h0) * p.W + w0) * p.ldx + chunk * 8) * 2) : OOB;
    unsigned mask = 0;
    for (int th = 0; th < nth; ++th)
      for (int tw = 0; tw < ntw; ++tw) {
        const int ih = (MODE == BF_GATHER) ? h0 - 2 + th : h0 + 1 - th, iw = (MODE == BF_GATHER) ? w0 - 2 + tw : w0 + 1 - tw;
        if (ok && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W) mask |= 1u << (th * ntw + tw);
      }
    a_mask[r] = mask;
  }
  unsigned b_voff[RB];
#pragma unroll
  for (int r = 0; r < RB; ++r) {
    const int row = (t >> 2) + 64 * r;
    b_voff[r] = (row < BN) ? (unsigned)(((long)(n0 + row) * Kw + chunk * 8) * 2) : OOB;
  }
  const long shift_px = (MODE == BF_GATHER) ? (2L * p.W + 2) * p.ldx : (1L * p.W + 1) * p.ldx;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x - shift_px), 0, OOB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)wp, 0, OOB, 0x00020000);
  // A "step" is KB consecutive K-tiles behind ONE barrier: a bf16 K-tile is only TM * TN MFMAs of 16 cycles per wave (128-256
  // cycles), far less than a barrier plus an LDS round trip (rocprofv3 at 216 tiles, one K-tile per barrier: MFMA-busy
  // 0.10-0.23, waves parked on waitcnt / barriers 0.41-0.63 of their cycles).  Two register sets of a whole step each: the
  // request for step s + 2 is issued when step s starts, so an operand has a full step to arrive before it goes to LDS.
  f32x4 ra[2][KB][RA], rb[2][KB][RB];
  auto load_step = [&](int kt0, auto setc) __attribute__((always_inline)) {      // tap outer, channel chunk inner; past kt_end: zeros
    constexpr int set_ = decltype(setc)::value;
#pragma unroll
    for (int u = 0; u < KB; ++u) {
      const int kt = kt0 + u;
      const bool live = kt < kt_end;                          // (block-uniform)
      const int ktc = live ? kt : kt_begin;
      const int tap = ktc / cpt, cc = ktc - tap * cpt;
      const int th = (ntw == 5) ? tap / 5 : (ntw == 3) ? tap / 3 : tap >> 1;
      const int tw = tap - th * ntw;
      const int pix = (MODE == BF_GATHER) ? th * p.W + tw : (2 - th) * p.W + (2 - tw);
      const int soff_a = __builtin_amdgcn_readfirstlane((int)((pix * p.ldx + (cc << 5)) * 2));
      const int soff_b = __builtin_amdgcn_readfirstlane((int)(((long)tap * p.C + (cc << 5)) * 2));
#pragma unroll
      for (int r = 0; r < RA; ++r) {
        const unsigned vo = (live && ((a_mask[r] >> tap) & 1u)) ? a_voff[r] : OOB;
        ra[set_][u][r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)vo, soff_a, 0));
      }
#pragma unroll
      for (int r = 0; r < RB; ++r)
        rb[set_][u][r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, live ? (int)b_voff[r] : (int)OOB, soff_b, 0));
    }
  };
  auto store_step = [&](int buf, auto setc) __attribute__((always_inline)) {
    constexpr int set_ = decltype(setc)::value;
#pragma unroll
    for (int u = 0; u < KB; ++u) {
#pragma unroll
      for (int r = 0; r < RA; ++r) {
        const int row = (t >> 2) + 64 * r;
        if (BM % 64 == 0 || row < BM) *(f32x4*)(&As[buf][u][row * 16 + swz16(row, chunk) * 4]) = ra[set_][u][r];
      }
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        const int row = (t >> 2) + 64 * r;
        if (BN % 64 == 0 || row < BN) *(f32x4*)(&Bs[buf][u][row * 16 + swz16(row, chunk) * 4]) = rb[set_][u][r];
      }
    }
  };
  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int nstep = (kt_end - kt_begin + KB - 1) / KB;
  if (nstep > 0) load_step(kt_begin, std::integral_constant<int, 0>{});
  if (nstep > 1) load_step(kt_begin + KB, std::integral_constant<int, 1>{});
  if (nstep > 0) store_step(0, std::integral_constant<int, 0>{});
  __syncthreads();
  for (int s0 = 0; s0 < nstep; s0 += 2) {
    bf_static_for<2>([&](auto dc) {
      constexpr int d_ = decltype(dc)::value;
      const int st = s0 + d_;
      if (st < nstep) {                                       // (block-uniform)
        const int buf = d_;                                   // step st lives in LDS buffer st & 1 = d_ (s0 is even)
        if (st + 2 < nstep) load_step(kt_begin + (st + 2) * KB, dc);      // register set d_ is free: step st went to LDS one step ago
#pragma unroll
        for (int u = 0; u < KB; ++u) {
          bf16x8 fa[TM], fb[TN];
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const int row = wm * (TM * 16) + i * 16 + lrow;
            fa[i] = __builtin_bit_cast(bf16x8, *(const f32x4*)(&As[buf][u][row * 16 + swz16(row, q) * 4]));
          }
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const int row = wn * (TN * 16) + j * 16 + lrow;
            fb[j] = __builtin_bit_cast(bf16x8, *(const f32x4*)(&Bs[buf][u][row * 16 + swz16(row, q) * 4]));
          }
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);   // weights first: see the epilogue
        }
        if (st + 1 < nstep) store_step(buf ^ 1, std::integral_constant<int, (d_ + 1) % 2>{});
        __syncthreads();
      }
    });
  }
  // epilogue.  The weights are the FIRST MFMA operand, so the D tile is [channel][pixel]: column lane & 15 = GEMM row (pixel),
  // rows 4 (lane >> 4) + reg = four CONSECUTIVE output channels -- one 8-byte bf16 store (or one 16-byte slab store) per lane
  // and tile instead of four 2-byte ones, and one pixel decode per lane and row tile
  const bool split = p.ksplit > 1;
  float* const slab = split ? p.slab + (long)blockIdx.y * ((long)p.B * p.Ho * p.Wo) * p.N : nullptr;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const long m = m0 + wm * (TM * 16) + i * 16 + lrow;
    if (m >= M) continue;
    long opix;
    if (MODE == BF_GATHER) opix = m;
    else {
      const unsigned um = (unsigned)m, utmp = um / (unsigned)Wa;
      const int wq = (int)(um - utmp * (unsigned)Wa);
      const unsigned ub = utmp / (unsigned)Ha;
      const int hq = (int)(utmp - ub * (unsigned)Ha);
      const long b = ub;
      opix = (b * p.Ho + 2 * hq + ph) * p.Wo + 2 * wq + pw;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * (TN * 16) + j * 16 + q * 4;
      if (split) { *(f32x4*)(slab + opix * p.N + n) = acc[i][j]; continue; }
      const f32x4 sh = *(const f32x4*)(p.shift + n);
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float u = acc[i][j][r] + sh[r]; v[r] = u > 0.f ? u : u * p.slope; }
      uint2 o;
      o.x = (unsigned)to_bf16(v[0]) | ((unsigned)to_bf16(v[1]) << 16);
      o.y = (unsigned)to_bf16(v[2]) | ((unsigned)to_bf16(v[3]) << 16);
      *(uint2*)(p.y + opix * p.ldy + n) = o;
    }
  }
}



// ------------------------------------------------------------------------------------------------
// LDS-window form of the PARITY mode for the shallow decoder layers (deconv4: 128 -> 32, deconv5: 64 -> 16 channels; 2/3 of
// the bf16 forward's time in the GEMM form: with 16 / 32 output channels an im2col element feeds one or two MFMAs, so the
// GEMM is bound by moving each input element ~6 times through L2 / LDS, not by the MFMA).  A block owns 8 x 16 anchor
// pixels of one image (= a 16 x 32 output patch, all four parity classes), stages that window plus a one-pixel halo ONCE
// (zero-filled outside the image; pixel pitch 2C + 16 bytes: conflict-free ds_read_b128), and every tap of every class
// reads its A fragments at base + compile-time offset.  Weight fragments come from global memory (shared by all blocks:
// L2-resident), one tap ahead.  Same structure as parity_window_kernel of gemm_conv.hip; one MFMA covers 32 channels.
// ------------------------------------------------------------------------------------------------
// TW = 16: tiles of 8 x 16 anchors (a 16-anchor MFMA row tile is one anchor row); TW = 8: tiles of 16 x 8 anchors (a row tile is two
// anchor rows of 8) for deconv3, whose input is 32 x 8 anchors per image.  The window and the weight buffers live in dynamic LDS
// (deconv3: 180 pixels x 528 bytes + 2 x 64 x 528 bytes = 159 KB, one block per CU).
template <int C, int TN, int TW = 16>
__global__ __launch_bounds__(256) void parity_window_bf16_kernel(ConvBf16Args p) {
  constexpr int TH = 128 / TW, TM = 2, RPT = 16 / TW;            // RPT: anchor rows per 16-anchor row tile
  constexpr int LPB = C * 2 + 16;                               // bytes per staged pixel
  constexpr int WW = TW + 2, NPX = (TH + 2) * WW;
  constexpr int CQ = C / 8, CC = C / 32;
  constexpr int NST = (NPX * CQ + 255) / 256;
  constexpr unsigned OOB = 0x80000000u;
  constexpr int POFF[4] = {0, 9, 15, 21};
  extern __shared__ __attribute__((aligned(16))) unsigned char pw_smem[];
  unsigned char* const win = pw_smem;                            // [NPX * LPB]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lrow = lane & 15, q = lane >> 4;
  const int tiles_w = (p.W + TW - 1) / TW, tiles_h = (p.H + TH - 1) / TH;
  const int ntiles = p.B * tiles_h * tiles_w;
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, OOB, 0x00020000);
  // persistent blocks: a block walks tiles blockIdx.x, + gridDim.x, ... so that resident weights (WALL) are fetched once per
  // block instead of once per tile (51 KB per 128 anchors was most of the L2 traffic)
  int th0 = 0, tw0 = 0;
  long b = 0;
  // The window of the NEXT tile is requested (global -> registers) before the current tile's taps run and written to LDS
  // after them: with one tile in flight per block the ~2 us of load latency were as long as the tile's MFMAs.
  f32x4 stage[NST];
  auto fetch_window = [&](int tile) __attribute__((always_inline)) {
    const int ftw0 = (tile % tiles_w) * TW, fth0 = ((tile / tiles_w) % tiles_h) * TH;
    const long fb_ = tile / (tiles_w * tiles_h);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + fb_ * p.H * p.W * p.ldx), 0, OOB, 0x00020000);
#pragma unroll
    for (int k = 0; k < NST; ++k) {
      const int e = t + k * 256;
      const int cq = e % CQ, px = e / CQ;
      const int lw = px % WW, lh = px / WW;
      const int ih = fth0 - 1 + lh, iw = ftw0 - 1 + lw;
      const bool ok = px < NPX && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
      const unsigned vo = ok ? (unsigned)(((ih * p.W + iw) * (int)p.ldx + cq * 8) * 2) : OOB;
      stage[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)vo, 0, 0));
    }
  };
  auto commit_window = [&](int tile) __attribute__((always_inline)) {           // registers -> LDS; this tile becomes the current one
    tw0 = (tile % tiles_w) * TW;
    th0 = ((tile / tiles_w) % tiles_h) * TH;
    b = tile / (tiles_w * tiles_h);
#pragma unroll
    for (int k = 0; k < NST; ++k) {
      const int e = t + k * 256;
      const int cq = e % CQ, px = e / CQ;
      if (px < NPX) *(f32x4*)(&win[px * LPB + cq * 16]) = stage[k];
    }
  };
  // A fragment of row tile i: anchor (2 wave + i, lrow) at window coordinate (+1, +1), 16-byte chunk q of a 32-channel step
  const unsigned char* abase = &win[((RPT * 2 * wave + lrow / TW + 1) * WW + lrow % TW + 1) * LPB + q * 16];
  // Weights of one tap ([n][C] bf16, N * C * 2 bytes) are brought into LDS ONCE per block and tap (double-buffered, one
  // barrier per tap) and every wave reads its B fragments from there: fetched per wave from L2 they were 4x the traffic
  // (1.4 GB per launch at 216 tiles -- the L2, not the MFMA, set the kernel's time).
  constexpr int N = TN * 16, WCH = N * CQ;                         // 16-byte pieces per tap
  constexpr int NWL = (WCH + 255) / 256;
  // weight row pitch: padded by 16 bytes, or -- when all 25 taps are resident and the padding would cost the second block
  // per CU -- unpadded with the 16-byte pieces XOR-swizzled by the row (2-way conflicts instead of 16-way)
  constexpr bool WSWZ = 25 * N * (C * 2 + 16) <= 60 * 1024;
  constexpr int WPB = WSWZ ? C * 2 : C * 2 + 16;
  auto wpos = [](int n, int piece) { return WSWZ ? (piece ^ (n & 7)) : piece; };
  // deconv5's 25 taps are 51 KB in all: they are staged once, up front, and the tap loop has no barriers at all (with 4 MFMAs
  // per wave and tap a barrier per tap cost more than the MFMAs); deconv4's 205 KB go through two per-tap buffers
  constexpr bool WALL = WSWZ;
  constexpr int NBUF = WALL ? 25 : 2;
  unsigned char (*const wts)[N * WPB] = (unsigned char (*)[N * WPB])(pw_smem + NPX * LPB);      // [NBUF][N * WPB]
  static_assert((NPX * LPB) % 16 == 0, "weight buffers 16-byte aligned");
  // per-tap weights (deconv4): DEPTH register sets, so that the request for tap s + DEPTH (wrapping into the next tile: the
  // weights do not depend on the tile) is in flight during DEPTH taps -- with one tap ahead every tap waited an L2 round trip
  constexpr int DEPTH = WALL ? 1 : 5;                           // (must divide 25: tap s lives in set s % DEPTH across the tile boundary)
  f32x4 wreg[DEPTH][NWL];
  auto fetch_w = [&](auto sc, auto setc) __attribute__((always_inline)) {         // global -> register set: weights of step sc
    constexpr int s_ = decltype(sc)::value;
    constexpr int set_ = decltype(setc)::value;
    constexpr int par = s_ < 9 ? 0 : s_ < 15 ? 1 : s_ < 21 ? 2 : 3;
    constexpr int tap = s_ - POFF[par];
    constexpr int ntaps = (3 - (par >> 1)) * (3 - (par & 1));
#pragma unroll
    for (int k = 0; k < NWL; ++k) {
      const int e = t + k * 256, n = e / CQ, cq = e - n * CQ;
      const unsigned vo = e < WCH ? (unsigned)((n * ntaps * C + cq * 8) * 2) : OOB;
      wreg[set_][k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, (int)vo, (POFF[par] * N * C + tap * C) * 2, 0));
    }
  };
  auto stash_w = [&](int buf, auto setc) __attribute__((always_inline)) {
    constexpr int set_ = decltype(setc)::value;
#pragma unroll
    for (int k = 0; k < NWL; ++k) {
      const int e = t + k * 256, n = e / CQ, cq = e - n * CQ;
      if (e < WCH) *(f32x4*)(&wts[buf][n * WPB + wpos(n, cq) * 16]) = wreg[set_][k];
    }
  };
  auto step = [&](auto sc, f32x4 (&acc)[TM][TN], int wbuf) {
    constexpr int s_ = decltype(sc)::value;
    constexpr int par = s_ < 9 ? 0 : s_ < 15 ? 1 : s_ < 21 ? 2 : 3;
    constexpr int tap = s_ - POFF[par];
    constexpr int ntw = 3 - (par & 1);
    constexpr int th = tap / ntw, tw = tap % ntw;
    constexpr int aoff = ((1 - th) * WW + (1 - tw)) * LPB;
    const unsigned char* wb = &wts[WALL ? s_ : wbuf][lrow * WPB];
#pragma unroll
    for (int cc = 0; cc < CC; ++cc) {
      bf16x8 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = __builtin_bit_cast(bf16x8, *(const f32x4*)(abase + aoff + i * RPT * WW * LPB + cc * 64));
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = __builtin_bit_cast(bf16x8, *(const f32x4*)(wb + j * 16 * WPB + wpos(lrow, cc * 4 + q) * 16));
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);   // weights first: see store_class
    }
  };
  // The weights are the FIRST MFMA operand, so the D tile is [channel][pixel]: column lrow = anchor (th0 + 2 wave + i, tw0 + lrow),
  // rows q*4 + r = output channels (+16 j) -- four consecutive channels per lane, one 8-byte store (2-byte stores of one
  // channel per lane were most of the kernel's time: deconv5 134 -> us at 216 tiles)
  float shv[TN][4];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) shv[j][r] = p.shift[j * 16 + q * 4 + r];
  auto store_class = [&](int par, const f32x4 (&acc)[TM][TN]) __attribute__((always_inline)) {
    const int ph = par >> 1, pw = par & 1;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int a = th0 + RPT * (2 * wave + i) + lrow / TW, oh = 2 * a + ph;
      const int c = tw0 + lrow % TW, ow = 2 * c + pw;
      if (a >= p.H || oh >= p.Ho || c >= p.W || ow >= p.Wo) continue;
      u16* const dst = p.y + ((b * p.Ho + oh) * p.Wo + ow) * p.ldy + q * 4;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float u = acc[i][j][r] + shv[j][r]; v[r] = u > 0.f ? u : u * p.slope; }
        uint2 o;
        o.x = (unsigned)to_bf16(v[0]) | ((unsigned)to_bf16(v[1]) << 16);
        o.y = (unsigned)to_bf16(v[2]) | ((unsigned)to_bf16(v[3]) << 16);
        *(uint2*)(dst + j * 16) = o;
      }
    }
  };
  f32x4 acc[TM][TN];
  using I0 = std::integral_constant<int, 0>;
  if constexpr (WALL) bf_static_for<25>([&](auto sc) { fetch_w(sc, I0{}); stash_w(decltype(sc)::value, I0{}); });
  else {
    bf_static_for<DEPTH>([&](auto sc) { fetch_w(sc, sc); });
    stash_w(0, I0{});
  }
  int gstep = 0;                               // taps done so far (all tiles): its parity is the weight buffer of the current tap
  // (deconv3: 23 pieces of window per thread -- no registers left to hold the next tile's beside five weight sets; a block has
  // at most two tiles there, the window is fetched at the start of each)
  constexpr bool PREFETCH = NST <= 12;
  if (PREFETCH && (int)blockIdx.x < ntiles) fetch_window(blockIdx.x);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    if constexpr (WALL) __syncthreads();       // the previous tile's readers are done with the window (per-tap kernels: the tap barrier)
    if constexpr (!PREFETCH) fetch_window(tile);
    commit_window(tile);
    if (PREFETCH && tile + (int)gridDim.x < ntiles) fetch_window(tile + gridDim.x);
    __syncthreads();
    bf_static_for<25>([&](auto sc) {
      constexpr int s_ = decltype(sc)::value;
      constexpr int par = s_ < 9 ? 0 : s_ < 15 ? 1 : s_ < 21 ? 2 : 3;
      if constexpr (s_ == POFF[par]) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      if constexpr (!WALL) fetch_w(std::integral_constant<int, (s_ + DEPTH) % 25>{}, std::integral_constant<int, s_ % DEPTH>{});
      step(sc, acc, gstep & 1);
      if constexpr (s_ == 24 || s_ + 1 == POFF[par < 3 ? par + 1 : 3]) store_class(par, acc);
      if constexpr (!WALL) {
        stash_w((gstep + 1) & 1, std::integral_constant<int, (s_ + 1) % DEPTH>{});   // the other buffer: its last readers passed the barrier that ended the previous tap
        __syncthreads();
        ++gstep;
      }
    });
  }
}

// out[pix][n] = bf16(act(sum_z slab[z][pix][n] + shift[n]))
__global__ __launch_bounds__(256) void splitk_epilogue_bf16_kernel(const float* __restrict__ slab, int ksplit, long P, int N,
                                                                   const float* __restrict__ shift, float slope, u16* y, long ldy) {
  const long total4 = P * N / 4, stride = P * N;
  int n_shift = -1;                                             // N is a power of two for every layer: no 64-bit division per float4
  if ((N & (N - 1)) == 0) n_shift = 31 - __builtin_clz((unsigned)N);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
    const long e = i * 4, pix = n_shift >= 0 ? (e >> n_shift) : e / N;
    const int n = (int)(e - pix * N);
    f32x4 s = *(const f32x4*)(slab + e);
    for (int z = 1; z < ksplit; ++z) s += *(const f32x4*)(slab + z * stride + e);
    u16 o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { float v = s[k] + shift[n + k]; v = v > 0.f ? v : v * slope; o[k] = to_bf16(v); }
    *(uint2*)(y + pix * ldy + n) = make_uint2((unsigned)o[0] | ((unsigned)o[1] << 16), (unsigned)o[2] | ((unsigned)o[3] << 16));
  }
}

// conv1 (model.py:47-51): 1 -> 16 channels, 5x5, stride 2, fp32 input tile -> the skip plane of level 1.  One input channel gives no channel reduction, so the K of the MFMA is the 25 taps (padded to 32): first operand =
// weights [16 n][32 k], second = the im2col patch of 16 output pixels, gathered from an fp32 LDS window (lane (pixel, q) reads
// the taps 8q .. 8q+7 of its patch).  The input keeps its fp32 information: x = hi + lo in two bf16 limbs, two MFMAs per 16
// pixels against the same weights.  400 FMAs per pixel on the VALU (118 us at 216 tiles) become two MFMAs per 16 pixels; what
// is left is the gather (8 ds_read_b32 + ~24 VALU per lane) and HBM: 57 MB in, 113 MB out.
struct Conv1Args { const float* x; int B, H, W; const u16* w1b; const float* shift; float slope; u16* y; long ldy; int Ho, Wo; };
__global__ __launch_bounds__(256) void conv1_mfma_bf16_kernel(Conv1Args p) {
  constexpr int TH = 16, TW = 32, WR = 2 * TH + 3, WC = 2 * TW + 4;              // window rows; columns (67 used)
  constexpr int NST = (WR * WC + 255) / 256;                                     // window samples per thread (10)
  constexpr unsigned OOB = 0x80000000u;
  __shared__ unsigned win[WR * WC];            // a sample as its two bf16 limbs, hi | lo << 16: split ONCE here, not once per tap that reads it
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, lrow = lane & 15, q = lane >> 4;
  const int tiles_w = (p.Wo + TW - 1) / TW, tiles_h = (p.Ho + TH - 1) / TH;
  const int ntiles = p.B * tiles_h * tiles_w;
  // Persistent blocks with the NEXT tile's window in flight (registers) while the current tile is computed: as one block per tile
  // the kernel was a chain of load latency -> barrier -> gather -> store per block (59 us for 170 MB at 216 tiles).  A thread's
  // samples are the same window positions in every tile: row, column and byte offset are computed once, the tile's position
  // is the scalar offset of a buffer load, a sample outside the image is pointed past num_records and reads as zero.
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x - (2L * p.W + 2)), 0, OOB, 0x00020000);
  int s_wr[NST], s_wc[NST]; unsigned s_rel[NST];
#pragma unroll
  for (int k = 0; k < NST; ++k) {
    const int e = t + k * 256;
    s_wr[k] = e / WC; s_wc[k] = e - s_wr[k] * WC;
    if (e >= WR * WC) s_wr[k] = -(1 << 20);
    s_rel[k] = (unsigned)(((long)s_wr[k] * p.W + s_wc[k]) * 4);
  }
  float stage[NST];
  auto fetch_window = [&](int tile) __attribute__((always_inline)) {
    const int fow0 = (tile % tiles_w) * TW, foh0 = ((tile / tiles_w) % tiles_h) * TH;
    const int fb_ = tile / (tiles_w * tiles_h);
    const int soff = __builtin_amdgcn_readfirstlane((int)((((long)fb_ * p.H + 2 * foh0) * p.W + 2 * fow0) * 4));
#pragma unroll
    for (int k = 0; k < NST; ++k) {
      const bool ok = (unsigned)(2 * foh0 - 2 + s_wr[k]) < (unsigned)p.H && (unsigned)(2 * fow0 - 2 + s_wc[k]) < (unsigned)p.W;
      stage[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, (int)(ok ? s_rel[k] : OOB), soff, 0));
    }
  };
  const bf16x8 fw = *(const bf16x8*)(p.w1b + lrow * 32 + q * 8);
  int off[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { const int k = q * 8 + e < 25 ? q * 8 + e : 24; off[e] = (k / 5) * WC + (k % 5); }   // (taps 25..31: zero weights)
  const f32x4 sh = *(const f32x4*)(p.shift + q * 4);
  if ((int)blockIdx.x < ntiles) fetch_window(blockIdx.x);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int ow0 = (tile % tiles_w) * TW, oh0 = ((tile / tiles_w) % tiles_h) * TH;
    const long b = tile / (tiles_w * tiles_h);
    __syncthreads();                                                               // the previous tile's readers are done
#pragma unroll
    for (int k = 0; k < NST; ++k) {
      const int e = t + k * 256;
      const float x = stage[k];
      const unsigned hi = to_bf16(x);
      if (e < WR * WC) win[e] = hi | ((unsigned)to_bf16(x - __builtin_bit_cast(float, hi << 16)) << 16);
    }
    if (tile + (int)gridDim.x < ntiles) fetch_window(tile + gridDim.x);
    __syncthreads();
#pragma unroll
    for (int g = 0; g < 8; ++g) {                                                  // wave: output rows 4 wave .. 4 wave + 3, two 16-pixel groups each
      const int ohl = 4 * wave + (g >> 1), owl = (g & 1) * 16 + lrow;
      const unsigned* const base = &win[(2 * ohl) * WC + 2 * owl];
      unsigned xv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) xv[e] = base[off[e]];
      unsigned hi[4], lo[4];
#pragma unroll
      for (int h = 0; h < 4; ++h) {                                                // v_perm_b32: the two samples' hi (lo) halves side by side
        hi[h] = __builtin_amdgcn_perm(xv[2 * h + 1], xv[2 * h], 0x05040100u);
        lo[h] = __builtin_amdgcn_perm(xv[2 * h + 1], xv[2 * h], 0x07060302u);
      }
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw, __builtin_bit_cast(bf16x8, (uint4){lo[0], lo[1], lo[2], lo[3]}), acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw, __builtin_bit_cast(bf16x8, (uint4){hi[0], hi[1], hi[2], hi[3]}), acc, 0, 0, 0);
      const int oh = oh0 + ohl, ow = ow0 + owl;
      if (oh >= p.Ho || ow >= p.Wo) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float u = acc[r] + sh[r]; v[r] = u > 0.f ? u : u * p.slope; }
      uint2 o;
      o.x = (unsigned)to_bf16(v[0]) | ((unsigned)to_bf16(v[1]) << 16);
      o.y = (unsigned)to_bf16(v[2]) | ((unsigned)to_bf16(v[3]) << 16);
      *(uint2*)(p.y + ((b * p.Ho + oh) * p.Wo + ow) * p.ldy + q * 4) = o;
    }
  }
}
// w1b[n][k]: scale[n] * W[n][tap k] for k < 25, zero for the padding taps
__global__ void conv1_pack_kernel(const float* __restrict__ w, const float* __restrict__ scale, u16* __restrict__ w1b) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 16 * 32) return;
  const int k = i & 31, n = i >> 5;
  w1b[i] = k < 25 ? to_bf16(w[n * 25 + k] * scale[n]) : (u16)0;
}

// ------------------------------------------------------------------------------------------------
// conv2 (model.py:53-57: 16 -> 32 channels, 5x5, stride 2) as an LDS-window kernel.  In the GEMM form each input element
// travels 25/4 times through L2 and a K-tile is one tap of 32 channels, half of them the zero-weighted decoder half of the
// interleaved level-1 buffer: 185 us at 216 tiles for 57 us of HBM traffic.  Here a block owns 8 x 16 OUTPUT pixels,
// stages their (2*8+3) x (2*16+3) input window (the 16 skip channels only) once, split by column parity so that the 16
// pixels of an operand fragment -- stride 2 in the image -- are consecutive 32-byte LDS slots (conflict-free ds_read_b128),
// and walks 13 K-steps of two taps x 16 channels.  The 26 KB of weights stay in LDS while the block walks its tiles
// (persistent blocks).  Operand roles are swapped (weights = first MFMA operand): the four accumulator registers of a
// lane are then four consecutive CHANNELS of one pixel, which leave as one 8-byte store.
// ------------------------------------------------------------------------------------------------
struct Conv2WinArgs {
  const u16* x; long ldx; int B, H, W;       // skip half of level 1 (16 channels at x, pixel pitch ldx)
  const u16* wk;                             // [13 steps][32 n][32 k = (tap parity, channel)] bf16, scale folded
  const float* shift; float slope;
  u16* y; long ldy; int Ho, Wo;              // 32 channels at y
};
__global__ __launch_bounds__(256) void conv2_window_bf16_kernel(Conv2WinArgs p) {
  constexpr int TH = 8, TW = 16, WR = 2 * TH + 3, WC = 2 * TW + 3, PW = TW + 2;     // window rows / columns; slots per parity plane
  constexpr int NCH = WR * WC * 2, NST = (NCH + 255) / 256;                         // 16-byte pieces of the window
  constexpr int NWP = 13 * 32 * 4, NWL = (NWP + 255) / 256;                         // 16-byte pieces of the weights
  __shared__ __attribute__((aligned(16))) unsigned char win[WR * 2 * PW * 32];
  __shared__ __attribute__((aligned(16))) unsigned char wts[13 * 32 * 64];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, lrow = lane & 15, q = lane >> 4;
  const int tiles_w = (p.Wo + TW - 1) / TW, tiles_h = (p.Ho + TH - 1) / TH;
  const int ntiles = p.B * tiles_h * tiles_w;
#pragma unroll
  for (int k = 0; k < NWL; ++k) {                                                  // weights: row n of step s, piece swizzled by n
    const int e = t + k * 256;
    if (e < NWP) {
      const int piece = e & 3, n = (e >> 2) & 31, s_ = e >> 7;
      *(uint4*)(&wts[(s_ * 32 + n) * 64 + ((piece ^ ((n >> 1) & 3)) * 16)]) = *(const uint4*)(p.wk + (long)e * 8);
    }
  }
  const int t2 = q >> 1, half = q & 1;
  float sh[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) sh[j][r] = p.shift[j * 16 + q * 4 + r];
  uint4 stage[NST];
  auto fetch_window = [&](int tile) __attribute__((always_inline)) {              // global -> registers (in flight during the previous tile's MFMAs)
    const int fow0 = (tile % tiles_w) * TW, foh0 = ((tile / tiles_w) % tiles_h) * TH;
    const u16* const xb = p.x + (long)(tile / (tiles_w * tiles_h)) * p.H * p.W * p.ldx;
#pragma unroll
    for (int k = 0; k < NST; ++k) {
      const int e = t + k * 256, hf = e & 1, px = e >> 1;
      const int wr = px / WC, wc = px - wr * WC;
      const int ih = 2 * foh0 - 2 + wr, iw = 2 * fow0 - 2 + wc;
      stage[k] = make_uint4(0u, 0u, 0u, 0u);
      if (e < NCH && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W) stage[k] = *(const uint4*)(xb + ((long)ih * p.W + iw) * p.ldx + hf * 8);
    }
  };
  if ((int)blockIdx.x < ntiles) fetch_window(blockIdx.x);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int ow0 = (tile % tiles_w) * TW, oh0 = ((tile / tiles_w) % tiles_h) * TH;
    const long b = tile / (tiles_w * tiles_h);
    __syncthreads();                                                               // the previous tile's readers are done
#pragma unroll
    for (int k = 0; k < NST; ++k) {
      const int e = t + k * 256, hf = e & 1, px = e >> 1;
      const int wr = px / WC, wc = px - wr * WC;
      if (e < NCH) *(uint4*)(&win[((wr * 2 + (wc & 1)) * PW + (wc >> 1)) * 32 + hf * 16]) = stage[k];
    }
    if (tile + (int)gridDim.x < ntiles) fetch_window(tile + gridDim.x);
    __syncthreads();
    f32x4 acc[2][2];                                                               // [pixel row i][channel tile j]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // pixel fragment of output row 2*wave + i: lane (pixel lrow, q) reads channels half*8.. of tap 2s + t2
    const unsigned char* const pbase = &win[(((4 * wave) * 2) * PW + lrow) * 32 + half * 16];
    const unsigned char* const wbase = &wts[lrow * 64 + ((q ^ ((lrow >> 1) & 3)) * 16)];
    bf_static_for<13>([&](auto sc) {
      constexpr int s_ = decltype(sc)::value;
      constexpr int tap0 = 2 * s_, tap1 = (2 * s_ + 1 < 25) ? 2 * s_ + 1 : 24;     // (the 26th tap has zero weights: any finite data)
      constexpr int off0 = (((tap0 / 5) * 2 + ((tap0 % 5) & 1)) * PW + ((tap0 % 5) >> 1)) * 32;
      constexpr int off1 = (((tap1 / 5) * 2 + ((tap1 % 5) & 1)) * PW + ((tap1 % 5) >> 1)) * 32;
      const int off = t2 ? off1 : off0;
      bf16x8 fw[2], fp[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) fw[j] = __builtin_bit_cast(bf16x8, *(const f32x4*)(wbase + (s_ * 32 + j * 16) * 64));
#pragma unroll
      for (int i = 0; i < 2; ++i) fp[i] = __builtin_bit_cast(bf16x8, *(const f32x4*)(pbase + off + i * (4 * PW * 32)));
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[j], fp[i], acc[i][j], 0, 0, 0);
    });
    // D map: column lrow = pixel, rows q*4 + r = channels (+16 j): 4 consecutive channels per lane -> one 8-byte store
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int oh = oh0 + 2 * wave + i, ow = ow0 + lrow;
      if (oh >= p.Ho || ow >= p.Wo) continue;
      u16* const dst = p.y + ((b * p.Ho + oh) * p.Wo + ow) * p.ldy + q * 4;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float u = acc[i][j][r] + sh[j][r]; v[r] = u > 0.f ? u : u * p.slope; }
        uint2 o;
        o.x = (unsigned)to_bf16(v[0]) | ((unsigned)to_bf16(v[1]) << 16);
        o.y = (unsigned)to_bf16(v[2]) | ((unsigned)to_bf16(v[3]) << 16);
        *(uint2*)(dst + j * 16) = o;
      }
    }
  }
}
// conv3 (32 -> 64 channels, stride-2 gather) in the same LDS-window form.  In the GEMM form every input pixel goes through L2 /
// LDS 6.25 times (4.6x the algorithmic HBM bytes by the counters, 89 us at 216 tiles); here a block owns 8 x 16 output pixels,
// stages their 19 x 35 input window ONCE (even / odd columns in separate planes so that the 16 pixels of a fragment, two input
// columns apart, are adjacent slots; 80-byte pixel pitch: conflict-free ds_read_b128), keeps ALL 25 x 64 x 32 weights in LDS for
// its whole life (100 KB, 16-byte pieces XOR-swizzled by the row; gather-packed bf16 weights [n][tap][c] as the GEMM has them),
// and every tap is eight MFMAs per wave (2 pixel rows x 4 channel tiles, K = the tap's 32 channels) with both fragments at
// base + compile-time offset: no barrier and no global access inside a tile.  157 KB of LDS: one persistent block per CU, the next
// tile's window in flight (registers) during the current tile's MFMAs.
struct Conv3WinArgs {
  const u16* x; long ldx; int B, H, W;       // 32 channels at x
  const u16* wg;                             // [64][25][32] bf16, BatchNorm scale folded in
  const float* shift; float slope;
  u16* y; long ldy; int Ho, Wo;              // 64 channels at y
};
#define CONV3_WIN_LDS (19 * 2 * 18 * 80 + 25 * 64 * 64)
__global__ __launch_bounds__(256) void conv3_window_bf16_kernel(Conv3WinArgs p) {
  constexpr int TH = 8, TW = 16, WR = 2 * TH + 3, WC = 2 * TW + 3, PW = TW + 2, LPB = 80;
  constexpr int NCH = WR * WC * 4, NST = (NCH + 255) / 256;                       // 16-byte pieces of the window
  constexpr int WIN_BYTES = WR * 2 * PW * LPB;
  static_assert(WIN_BYTES + 25 * 64 * 64 == CONV3_WIN_LDS, "LDS size");
  extern __shared__ __attribute__((aligned(16))) unsigned char c3_smem[];
  unsigned char* const win = c3_smem;
  unsigned char* const wts = c3_smem + WIN_BYTES;                                 // [tap][n][64 bytes], piece ^ ((n >> 1) & 3)
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, lrow = lane & 15, q = lane >> 4;
  const int tiles_w = (p.Wo + TW - 1) / TW, tiles_h = (p.Ho + TH - 1) / TH;
  const int ntiles = p.B * tiles_h * tiles_w;
  for (int e = t; e < 25 * 64 * 4; e += 256) {                                    // global [n][tap][4 pieces]: contiguous
    const int piece = e & 3, tap = (e >> 2) % 25, n = e / 100;
    *(uint4*)(wts + (tap * 64 + n) * 64 + ((piece ^ ((n >> 1) & 3)) * 16)) = *(const uint4*)(p.wg + (long)e * 8);
  }
  float sh[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) sh[j][r] = p.shift[j * 16 + q * 4 + r];
  uint4 stage[NST];
  auto fetch_window = [&](int tile) __attribute__((always_inline)) {
    const int fow0 = (tile % tiles_w) * TW, foh0 = ((tile / tiles_w) % tiles_h) * TH;
    const u16* const xb = p.x + (long)(tile / (tiles_w * tiles_h)) * p.H * p.W * p.ldx;
#pragma unroll
    for (int k = 0; k < NST; ++k) {
      const int e = t + k * 256, c4 = e & 3, px = e >> 2;
      const int wr = px / WC, wc = px - wr * WC;
      const int ih = 2 * foh0 - 2 + wr, iw = 2 * fow0 - 2 + wc;
      stage[k] = make_uint4(0u, 0u, 0u, 0u);
      if (e < NCH && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W) stage[k] = *(const uint4*)(xb + ((long)ih * p.W + iw) * p.ldx + c4 * 8);
    }
  };
  if ((int)blockIdx.x < ntiles) fetch_window(blockIdx.x);
  const unsigned char* const pbase = win + ((8 * wave) * PW + lrow) * LPB + q * 16;       // window row 4 wave (+ 2 i + kh), plane 0, slot lrow
  const unsigned char* const wbase = wts + lrow * 64 + ((q ^ ((lrow >> 1) & 3)) * 16);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int ow0 = (tile % tiles_w) * TW, oh0 = ((tile / tiles_w) % tiles_h) * TH;
    const long b = tile / (tiles_w * tiles_h);
    __syncthreads();                                                               // the previous tile's readers are done (first tile: the weights are staged)
#pragma unroll
    for (int k = 0; k < NST; ++k) {
      const int e = t + k * 256, c4 = e & 3, px = e >> 2;
      const int wr = px / WC, wc = px - wr * WC;
      if (e < NCH) *(uint4*)(win + ((wr * 2 + (wc & 1)) * PW + (wc >> 1)) * LPB + c4 * 16) = stage[k];
    }
    if (tile + (int)gridDim.x < ntiles) fetch_window(tile + gridDim.x);
    __syncthreads();
    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf_static_for<25>([&](auto tc) {
      constexpr int tap = decltype(tc)::value, kh = tap / 5, kw = tap % 5;
      bf16x8 fw[4], fp[2];
#pragma unroll
      for (int j = 0; j < 4; ++j) fw[j] = __builtin_bit_cast(bf16x8, *(const f32x4*)(wbase + (tap * 64 + j * 16) * 64));
#pragma unroll
      for (int i = 0; i < 2; ++i) fp[i] = __builtin_bit_cast(bf16x8, *(const f32x4*)(pbase + (((2 * i + kh) * 2 + (kw & 1)) * PW + (kw >> 1)) * LPB));
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[j], fp[i], acc[i][j], 0, 0, 0);
    });
    // D map: column lrow = pixel, rows q*4 + r = channels (+16 j): 4 consecutive channels per lane -> one 8-byte store
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int oh = oh0 + 2 * wave + i, ow = ow0 + lrow;
      if (oh >= p.Ho || ow >= p.Wo) continue;
      u16* const dst = p.y + ((b * p.Ho + oh) * p.Wo + ow) * p.ldy + q * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float u = acc[i][j][r] + sh[j][r]; v[r] = u > 0.f ? u : u * p.slope; }
        uint2 o;
        o.x = (unsigned)to_bf16(v[0]) | ((unsigned)to_bf16(v[1]) << 16);
        o.y = (unsigned)to_bf16(v[2]) | ((unsigned)to_bf16(v[3]) << 16);
        *(uint2*)(dst + j * 16) = o;
      }
    }
  }
}

// wk[s][n][k]: k = t2*16 + c holds scale[n] * W[n][tap = 2s + t2][c] (zero for the 26th tap), from the gather-packed fp32 weights [n][25][16]
__global__ void conv2_pack_kernel(const float* __restrict__ wp, const float* __restrict__ scale, u16* __restrict__ wk) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 13 * 32 * 32) return;
  const int k = i & 31, n = (i >> 5) & 31, s_ = i >> 10;
  const int tap = 2 * s_ + (k >> 4), c = k & 15;
  wk[i] = tap < 25 ? to_bf16(wp[(n * 25 + tap) * 16 + c] * scale[n]) : (u16)0;
}

// deconv6 + sigmoid (model.py:109,198-200): 32 bf16 channels -> 1 fp32 channel, on the MFMA.  One output channel is no GEMM
// column, but the four output parities of an anchor pixel are: out(2a + ph, 2c + pw) = sum over the 3x3 input neighbourhood of
// (a, c) and 32 channels of x * W2[pos][ph, pw][ch], where W2 holds tap (ph + 2 - 2 dh, pw + 2 - 2 dw) or zero if that tap does
// not exist (9 / 6 / 6 / 4 taps per parity).  So M = anchor pixels, K = 9 x 32, N = 4 (padded to the MFMA's 16 columns; the
// padding costs nothing that matters: 72 MFMAs per 8x16-anchor block).  A block stages its (8+2) x (16+2) x 32 window once in
// LDS (96-byte pixel pitch: the 16 pixels x 16 bytes of a ds_read_b128 group then fall on 64 distinct banks); every operand
// fragment is one ds_read_b128.  Lanes of the columns 0..3 exchange with their pw-partner so that each writes one float4
// of an output row.  Bound: HBM (32 ch x 2 B in, 4 x 4 B out per anchor).
struct Deconv6Args {
  const u16* x; long ldx; int B, H, W; const u16* w2; const float* bias; float* y; int Ho, Wo;
  long plane;        // elements from channel 0 to channel 16 of a pixel: 16 in an interleaved view, the plane distance in the planar level 1
};
__global__ __launch_bounds__(256) void deconv6_mfma_bf16_kernel(Deconv6Args p) {
  constexpr int TH = 8, TW = 16, WW = TW + 2, NPX = (TH + 2) * WW, PS = 48;     // PS: u16 per staged pixel
  __shared__ __attribute__((aligned(16))) u16 win[NPX * PS];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, lrow = lane & 15, q = lane >> 4;
  const int tiles_w = (p.W + TW - 1) / TW, tiles_h = (p.H + TH - 1) / TH;
  const int ntiles = p.B * tiles_h * tiles_w;
  constexpr int NST = (NPX * 4 + 255) / 256;
  // persistent blocks: the nine weight fragments (36 KB per block from L2 -- twice the window) are fetched once, and the next
  // tile's window is requested before the current tile's MFMAs
  bf16x8 fb[9];
#pragma unroll
  for (int pos = 0; pos < 9; ++pos) fb[pos] = *(const bf16x8*)(p.w2 + (pos * 16 + lrow) * 32 + q * 8);
  const float bias = p.bias[0];
  // Window staging with as few VALU instructions per tile as possible (the kernel is VALU-bound: 423 per tile and wave before, of
  // which the index arithmetic of the 3 staged pieces per thread was ~100): a thread's pieces are the same window positions in
  // every tile, so (row, column, byte offset relative to the tile's first window pixel, LDS address) are computed ONCE; per tile
  // only the bounds test remains, and the tile's position is a scalar offset of a buffer load (out of the image: offset past
  // num_records, which reads as zeros).
  constexpr unsigned OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x - ((long)p.W + 1) * p.ldx), 0, OOB, 0x00020000);
  int s_lh[NST], s_lw[NST]; unsigned s_rel[NST]; u16* s_dst[NST];
#pragma unroll
  for (int k = 0; k < NST; ++k) {
    const int e = t + k * 256, px = e >> 2, cq = e & 3;
    s_lh[k] = px / WW; s_lw[k] = px - s_lh[k] * WW;
    if (px >= NPX) s_lh[k] = -(1 << 20);                          // never inside the image
    s_rel[k] = (unsigned)((((long)s_lh[k] * p.W + s_lw[k]) * p.ldx + (cq & 1) * 8 + (cq >> 1) * p.plane) * 2);
    s_dst[k] = &win[(px < NPX ? px : 0) * PS + cq * 8];
  }
  uint4 stage[NST];
  auto fetch_window = [&](int tile) __attribute__((always_inline)) {
    const int ftw0 = (tile % tiles_w) * TW, fth0 = ((tile / tiles_w) % tiles_h) * TH;
    const int fb_ = tile / (tiles_w * tiles_h);
    const int soff = __builtin_amdgcn_readfirstlane((int)((((long)fb_ * p.H + fth0) * p.W + ftw0) * p.ldx * 2));
#pragma unroll
    for (int k = 0; k < NST; ++k) {
      const bool ok = (unsigned)(fth0 - 1 + s_lh[k]) < (unsigned)p.H && (unsigned)(ftw0 - 1 + s_lw[k]) < (unsigned)p.W;
      stage[k] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(ok ? s_rel[k] : OOB), soff, 0));
    }
  };
  if ((int)blockIdx.x < ntiles) fetch_window(blockIdx.x);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int tw0 = (tile % tiles_w) * TW, th0 = ((tile / tiles_w) % tiles_h) * TH;
    const long b = tile / (tiles_w * tiles_h);
    __syncthreads();                                               // the previous tile's readers are done
#pragma unroll
    for (int k = 0; k < NST; ++k)
      if (t + k * 256 < NPX * 4) *(uint4*)s_dst[k] = stage[k];
    if (tile + (int)gridDim.x < ntiles) fetch_window(tile + gridDim.x);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ah = 2 * wave + i;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int pos = 0; pos < 9; ++pos) {
        const int dh = pos / 3, dw = pos % 3;                      // window coordinates of (anchor - 1 + d)
        const bf16x8 fa = *(const bf16x8*)(&win[((ah + dh) * WW + lrow + dw) * PS + q * 8]);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb[pos], acc, 0, 0, 0);
      }
      // C map: column lrow (= parity for lrow < 4), rows q*4 + r = anchors tw0 + q*4 + r.  Only the columns 0..3 carry outputs:
      // register r of lane (q, c < 4) moves to lane (q, c + 4 r) (DPP row_shr inside the 16-lane row, written through a bank
      // mask), so that every lane holds ONE output -- anchor q*4 + (lrow >> 2), parity lrow & 3 -- and evaluates one sigmoid
      // instead of four; a row of 32 outputs (16 anchors x 2 column parities) then leaves as 4-byte stores of 32 adjacent lanes.
      // (the four registers go through opaque asm copies first: hipcc of ROCm 7.2 otherwise reads accumulator register 0 for all
      // four DPP sources -- v_accvgpr_read a0 twice, no a1..a3 -- and the outputs are wrong)
      float v = acc[0], a1 = acc[1], a2 = acc[2], a3 = acc[3];
      asm volatile("" : "+v"(v), "+v"(a1), "+v"(a2), "+v"(a3));
      v = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, a1), 0x114, 0xF, 0x2, false));   // row_shr:4  -> lanes 4..7
      v = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, a2), 0x118, 0xF, 0x4, false));   // row_shr:8  -> lanes 8..11
      v = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, a3), 0x11C, 0xF, 0x8, false));   // row_shr:12 -> lanes 12..15
      v = 1.f / (1.f + __expf(-(v + bias)));
      const int par = lrow & 3, an = q * 4 + (lrow >> 2);
      const int oh = 2 * (th0 + ah) + (par >> 1), ow = 2 * (tw0 + an) + (par & 1);
      if (th0 + ah < p.H && tw0 + an < p.W && oh < p.Ho && ow < p.Wo) p.y[(b * p.Ho + oh) * p.Wo + ow] = v;
    }
  }
}
// W2[pos = (dh+1)*3 + (dw+1)][n = ph*2 + pw (rows 4..15 zero)][ch] from the torch-layout fp32 weights [ch][kh*5 + kw]
__global__ void deconv6_pack_kernel(const float* __restrict__ w, u16* __restrict__ w2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 9 * 16 * 32) return;
  const int c = i & 31, n = (i >> 5) & 15, pos = i >> 9;
  float v = 0.f;
  if (n < 4) {
    const int kh = (n >> 1) + 2 - 2 * (pos / 3 - 1), kw = (n & 1) + 2 - 2 * (pos % 3 - 1);
    if (kh >= 0 && kh <= 4 && kw >= 0 && kw <= 4) v = w[c * 25 + kh * 5 + kw];
  }
  w2[i] = to_bf16(v);
}

// packed fp32 weights (gather [n][25][c] or parity [class][n][taps][c]) times the folded BatchNorm scale of output channel n
// -> bf16.  cpad > c: the channels are placed at [cpad - c, cpad) of a cpad-wide K row and the rest is zero (conv2 reads the
// interleaved 32-channel level 1 with zero weights on the decoder half).
__global__ __launch_bounds__(256) void pack_bf16_kernel(const float* __restrict__ wp, const float* __restrict__ scale, u16* __restrict__ out,
                                                        int N, int C, int cpad, int parity) {
  const long total = (long)N * cpad * 25;
  for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < total; o += (long)gridDim.x * 256) {
    const int c = (int)(o % cpad);
    const long r = o / cpad;                                  // (class, n, tap) flattened as in the fp32 packing
    int n;
    if (!parity) n = (int)(r / 25);
    else {
      const long NC = N;                                      // rows per class = N * ntaps
      long rr = r; int ntaps = 9;
      if (rr >= 9 * NC) { rr -= 9 * NC; ntaps = 6; if (rr >= 6 * NC) { rr -= 6 * NC; if (rr >= 6 * NC) { rr -= 6 * NC; ntaps = 4; } } }
      n = (int)(rr / ntaps);
    }
    const int cs = c - (cpad - C);
    out[o] = cs >= 0 ? to_bf16(wp[r * C + cs] * scale[n]) : (u16)0;
  }
}

// ---- host side -----------------------------------------------------------------------------------
struct Bf16Plan { int BM, BN, cfg, ksplit, grid_y; long mtiles; };
static Bf16Plan plan_bf16(int mode, long Mmax, int N, int nkt) {
  Bf16Plan pl{};
  if (N % 128 == 0) { pl.cfg = Mmax >= 4096 ? 0 : 4; pl.BM = Mmax >= 4096 ? 128 : 64; pl.BN = 128; }
  else if (N == 64) { pl.cfg = 1; pl.BM = 128; pl.BN = 64; }
  else if (N == 32) { pl.cfg = 2; pl.BM = 256; pl.BN = 32; }
  else { pl.cfg = 3; pl.BM = 256; pl.BN = 16; }
  pl.mtiles = (Mmax + pl.BM - 1) / pl.BM;
  pl.grid_y = mode == BF_PARITY ? 4 : 1;
  const long blocks = pl.mtiles * (N / pl.BN) * pl.grid_y;
  int ks = 1;
  if (blocks < 512) {                                        // deep levels at small batch: split K to fill the chip
    ks = (int)((768 + blocks - 1) / blocks);
    const int cap = nkt / 8 > 1 ? nkt / 8 : 1;
    if (ks > cap) ks = cap;
    if (ks > 32) ks = 32;
  }
  if (svs_tune_on(SVS_TUNE_BF16_CFG)) {                       // sweeps: 0 128x128, 1 128x64, 4 64x128 (where N allows)
    const int c = (int)svs_tune(SVS_TUNE_BF16_CFG);
    if (c == 0 && N % 128 == 0) { pl.cfg = 0; pl.BM = 128; pl.BN = 128; }
    if (c == 1 && N % 64 == 0) { pl.cfg = 1; pl.BM = 128; pl.BN = 64; }
    if (c == 4 && N % 128 == 0) { pl.cfg = 4; pl.BM = 64; pl.BN = 128; }
    pl.mtiles = (Mmax + pl.BM - 1) / pl.BM;
    const long blocks2 = pl.mtiles * (N / pl.BN) * pl.grid_y;
    ks = 1;
    if (blocks2 < 512) { ks = (int)((768 + blocks2 - 1) / blocks2); const int cap = nkt / 8 > 1 ? nkt / 8 : 1; if (ks > cap) ks = cap; if (ks > 32) ks = 32; }
  }
  if (svs_tune_on(SVS_TUNE_BF16_KSPLIT)) { const int f = (int)svs_tune(SVS_TUNE_BF16_KSPLIT); if (f >= 1 && f <= 32 && f <= nkt) ks = f; }   // sweeps
  if (svs_tune_on(SVS_TUNE_BF16_CFG)) {                       // sweeps: 0 128x128, 1 128x64, 4 64x128 (where N allows)
    const int c = (int)svs_tune(SVS_TUNE_BF16_CFG);
    if (c == 0 && N % 128 == 0) { pl.cfg = 0; pl.BM = 128; pl.BN = 128; }
    if (c == 1 && N % 64 == 0) { pl.cfg = 1; pl.BM = 128; pl.BN = 64; }
    if (c == 4 && N % 128 == 0) { pl.cfg = 4; pl.BM = 64; pl.BN = 128; }
    pl.mtiles = (Mmax + pl.BM - 1) / pl.BM;
    const long blocks2 = pl.mtiles * (N / pl.BN) * pl.grid_y;
    ks = 1;
    if (blocks2 < 512) { ks = (int)((768 + blocks2 - 1) / blocks2); const int cap = nkt / 8 > 1 ? nkt / 8 : 1; if (ks > cap) ks = cap; if (ks > 32) ks = 32; }
  }
  if (svs_tune_on(SVS_TUNE_BF16_KSPLIT)) { const int f = (int)svs_tune(SVS_TUNE_BF16_KSPLIT); if (f >= 1 && f <= 32 && f <= nkt) ks = f; }   // sweeps
  pl.ksplit = ks;
  return pl;
}
static size_t bf16_layer_ws(int mode, int B, int H, int W, int C, int Ho, int Wo, int N) {
  const long Mmax = mode == BF_GATHER ? (long)B * Ho * Wo : (long)B * ((Ho + 1) / 2) * ((Wo + 1) / 2);
  const int nkt = (mode == BF_GATHER ? 25 : 4) * (C / 32);
  const Bf16Plan pl = plan_bf16(mode, Mmax, N, nkt);
  return pl.ksplit > 1 ? (size_t)pl.ksplit * B * Ho * Wo * N * sizeof(float) : 0;
}
static int conv_bf16_run(int mode, const u16* x, long ldx, int B, int H, int W, int C, const u16* wp, const float* shift, float slope,
                         u16* y, long ldy, int Ho, int Wo, int N, void* ws, size_t ws_bytes, hipStream_t stream) {
  SVS_REQUIRE(C % 32 == 0 && (N == 16 || N == 32 || N == 64 || N % 128 == 0), "conv_bf16: unsupported channels C=%d N=%d", C, N);
  SVS_REQUIRE(((long)B * H * W * ldx + 4L * (W + 2) * ldx) * 2 < (1L << 31) && (long)N * C * 25 * 2 < (1L << 31), "conv_bf16: view too large; split the batch");
  const long Mmax = mode == BF_GATHER ? (long)B * Ho * Wo : (long)B * ((Ho + 1) / 2) * ((Wo + 1) / 2);
  const int nkt = (mode == BF_GATHER ? 25 : 4) * (C / 32);
  const Bf16Plan pl = plan_bf16(mode, Mmax, N, nkt);
  ConvBf16Args a{x, ldx, B, H, W, C, wp, shift, slope, y, ldy, Ho, Wo, N, pl.ksplit, nullptr};
  // shallow decoder layers: LDS-window kernel (whenever its tiles fill the chip)
  const long wtiles = (long)B * ((H + 7) / 8) * ((W + 15) / 16);
  // LDS of parity_window_bf16_kernel<C, N / 16, TW>: window + weight buffers (all 25 taps when they fit 60 KB unpadded, else two)
  auto pw_lds = [](int C_, int N_, int TW_) -> size_t {
    const size_t win = (size_t)(128 / TW_ + 2) * (TW_ + 2) * (C_ * 2 + 16);
    const bool wall = 25L * N_ * (C_ * 2 + 16) <= 60 * 1024;
    return win + (wall ? (size_t)25 * N_ * C_ * 2 : (size_t)2 * N_ * (C_ * 2 + 16));
  };
  if (mode == BF_PARITY && ((C == 128 && N == 32) || (C == 64 && N == 16)) && H >= 8 && W >= 16 && wtiles >= 128 &&
      (long)H * W * ldx * 2 < (1L << 31) && svs_tune(SVS_TUNE_CONV_WINDOW) != 0) {
    a.ksplit = 1;
    const unsigned wgrid = (unsigned)(wtiles < 512 ? wtiles : 512);      // persistent: two blocks per CU
    const size_t lds = pw_lds(C, N, 16);
    if (C == 128) {
      SVS_HIP(hipFuncSetAttribute((const void*)parity_window_bf16_kernel<128, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL((parity_window_bf16_kernel<128, 2>), dim3(wgrid), dim3(256), lds, stream, a);
    } else {
      SVS_HIP(hipFuncSetAttribute((const void*)parity_window_bf16_kernel<64, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL((parity_window_bf16_kernel<64, 1>), dim3(wgrid), dim3(256), lds, stream, a);
    }
    SVS_CHECK_LAUNCH("parity_window_bf16");
    return SVS_OK;
  }
  // deconv3 (256 -> 64 channels on 32 x 8 anchors): the same kernel on 16 x 8 tiles.  In the GEMM form 1 GB goes from L2 to LDS
  // per launch at 216 tiles (every anchor's 512 bytes once per tap, the 128 x 64 tile re-reading them for 64 channels only):
  // 106 us; here the window is staged once and only the weights (819 KB per tile) stream.  SVS_BF16_DECONV3_WINDOW=0: GEMM form.
  const long wtiles8 = (long)B * ((H + 15) / 16) * ((W + 7) / 8);
  if (mode == BF_PARITY && C == 256 && N == 64 && W <= 8 && wtiles8 >= 128 && (long)H * W * ldx * 2 < (1L << 31) &&
      svs_tune(SVS_TUNE_CONV_WINDOW) != 0 && svs_tune(SVS_TUNE_BF16_DECONV3_WINDOW) != 0) {
    a.ksplit = 1;
    const size_t lds = pw_lds(256, 64, 8);
    SVS_HIP(hipFuncSetAttribute((const void*)parity_window_bf16_kernel<256, 4, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((parity_window_bf16_kernel<256, 4, 8>), dim3((unsigned)(wtiles8 < 256 ? wtiles8 : 256)), dim3(256), lds, stream, a);
    SVS_CHECK_LAUNCH("parity_window_bf16");
    return SVS_OK;
  }
  if (pl.ksplit > 1) {
    const size_t need = (size_t)pl.ksplit * B * Ho * Wo * N * sizeof(float);
    if (!ws || ws_bytes < need) { svs_set_error("conv_bf16: workspace too small (%zu < %zu)", ws_bytes, need); return SVS_ERR_WORKSPACE; }
    a.slab = (float*)ws;
  }
  dim3 grid((unsigned)(pl.mtiles * (N / pl.BN)), (unsigned)pl.ksplit, (unsigned)pl.grid_y);
#define SVS_BF16_LAUNCH_KB(MODE_) \
  switch (pl.cfg) { \
    case 0: hipLaunchKernelGGL((conv_gemm_bf16_kernel<MODE_, 128, 128, 2, 2, KB_>), grid, dim3(256), 0, stream, a); break; \
    case 1: hipLaunchKernelGGL((conv_gemm_bf16_kernel<MODE_, 128, 64, 2, 2, KB_>), grid, dim3(256), 0, stream, a); break; \
    case 2: hipLaunchKernelGGL((conv_gemm_bf16_kernel<MODE_, 256, 32, 4, 1, 1>), grid, dim3(256), 0, stream, a); break; \
    case 3: hipLaunchKernelGGL((conv_gemm_bf16_kernel<MODE_, 256, 16, 4, 1, 1>), grid, dim3(256), 0, stream, a); break; \
    default: hipLaunchKernelGGL((conv_gemm_bf16_kernel<MODE_, 64, 128, 2, 2, KB_>), grid, dim3(256), 0, stream, a); break; \
  }
  // K-tiles per barrier: 2 by default; BF16_KB = 1 / 2 / 4 for sweeps
  const int kb = svs_tune_on(SVS_TUNE_BF16_KB) ? (int)svs_tune(SVS_TUNE_BF16_KB) : 2;
#define SVS_BF16_LAUNCH(MODE_) \
  if (kb >= 4) { constexpr int KB_ = 4; SVS_BF16_LAUNCH_KB(MODE_) } else if (kb == 2) { constexpr int KB_ = 2; SVS_BF16_LAUNCH_KB(MODE_) } \
  else { constexpr int KB_ = 1; SVS_BF16_LAUNCH_KB(MODE_) }
  if (mode == BF_GATHER) { SVS_BF16_LAUNCH(BF_GATHER) } else { SVS_BF16_LAUNCH(BF_PARITY) }
#undef SVS_BF16_LAUNCH_KB
#undef SVS_BF16_LAUNCH
  SVS_CHECK_LAUNCH("conv_gemm_bf16");
  if (pl.ksplit > 1) {
    const long P = (long)B * Ho * Wo, total4 = P * N / 4;
    int g = (int)((total4 + 255) / 256);
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(splitk_epilogue_bf16_kernel, dim3(g), dim3(256), 0, stream, (const float*)a.slab, pl.ksplit, P, N, shift, slope, y, ldy);
    SVS_CHECK_LAUNCH("splitk_epilogue_bf16");
  }
  return SVS_OK;
}

// ---- whole network ---------------------------------------------------------------------------------
static const int BCH[7] = {1, 16, 32, 64, 128, 256, 512};
static const int BDEC_C[6] = {512, 512, 256, 128, 64, 32};
static const int BDEC_N[6] = {256, 128, 64, 32, 16, 1};
struct Bf16Prepared { long w[12], shift[11], w1, w2k, w6, bias6, total; };          // byte offsets into the prepared blob
static Bf16Prepared bf16_prepared_layout() {
  Bf16Prepared L{};
  long off = 0;
  auto take = [&](long bytes) { long o = off; off += (bytes + 255) / 256 * 256; return o; };
  L.w1 = take(16 * 32 * 2);                                    // conv1: bf16 [16 n][32 taps (25 used)], scale folded
  for (int k = 3; k <= 6; ++k) L.w[k - 1] = take((long)BCH[k] * BCH[k - 1] * 25 * 2);      // (conv2 has its own form: w2k)
  for (int j = 0; j < 5; ++j) L.w[6 + j] = take((long)BDEC_C[j] * BDEC_N[j] * 25 * 2);
  L.w2k = take(13 * 32 * 32 * 2);                              // conv2, window kernel: bf16 [13 steps][32 n][2 taps x 16 channels]
  L.w6 = take(9 * 16 * 32 * 2);                                // deconv6: bf16 [9 positions][16 columns (4 parities used)][32 channels]
  for (int l = 0; l < 11; ++l) L.shift[l] = take((l < 6 ? BCH[l + 1] : BDEC_N[l - 6]) * 4);
  L.bias6 = take(4);
  L.total = off;
  return L;
}
extern "C" size_t svs_unet_prepared_bf16_bytes(void) { return (size_t)bf16_prepared_layout().total; }

// prepared_f32: the blob of svs_unet_prepare_eval (packed fp32 weights + folded scale / shift); this adds the bf16 forms
extern "C" int svs_unet_prepare_eval_bf16(const void* prepared_f32, void* prepared_bf16, hipStream_t stream) {
  SVS_REQUIRE(prepared_f32 && prepared_bf16 && svs_aligned16(prepared_bf16), "svs_unet_prepare_eval_bf16: bad pointers");
  const Bf16Prepared L = bf16_prepared_layout();
  char* out = (char*)prepared_bf16;
  const float* blob = (const float*)prepared_f32;
  long wp[12], scale[11], shift[11], bias6;
  svs_unet_prepared_offsets(wp, scale, shift, &bias6);
  hipLaunchKernelGGL(conv1_pack_kernel, dim3(2), dim3(256), 0, stream, blob + wp[0], blob + scale[0], (u16*)(out + L.w1));
  SVS_CHECK_LAUNCH("conv1_pack");
  for (int k = 3; k <= 6; ++k) {
    const int N = BCH[k], C = BCH[k - 1];
    hipLaunchKernelGGL(pack_bf16_kernel, dim3(512), dim3(256), 0, stream, blob + wp[k - 1], blob + scale[k - 1], (u16*)(out + L.w[k - 1]), N, C, C, 0);
    SVS_CHECK_LAUNCH("pack_bf16");
  }
  for (int j = 0; j < 5; ++j) {
    hipLaunchKernelGGL(pack_bf16_kernel, dim3(512), dim3(256), 0, stream, blob + wp[6 + j], blob + scale[6 + j], (u16*)(out + L.w[6 + j]), BDEC_N[j],
                       BDEC_C[j], BDEC_C[j], 1);
    SVS_CHECK_LAUNCH("pack_bf16");
  }
  hipLaunchKernelGGL(conv2_pack_kernel, dim3(52), dim3(256), 0, stream, blob + wp[1], blob + scale[1], (u16*)(out + L.w2k));
  SVS_CHECK_LAUNCH("conv2_pack");
  hipLaunchKernelGGL(deconv6_pack_kernel, dim3(18), dim3(256), 0, stream, blob + wp[11], (u16*)(out + L.w6));
  SVS_CHECK_LAUNCH("deconv6_pack");
  SVS_HIP(hipMemcpyAsync(out + L.bias6, blob + bias6, 4, hipMemcpyDeviceToDevice, stream));
  for (int l = 0; l < 11; ++l)
    SVS_HIP(hipMemcpyAsync(out + L.shift[l], blob + shift[l], (l < 6 ? BCH[l + 1] : BDEC_N[l - 6]) * 4, hipMemcpyDeviceToDevice, stream));
  return SVS_OK;
}

struct Bf16Ws { u16* cat[6]; u16* c6; void* scratch; size_t scratch_bytes; size_t total; int h[7], w[7]; long P[7]; };
static int bf16_ws_layout(int B, int H, int W, void* ws, Bf16Ws& e) {
  SVS_REQUIRE(B > 0 && H > 0 && W > 0, "bad tile geometry B=%d H=%d W=%d", B, H, W);
  e.h[0] = H; e.w[0] = W;
  for (int k = 1; k <= 6; ++k) { e.h[k] = svs_conv_out(e.h[k - 1]); e.w[k] = svs_conv_out(e.w[k - 1]); }
  for (int k = 0; k <= 6; ++k) e.P[k] = (long)B * e.h[k] * e.w[k];
  size_t used = 0;
  auto take = [&](size_t bytes) { void* p = ws ? (char*)ws + used : nullptr; used += svs_align_up(bytes, 256); return p; };
  for (int k = 1; k <= 5; ++k) e.cat[k] = (u16*)take((size_t)e.P[k] * 2 * BCH[k] * 2);
  e.c6 = (u16*)take((size_t)e.P[6] * 512 * 2);
  size_t sb = 0;
  for (int k = 3; k <= 6; ++k) { const size_t s = bf16_layer_ws(BF_GATHER, B, e.h[k - 1], e.w[k - 1], BCH[k - 1], e.h[k], e.w[k], BCH[k]); if (s > sb) sb = s; }
  for (int j = 0; j < 5; ++j) { const size_t s = bf16_layer_ws(BF_PARITY, B, e.h[6 - j], e.w[6 - j], BDEC_C[j], e.h[5 - j], e.w[5 - j], BDEC_N[j]); if (s > sb) sb = s; }
  e.scratch_bytes = sb;
  e.scratch = take(sb + 256);
  e.total = used;
  return SVS_OK;
}
extern "C" size_t svs_unet_eval_bf16_workspace_bytes(int B, int H, int W) {
  Bf16Ws e;
  if (bf16_ws_layout(B, H, W, nullptr, e)) return 0;
  return e.total;
}

extern "C" int svs_unet_forward_eval_bf16(const void* prepared_bf16, const float* mix, float* mask, int B, int H, int W, void* ws,
                                          size_t ws_bytes, hipStream_t stream) {
  Bf16Ws e;
  int rc = bf16_ws_layout(B, H, W, ws, e);
  if (rc) return rc;
  SVS_REQUIRE(prepared_bf16 && mix && mask && svs_aligned16(mix) && svs_aligned16(mask), "svs_unet_forward_eval_bf16: bad pointers");
  // conv1 / deconv6 address the input tiles and the two level-1 planes with 32-bit byte offsets (buffer loads)
  SVS_REQUIRE(((long)B * H * W + 4L * W) * 4 < (1L << 31) && (e.P[1] * 32 + 4L * e.w[1] * 16) * 2 < (1L << 31),
              "svs_unet_forward_eval_bf16: %d tiles of %dx%d need 64-bit offsets; split the batch", B, H, W);
  if (!ws || ws_bytes < e.total || !svs_aligned16(ws)) { svs_set_error("svs_unet_forward_eval_bf16: workspace too small (%zu < %zu)", ws_bytes, e.total); return SVS_ERR_WORKSPACE; }
  const Bf16Prepared L = bf16_prepared_layout();
  const char* blob = (const char*)prepared_bf16;
  auto SH = [&](int l) { return (const float*)(blob + L.shift[l]); };
  // encoder (model.py:176-181).  Levels 2..5 are one interleaved buffer [decoder half | skip half] each; conv_k writes the skip half
  // of level k and reads the skip half of level k - 1.  Level 1 is PLANAR (decoder plane, then skip plane, 16 channels = 32 bytes per
  // pixel each): interleaved, each producer wrote -- and conv2 read -- 32 bytes of every 64, i.e. half of every HBM burst
  // (conv1 72 us for 170 MB at 216 tiles).  conv1 and conv2 have kernels of their own (1 and 16 input channels)
  {
    Conv1Args c{mix, B, H, W, (const u16*)(blob + L.w1), SH(0), 0.2f, e.cat[1] + e.P[1] * 16, 16L, e.h[1], e.w[1]};
    const long tiles = (long)B * ((e.h[1] + 15) / 16) * ((e.w[1] + 31) / 32);
    hipLaunchKernelGGL(conv1_mfma_bf16_kernel, dim3((unsigned)(tiles < 2048 ? tiles : 2048)), dim3(256), 0, stream, c);      // persistent: 8 blocks per CU
    SVS_CHECK_LAUNCH("conv1_mfma_bf16");
  }
  {
    Conv2WinArgs c{e.cat[1] + e.P[1] * 16, 16L, B, e.h[1], e.w[1], (const u16*)(blob + L.w2k), SH(1), 0.2f, e.cat[2] + 32, 64L, e.h[2], e.w[2]};
    const long tiles = (long)B * ((e.h[2] + 7) / 8) * ((e.w[2] + 15) / 16);
    hipLaunchKernelGGL(conv2_window_bf16_kernel, dim3((unsigned)(tiles < 768 ? tiles : 768)), dim3(256), 0, stream, c);
    SVS_CHECK_LAUNCH("conv2_window_bf16");
  }
  for (int k = 3; k <= 6; ++k) {
    const u16* x = e.cat[k - 1] + BCH[k - 1];
    const long ldx = 2L * BCH[k - 1];
    u16* y = k == 6 ? e.c6 : e.cat[k] + BCH[k];
    const long ldy = k == 6 ? 512 : 2L * BCH[k];
    if (k == 3 && svs_tune(SVS_TUNE_BF16_CONV3_WINDOW) != 0) {           // LDS-window form (SVS_BF16_CONV3_WINDOW=0: the GEMM form)
      Conv3WinArgs c{x, ldx, B, e.h[2], e.w[2], (const u16*)(blob + L.w[2]), SH(2), 0.2f, y, ldy, e.h[3], e.w[3]};
      const long tiles = (long)B * ((e.h[3] + 7) / 8) * ((e.w[3] + 15) / 16);
      SVS_HIP(hipFuncSetAttribute((const void*)conv3_window_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, CONV3_WIN_LDS));
      hipLaunchKernelGGL(conv3_window_bf16_kernel, dim3((unsigned)(tiles < 256 ? tiles : 256)), dim3(256), CONV3_WIN_LDS, stream, c);
      SVS_CHECK_LAUNCH("conv3_window_bf16");
      continue;
    }
    if ((rc = conv_bf16_run(BF_GATHER, x, ldx, B, e.h[k - 1], e.w[k - 1], BCH[k - 1], (const u16*)(blob + L.w[k - 1]), SH(k - 1), 0.2f, y, ldy, e.h[k], e.w[k],
                            BCH[k], e.scratch, e.scratch_bytes, stream))) return rc;
  }
  // decoder (model.py:183-196); Dropout2d is the identity in eval
  for (int j = 0; j < 5; ++j) {
    const int lin = 6 - j, lout = 5 - j;
    const u16* x = j == 0 ? e.c6 : e.cat[lin];
    if ((rc = conv_bf16_run(BF_PARITY, x, BDEC_C[j], B, e.h[lin], e.w[lin], BDEC_C[j], (const u16*)(blob + L.w[6 + j]), SH(6 + j), 0.f, e.cat[lout],
                            lout == 1 ? 16L : 2L * BCH[lout], e.h[lout], e.w[lout], BDEC_N[j], e.scratch, e.scratch_bytes, stream))) return rc;
  }
  // deconv6 + sigmoid (model.py:198-200)
  {
    Deconv6Args d{e.cat[1], 16L, B, e.h[1], e.w[1], (const u16*)(blob + L.w6), (const float*)(blob + L.bias6), mask, H, W, e.P[1] * 16};
    const long tiles = (long)B * ((e.h[1] + 7) / 8) * ((e.w[1] + 15) / 16);
    hipLaunchKernelGGL(deconv6_mfma_bf16_kernel, dim3((unsigned)(tiles < 2048 ? tiles : 2048)), dim3(256), 0, stream, d);
    SVS_CHECK_LAUNCH("deconv6_mfma_bf16");
  }
  return SVS_OK;
}
