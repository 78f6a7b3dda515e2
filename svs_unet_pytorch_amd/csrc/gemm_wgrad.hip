// Weight gradients of the 5x5 / stride-2 layers as an MFMA GEMM over pixels (fp32, 16x16x4), gfx950.
//
//   dw[cs][cl][kh][kw] = sum_{b,i,j} S[b,i,j,cs] * L[b,2i-2+kh,2j-2+kw,cl]
//
// S is the image on the strided (small) grid, L the one that is read through the 5x5 window:
//   Conv2d          (model.py:48):  S = dy (N ch), L = x (C ch)   -> dw is (N,C,5,5)
//   ConvTranspose2d (model.py:79):  S = x (C ch),  L = dy (N ch)  -> dw is (C,N,5,5)
// so one kernel serves both.  GEMM view: M = cs, N = (tap, cl) with cl fastest (25*Cl columns),
// K = pixels (b,i,j).  Both operands are contiguous along M resp. N in NHWC, so K-tiles of 16 pixels
// are staged as As[k][m], Bs[k][n] (16-byte global loads / LDS stores) and the MFMA operands are
// fetched with conflict-free ds_read_b32 (rows padded by 4 floats).  K is split over gridDim.z; each
// split writes an fp32 slab and wgrad_reduce_kernel sums the slabs in a fixed order (bitwise
// reproducible, no atomics) while transposing into torch's (cs,cl,kh,kw) layout.
//
// Bound: MFMA; algorithmic FLOPs = 2 * Cs * 25*Cl * B*Hs*Ws.
#include <stdlib.h>

#include "internal.h"
#include "mfma_split.h"

struct WgradArgs {
  const float* s; long lds; int Hs, Ws, Cs;
  const float* l; long ldl; int Hl, Wl, Cl;
  int B;
  float* slab;           // [ksplit][Cs][25*Cl]
  long pix_per_split;    // multiple of 16
  int ws_shift, hs_shift; // log2 of Ws / Hs when they are powers of two, else -1
  int b_shift;            // SKIP kernels: log2(B)
};

template <int T> __device__ __forceinline__ void load_vec(const float* p, float (&f)[T]);
template <> __device__ __forceinline__ void load_vec<4>(const float* p, float (&f)[4]) {
  const f32x4 v = *(const f32x4*)p; f[0] = v[0]; f[1] = v[1]; f[2] = v[2]; f[3] = v[3];
}
template <> __device__ __forceinline__ void load_vec<2>(const float* p, float (&f)[2]) {
  const float2 v = *(const float2*)p; f[0] = v.x; f[1] = v.y;
}
template <int T> __device__ __forceinline__ void store_vec(float* p, const float (&f)[T]);
template <> __device__ __forceinline__ void store_vec<4>(float* p, const float (&f)[4]) { *(f32x4*)p = (f32x4){f[0], f[1], f[2], f[3]}; }
template <> __device__ __forceinline__ void store_vec<2>(float* p, const float (&f)[2]) { *(float2*)p = make_float2(f[0], f[1]); }

// SKIP = true (deep levels, B % 16 == 0, Ws a power of two): pixels are taken batch-innermost, k = (i*Ws + j)*B + b, so
// the 16 pixels of a K-tile are the SAME position (i, j) of 16 images.  A tap that falls into the zero padding at
// that position does so for the whole tile; a K-tile none of whose N-tile taps is inside the image is skipped
// outright (about a third of the tiles on the 8x2 level).  The per-tile pixel decode also becomes scalar.
template <int BM, int BN, int WM, int WN, bool SKIP = false, bool SPLIT = false, int PF = 1>     // SPLIT: mfma_split.h (optional mode); PF: K-tiles requested ahead
__global__ __launch_bounds__(256) void wgrad_gemm_kernel(WgradArgs p) {
  constexpr int TM = BM / WM / 16;
  constexpr int TN = BN / WN / 16;
  constexpr int LDA = BM + 4, LDB = BN + 4;
  constexpr int CA = BM / 4, CB = BN / 4;            // 16-byte chunks per k-row
  constexpr int RA = (16 * CA + 255) / 256, RB = (16 * CB + 255) / 256;
  static_assert(WM * WN == 4, "4 waves");

  __shared__ __attribute__((aligned(16))) float As[2][16 * LDA];
  __shared__ __attribute__((aligned(16))) float Bs[2][16 * LDB];

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int lrow = lane & 15, q = lane >> 4;

  const int Ntot = 25 * p.Cl;
  const int cs0 = blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const long P = (long)p.B * p.Hs * p.Ws;
  const long p_begin = (long)blockIdx.z * p.pix_per_split;
  long p_end = p_begin + p.pix_per_split;
  if (p_end > P) p_end = P;

  constexpr unsigned OOB = 0x80000000u;
  // fixed (k-row, chunk) assignments
  int ak[RA], ac[RA]; bool aok[RA];
#pragma unroll
  for (int r = 0; r < RA; ++r) {
    const int idx = t + 256 * r;
    ak[r] = idx / CA; ac[r] = idx % CA; aok[r] = idx < 16 * CA;
  }
  int bk[RB], bc[RB], bkh[RB], bkw[RB], bcl[RB]; bool bok[RB];
#pragma unroll
  for (int r = 0; r < RB; ++r) {
    const int idx = t + 256 * r;
    bk[r] = idx / CB; bc[r] = idx % CB;
    const int n = n0 + bc[r] * 4;
    bok[r] = (idx < 16 * CB) && (n < Ntot);
    const int tap = bok[r] ? n / p.Cl : 0;
    bcl[r] = bok[r] ? n - tap * p.Cl : 0;
    bkh[r] = tap / 5; bkw[r] = tap - bkh[r] * 5;
  }

  // (b, i, j) of each B row's pixel, advanced by 16 pixels per K-tile with 32-bit arithmetic
  // (a 64-bit divide per row per tile used to cost more VALU time than the tile's MFMAs)
  int bj[RB], bi[RB]; unsigned bimg[RB];            // bimg: pixel index of (b, 0, 0) in L (< 2^31, checked on the host)
#pragma unroll
  for (int r = 0; r < RB; ++r) {
    const unsigned pix = (unsigned)(p_begin + bk[r]);
    const unsigned row = pix / (unsigned)p.Ws;
    bj[r] = (int)(pix - row * (unsigned)p.Ws);
    const unsigned img = row / (unsigned)p.Hs;
    bi[r] = (int)(row - img * (unsigned)p.Hs);
    bimg[r] = img * (unsigned)(p.Hl * p.Wl);
  }
  const unsigned img_stride = (unsigned)(p.Hl * p.Wl);
  auto advance = [&]() {
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      unsigned j = (unsigned)bj[r] + 16u, i = (unsigned)bi[r];
      if (p.ws_shift >= 0) { i += j >> p.ws_shift; j &= (unsigned)p.Ws - 1u; }
      else { const unsigned q = j / (unsigned)p.Ws; j -= q * (unsigned)p.Ws; i += q; }
      unsigned nb;
      if (p.hs_shift >= 0) { nb = i >> p.hs_shift; i &= (unsigned)p.Hs - 1u; }
      else { nb = i / (unsigned)p.Hs; i -= nb * (unsigned)p.Hs; }
      bj[r] = (int)j; bi[r] = (int)i; bimg[r] += nb * img_stride;
    }
  };

  // Operands come through buffer loads (few VALU instructions per tile -- fp32 MFMA and VALU do not overlap on a
  // SIMD): A rows have a fixed per-thread byte offset plus a per-tile scalar offset; B rows a 32-bit offset from
  // the incremental (b,i,j); a row outside the image / the split / the matrix is pointed past num_records and
  // reads zeros.
  unsigned a_voff[RA];
#pragma unroll
  for (int r = 0; r < RA; ++r) a_voff[r] = aok[r] ? (unsigned)(((long)ak[r] * p.lds + cs0 + ac[r] * 4) * 4) : OOB;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.s + p_begin * p.lds), 0, OOB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc((void*)p.l, 0, OOB, 0x00020000);
  // SKIP addressing: thread-constant byte offsets + one scalar offset per tile (position and first image of the tile)
  const int HsWs = p.Hs * p.Ws, HlWl = p.Hl * p.Wl;
  unsigned a_voff_s[RA], b_voff_s[RB];
  int tap_lo = 0, tap_hi = 0;
  if (SKIP) {
#pragma unroll
    for (int r = 0; r < RA; ++r) a_voff_s[r] = aok[r] ? (unsigned)((((long)ak[r] * HsWs) * p.lds + cs0 + ac[r] * 4) * 4) : OOB;
#pragma unroll
    for (int r = 0; r < RB; ++r)
      b_voff_s[r] = (unsigned)((((long)bk[r] * HlWl + bkh[r] * p.Wl + bkw[r]) * p.ldl + bcl[r]) * 4);
    tap_lo = n0 / p.Cl;
    const int nlast = (n0 + BN - 1 < Ntot - 1) ? n0 + BN - 1 : Ntot - 1;
    tap_hi = nlast / p.Cl;
  }
  // L is read relative to pixel (-2, -2) so that every thread-constant offset above is >= 0
  const __amdgpu_buffer_rsrc_t rs_all = __builtin_amdgcn_make_buffer_rsrc((void*)p.s, 0, OOB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rl_sh = __builtin_amdgcn_make_buffer_rsrc((void*)(p.l - (2L * p.Wl + 2) * p.ldl), 0, OOB, 0x00020000);
  auto position = [&](int pk, int& i, int& j, int& b0) {       // scalar: B and Ws are powers of two here
    const int pos = pk >> p.b_shift;
    b0 = pk & (p.B - 1);
    i = pos >> p.ws_shift;
    j = pos & (p.Ws - 1);
  };
  auto first_valid = [&](int pk) -> int {  // first K-tile >= pk (< p_end) with a tap of this N-tile inside the image
    if (!SKIP) return pk;
    while (pk < (int)p_end) {
      int i, j, b0;
      position(pk, i, j, b0);
      bool any = false;
      for (int tp = tap_lo; tp <= tap_hi; ++tp) {
        const int kh = tp / 5, kw = tp - kh * 5;
        any = any || ((unsigned)(2 * i - 2 + kh) < (unsigned)p.Hl && (unsigned)(2 * j - 2 + kw) < (unsigned)p.Wl);
      }
      if (any) return pk;
      pk = ((pk >> p.b_shift) + 1) << p.b_shift;       // next position
    }
    return (int)p_end;
  };
  f32x4 ra[RA], rb[RB];
  f32x4 ra2[PF == 2 ? RA : 1], rb2[PF == 2 ? RB : 1];       // PF = 2: a second register set (requests two K-tiles ahead)
  auto load_tile_skip = [&](int pk, f32x4 (&ra)[RA], f32x4 (&rb)[RB]) __attribute__((always_inline)) {
    int i, j, b0;
    position(pk, i, j, b0);
    const int soff_a = __builtin_amdgcn_readfirstlane((int)((((long)b0 * HsWs + (i * p.Ws + j)) * p.lds) * 4));
    const int soff_b = __builtin_amdgcn_readfirstlane((int)((((long)b0 * HlWl + 2 * i * p.Wl + 2 * j) * p.ldl) * 4));
    const int left = (int)p_end - pk;
#pragma unroll
    for (int r = 0; r < RA; ++r) {
      const unsigned vo = ak[r] < left ? a_voff_s[r] : OOB;
      ra[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_all, (int)vo, soff_a, 0));
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int ih = 2 * i - 2 + bkh[r], iw = 2 * j - 2 + bkw[r];
      const bool ok = bok[r] && bk[r] < left && (unsigned)ih < (unsigned)p.Hl && (unsigned)iw < (unsigned)p.Wl;
      rb[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rl_sh, (int)(ok ? b_voff_s[r] : OOB), soff_b, 0));
    }
  };
  auto load_tile = [&](long pk, f32x4 (&ra)[RA], f32x4 (&rb)[RB]) __attribute__((always_inline)) {      // must be called with pk advancing by 16 from p_begin
    const int soff_a = (int)((pk - p_begin) * p.lds * 4);
    const int left = (int)(p_end - pk);          // rows of this tile inside the split (16 except in its last tile)
#pragma unroll
    for (int r = 0; r < RA; ++r) {
      const unsigned vo = ak[r] < left ? a_voff[r] : OOB;
      ra[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)vo, soff_a, 0));
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int ih = 2 * bi[r] - 2 + bkh[r], iw = 2 * bj[r] - 2 + bkw[r];
      const bool ok = bok[r] && bk[r] < left && (unsigned)ih < (unsigned)p.Hl && (unsigned)iw < (unsigned)p.Wl;
      const unsigned vo = ((bimg[r] + (unsigned)(ih * p.Wl + iw)) * (unsigned)p.ldl + (unsigned)bcl[r]) * 4u;
      rb[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rl, (int)(ok ? vo : OOB), 0, 0));
    }
    advance();
  };
  auto store_tile = [&](int buf, const f32x4 (&ra)[RA], const f32x4 (&rb)[RB]) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < RA; ++r)
      if (aok[r]) *(f32x4*)(&As[buf][ak[r] * LDA + ac[r] * 4]) = ra[r];
#pragma unroll
    for (int r = 0; r < RB; ++r)
      if (t + 256 * r < 16 * CB) *(f32x4*)(&Bs[buf][bk[r] * LDB + bc[r] * 4]) = rb[r];
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto multiply = [&](int buf) __attribute__((always_inline)) {
    if constexpr (SPLIT) {                     // the lane's four k of every tile row / column, split into bf16 limbs
      float ga[4][TM], gb[4][TN];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int kr = 4 * q + k;
        load_vec<TM>(&As[buf][kr * LDA + wm * (TM * 16) + TM * lrow], ga[k]);
        load_vec<TN>(&Bs[buf][kr * LDB + wn * (TN * 16) + TN * lrow], gb[k]);
      }
      SvsSplitA sa[TM];
      SvsSplitB sb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) sa[i] = svs_split_a(ga[0][i], ga[1][i], ga[2][i], ga[3][i]);
#pragma unroll
      for (int j = 0; j < TN; ++j) sb[j] = svs_split_b(gb[0][j], gb[1][j], gb[2][j], gb[3][j]);
      svs_mma_split<TM, TN>(acc, sa, sb);
    } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      // Interleaved tiles: MFMA tile i of this wave owns the rows {TM*r + i : r = 0..15} (and tile j the columns
      // {TN*c + j}), so the TM (TN) operands a lane needs for one k are CONTIGUOUS in the m- (n-) major LDS rows
      // and come with one ds_read_b128 / b64 instead of TM (TN) ds_read_b32.
      float fa[TM], fb[TN];
      const int kr = 4 * q + k;
      load_vec<TM>(&As[buf][kr * LDA + wm * (TM * 16) + TM * lrow], fa);
      load_vec<TN>(&Bs[buf][kr * LDB + wn * (TN * 16) + TN * lrow], fb);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    }
  };
  auto fetch = [&](int pkx, f32x4 (&xa)[RA], f32x4 (&xb)[RB]) __attribute__((always_inline)) { if (SKIP) load_tile_skip(pkx, xa, xb); else load_tile(pkx, xa, xb); };
  int pk = first_valid((int)p_begin);
  if constexpr (PF == 2) {
    // two K-tiles in flight (see conv_gemm_kernel): the request for tile n + 2 goes out when tile n starts
    int pk1 = first_valid(pk + 16);
    if (pk < (int)p_end) fetch(pk, ra, rb);
    if (pk1 < (int)p_end) fetch(pk1, ra2, rb2);
    if (pk < (int)p_end) store_tile(0, ra, rb);
    __syncthreads();
    while (pk < (int)p_end) {
      int pk2 = first_valid(pk1 + 16);
      if (pk2 < (int)p_end) fetch(pk2, ra, rb);
      multiply(0);
      if (pk1 < (int)p_end) store_tile(1, ra2, rb2);
      __syncthreads();
      pk = pk1; pk1 = pk2;
      if (pk >= (int)p_end) break;
      pk2 = first_valid(pk1 + 16);
      if (pk2 < (int)p_end) fetch(pk2, ra2, rb2);
      multiply(1);
      if (pk1 < (int)p_end) store_tile(0, ra, rb);
      __syncthreads();
      pk = pk1; pk1 = pk2;
    }
  } else {
    if (pk < (int)p_end) {
      fetch(pk, ra, rb);
      store_tile(0, ra, rb);
    }
    __syncthreads();
    for (int it = 0; pk < (int)p_end; ++it) {
      const int buf = it & 1;
      const int pkn = first_valid(pk + 16);
      const bool more = pkn < (int)p_end;
      if (more) fetch(pkn, ra, rb);
      pk = pkn;
      multiply(buf);
      if (more) store_tile(buf ^ 1, ra, rb);
      __syncthreads();
    }
  }

  // C/D map of the 16x16 MFMA (col = lane & 15, row = 4*(lane>>4) + reg) through the interleaving above:
  // tile (i, j), reg r holds row TM*(4q+r) + i, column TN*lrow + j -> the TN columns of a lane are adjacent
  float* slab = p.slab + (long)blockIdx.z * p.Cs * Ntot;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int cs = cs0 + wm * (TM * 16) + TM * (4 * q + r) + i;
      const int n = n0 + wn * (TN * 16) + TN * lrow;
      if (n < Ntot) {          // Ntot and n are multiples of TN (Cl % 4 == 0), so the vector is all in or all out
        float v[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) v[j] = acc[i][j][r];
        store_vec<TN>(&slab[(long)cs * Ntot + n], v);
      }
    }
}

// ------------------------------------------------------------------------------------------------
// LDS-window weight gradient for the shallow layers (few channels, many pixels: conv2, conv3, deconv4, deconv5).
// There the GEMM above has M = Cs = 32..128 rows only, so its tiles are small, every windowed pixel is re-staged for
// each of the 25 taps and K is split 147-256 ways.  Here a block walks a run of S-tiles (4 x 16 pixels of one image):
// the tile's S pixels (a 16*MT-channel slice) and the (2*4+3) x (2*16+3) window of L they touch (a 16-channel slice)
// are staged ONCE in LDS, and wave (mt, half) accumulates dw[16 cs of m-tile mt][25 taps x 16 cl] in 25 MFMA
// accumulators over its share of the tile's pixels -- every operand is a ds_read_b32 at base + compile-time offset
// (the tap only changes the offset), there is no per-tap global traffic and no address arithmetic in the loop.
// A block keeps its accumulators across all its tiles and writes one slab per wave-half at the end; the slabs go
// through the same fixed-order reductions as the GEMM's.
//   MFMA operands: A[m = cs][k = pixel] = S[pixel][cs] (pixel-major LDS rows, pitch LS: 4 pixels x 16 cs hit 64
//   different banks), B[k = pixel][n = cl] = L[window pixel of (pixel, tap)][cl] (pitch LL = 24 floats: the four
//   pixels of a k-step are 2 window columns apart -> offsets 0/48/96/144 floats, again 64 different banks).
// ------------------------------------------------------------------------------------------------
struct WgWinArgs {
  const float* s; long lds; const float* l; long ldl;
  int B, Hs, Ws, Cs, Hl, Wl, Cl;
  float* slab;                 // [gridDim.x * (4 / MT)][Cs][25 * Cl]
  int ntiles, tiles_per_block;
};

template <int MT>              // 16-row m-tiles per block (one per wave, or per pair of waves when MT = 2)
__global__ __launch_bounds__(256, 2) void wgrad_window_kernel(WgWinArgs p) {
  constexpr int TH = 4, TW = 16, KS = 4 / MT, CSB = 16 * MT;
  constexpr int LS = CSB + 16;                       // 48 / 80 floats
  constexpr int LH = 2 * TH + 3, LW = 2 * TW + 3, LL = 24;
  constexpr int NS = TH * TW * (CSB / 4), NL = LH * LW * 4;       // 16-byte pieces to stage
  constexpr int RS = (NS + 255) / 256, RL = (NL + 255) / 256;
  constexpr int KSTEPS = TH * TW / 4 / KS;           // k-steps (4 pixels each) per wave and tile
  constexpr unsigned OOB = 0x80000000u;
  __shared__ __attribute__((aligned(16))) float Ssm[TH * TW * LS];
  __shared__ __attribute__((aligned(16))) float Lsm[LH * LW * LL];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lrow = lane & 15, q = lane >> 4;
  const int mt = wave % MT, half = wave / MT;
  const int mchunks = p.Cs / CSB;
  const int m0 = (blockIdx.y % mchunks) * CSB, cl0 = (blockIdx.y / mchunks) * 16;
  const int tiles_w = (p.Ws + TW - 1) / TW, tiles_h = (p.Hs + TH - 1) / TH;
  const int tile_lo = blockIdx.x * p.tiles_per_block;
  const int tile_hi = min(tile_lo + p.tiles_per_block, p.ntiles);

  f32x4 sreg[RS], lreg[RL];
  auto fetch = [&](int tile) {                       // global -> registers; outside the image / the tile -> zeros
    const int tw0 = (tile % tiles_w) * TW, th0 = ((tile / tiles_w) % tiles_h) * TH;
    const long b = tile / (tiles_w * tiles_h);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.s + b * p.Hs * p.Ws * p.lds), 0, OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc((void*)(p.l + b * p.Hl * p.Wl * p.ldl), 0, OOB, 0x00020000);
#pragma unroll
    for (int r = 0; r < RS; ++r) {
      const int e = t + 256 * r;
      const int c4 = e % (CSB / 4), px = e / (CSB / 4);
      const int sh = th0 + px / TW, sw = tw0 + px % TW;
      const bool ok = e < NS && sh < p.Hs && sw < p.Ws;
      const unsigned vo = ok ? (unsigned)(((sh * p.Ws + sw) * (int)p.lds + m0 + c4 * 4) * 4) : OOB;
      sreg[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)vo, 0, 0));
    }
#pragma unroll
    for (int r = 0; r < RL; ++r) {
      const int e = t + 256 * r;
      const int c4 = e & 3, px = e >> 2;
      const int ih = 2 * th0 - 2 + px / LW, iw = 2 * tw0 - 2 + px % LW;
      const bool ok = e < NL && (unsigned)ih < (unsigned)p.Hl && (unsigned)iw < (unsigned)p.Wl;
      const unsigned vo = ok ? (unsigned)(((ih * p.Wl + iw) * (int)p.ldl + cl0 + c4 * 4) * 4) : OOB;
      lreg[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rl, (int)vo, 0, 0));
    }
  };
  auto stage = [&]() {                               // registers -> LDS
#pragma unroll
    for (int r = 0; r < RS; ++r) {
      const int e = t + 256 * r;
      if (e < NS) *(f32x4*)(&Ssm[(e / (CSB / 4)) * LS + (e % (CSB / 4)) * 4]) = sreg[r];
    }
#pragma unroll
    for (int r = 0; r < RL; ++r) {
      const int e = t + 256 * r;
      if (e < NL) *(f32x4*)(&Lsm[(e >> 2) * LL + (e & 3) * 4]) = lreg[r];
    }
  };

  f32x4 acc[25];
#pragma unroll
  for (int i = 0; i < 25; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // this wave's k-steps are rows [half * TH / KS, ...) of the tile; lane (lrow, q): pixel column 4 * (kstep % 4) + q
  const float* abase = &Ssm[(half * (TH / KS) * TW + q) * LS + mt * 16 + lrow];
  const float* bbase = &Lsm[(half * (TH / KS) * 2 * LW + 2 * q) * LL + lrow];

  if (tile_lo < tile_hi) fetch(tile_lo);
  for (int tile = tile_lo; tile < tile_hi; ++tile) {
    __syncthreads();                                 // the previous tile's readers are done
    stage();
    __syncthreads();
    if (tile + 1 < tile_hi) fetch(tile + 1);         // in flight while this tile is multiplied
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int sh = ks / 4, sg = ks % 4;            // row within this wave's share, group of 4 pixel columns
      const float a = abase[(sh * TW + sg * 4) * LS];
#pragma unroll
      for (int tap = 0; tap < 25; ++tap) {
        const int kh = tap / 5, kw = tap % 5;
        const float bv = bbase[((2 * sh + kh) * LW + 8 * sg + kw) * LL];
        acc[tap] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv, acc[tap], 0, 0, 0);
      }
    }
  }
  // acc[tap][r]: row cs = m0 + 16 mt + 4 q + r, column n = tap * Cl + cl0 + lrow
  const int Ntot = 25 * p.Cl;
  float* slab = p.slab + ((long)blockIdx.x * KS + half) * p.Cs * Ntot;
#pragma unroll
  for (int tap = 0; tap < 25; ++tap)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      slab[(long)(m0 + mt * 16 + 4 * q + r) * Ntot + tap * p.Cl + cl0 + lrow] = acc[tap][r];
}

// dw[(cs*Cl + cl)*25 + tap] = sum_z slab[z][cs][tap*Cl + cl].  One block per (cs, 64-channel chunk of cl):
// slabs are read along cl (coalesced), transposed through LDS, and written as one contiguous run of dw.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, int ksplit, int Cs, int Cl,
                                                           float* __restrict__ dw) {
  __shared__ float tile[25][65];
  const int clt = Cl < 64 ? Cl : 64;
  const int cs = blockIdx.x, cl0 = blockIdx.y * clt;
  const long zstride = (long)Cs * Cl * 25;
  const float* src0 = slab + (long)cs * 25 * Cl + cl0;
  // four channels per thread (clt % 4 == 0: Cl % 4 == 0) and four slabs' loads in flight; ascending z as before
  const int q4 = clt >> 2;
  for (int e = threadIdx.x; e < 25 * q4; e += 256) {
    const int tap = e / q4, cl = (e - tap * q4) * 4;
    const float* src = src0 + (long)tap * Cl + cl;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int z = 0;
    for (; z + 4 <= ksplit; z += 4, src += 4 * zstride) {
      const f32x4 a = *(const f32x4*)src, b = *(const f32x4*)(src + zstride), c = *(const f32x4*)(src + 2 * zstride), d = *(const f32x4*)(src + 3 * zstride);
      acc += a; acc += b; acc += c; acc += d;
    }
    for (; z < ksplit; ++z, src += zstride) acc += *(const f32x4*)src;
    tile[tap][cl] = acc[0]; tile[tap][cl + 1] = acc[1]; tile[tap][cl + 2] = acc[2]; tile[tap][cl + 3] = acc[3];
  }
  __syncthreads();
  float* dst = dw + ((long)cs * Cl + cl0) * 25;
  for (int e = threadIdx.x; e < 25 * clt; e += 256) {
    const int cl = e / 25, tap = e - cl * 25;
    dst[e] = tile[tap][cl];
  }
}

struct WgradPlan { int cfg, BM, BN, ksplit; long pps; };

static WgradPlan plan_wgrad(int B, int Hs, int Ws, int Cs, int Cl) {
  WgradPlan pl{};
  if (Cs % 128 == 0) { pl.cfg = 0; pl.BM = 128; }
  else if (Cs % 64 == 0) { pl.cfg = 1; pl.BM = 64; }
  else { pl.cfg = 2; pl.BM = 32; }
  // 64-row tiles where the 128-row ones leave too few blocks per K-split (same-device sweep at B=64: the 8x2 level and
  // the 32-channel windows gain 4-8 %, everything else is neutral or loses)
  if (pl.cfg == 0 && ((long)B * Hs * Ws <= 1024 || Cl <= 32)) { pl.cfg = 1; pl.BM = 64; }
  if (svs_tune_on(SVS_TUNE_WGRAD_CFG)) {     // sweeps only
    const int c = (int)svs_tune(SVS_TUNE_WGRAD_CFG);
    if (c == 0 && Cs % 128 == 0) { pl.cfg = 0; pl.BM = 128; }
    if (c == 1 && Cs % 64 == 0) { pl.cfg = 1; pl.BM = 64; }
    if (c == 2 && Cs % 32 == 0) { pl.cfg = 2; pl.BM = 32; }
  }
  pl.BN = 128;
  const long P = (long)B * Hs * Ws;
  const long blocks_mn = (long)(Cs / pl.BM) * ((25L * Cl + pl.BN - 1) / pl.BN);
  long ks = (1024 + blocks_mn - 1) / blocks_mn;
  const long cap = P / 128 > 1 ? P / 128 : 1;      // >= 8 K-tiles per split
  if (ks > cap) ks = cap;
  if (ks > 512) ks = 512;
  if (ks < 1) ks = 1;
  if (svs_tune_on(SVS_TUNE_WGRAD_KSPLIT)) { long f = svs_tune(SVS_TUNE_WGRAD_KSPLIT); if (f >= 1 && f <= cap) ks = f; }
  long pps = (P + ks - 1) / ks;
  pps = (pps + 15) / 16 * 16;
  ks = (P + pps - 1) / pps;
  pl.ksplit = (int)ks;
  pl.pps = pps;
  return pl;
}

// deep levels: batch-innermost pixels, K-tiles whose taps are all in the padding are skipped (wgrad_gemm_kernel)
static int use_wgrad_skip(int B, int Hs, int Ws, int Cl, long lds, int cfg) {
  const bool can_skip = B >= 16 && (B & (B - 1)) == 0 && (Ws & (Ws - 1)) == 0 &&
                        (long)B * Hs * Ws * lds * 4 < (1L << 31);
  int skip = can_skip && Ws <= 8;
  if (svs_tune_on(SVS_TUNE_WGRAD_SKIP)) { const int f = (int)svs_tune(SVS_TUNE_WGRAD_SKIP); skip = (f == 0) ? 0 : (f == 2) ? can_skip : skip; }   // sweeps, tests
  return skip;
}

#define WG_GROUPS 8      // slabs are pre-summed in WG_GROUPS parallel groups when there are many of them

// window kernel: which layers, and how many blocks / slabs
struct WgWinPlan { int use, MT, gy, gx, tpb, ntiles, nslab; };
static WgWinPlan plan_wgrad_window(int B, int Hs, int Ws, int Cs, int Cl) {
  WgWinPlan w{};
  const bool eligible = (Cl == 16 || Cl == 32) && (Cs == 32 || Cs == 64 || Cs == 128);
  w.ntiles = B * ((Hs + 3) / 4) * ((Ws + 15) / 16);
  w.use = eligible && w.ntiles >= 1024;
  if (svs_tune_on(SVS_TUNE_WGRAD_WINDOW)) { const int f = (int)svs_tune(SVS_TUNE_WGRAD_WINDOW); w.use = (f == 0) ? 0 : (f == 2) ? eligible : w.use; }   // sweeps, tests
  if (!w.use) return w;
  w.MT = (Cs % 64 == 0) ? 4 : 2;
  w.gy = (Cs / (16 * w.MT)) * (Cl / 16);
  int gx = 512 / w.gy;                       // two resident blocks per CU
  if (gx > w.ntiles) gx = w.ntiles;
  w.tpb = (w.ntiles + gx - 1) / gx;
  w.gx = (w.ntiles + w.tpb - 1) / w.tpb;
  w.nslab = w.gx * (4 / w.MT);
  return w;
}

size_t svs_wgrad_gemm_workspace(int B, int Hs, int Ws, int Cs, int Cl) {
  WgradPlan pl = plan_wgrad(B, Hs, Ws, Cs, Cl);
  size_t nslab = (size_t)pl.ksplit;
  // the window kernel's slab count does not depend on the sweep / test switches' current value: take the larger
  const int tiles = B * ((Hs + 3) / 4) * ((Ws + 15) / 16);
  if ((Cl == 16 || Cl == 32) && (Cs == 32 || Cs == 64 || Cs == 128)) {
    const int MT = (Cs % 64 == 0) ? 4 : 2, gy = (Cs / (16 * MT)) * (Cl / 16);
    size_t gx = (size_t)(512 / gy < tiles ? 512 / gy : tiles);
    if (gx * (4 / MT) > nslab) nslab = gx * (4 / MT);
  }
  return (nslab + WG_GROUPS) * Cs * 25 * Cl * sizeof(float);
}

// fixed-order sum of nslab slabs [Cs][25*Cl] into torch's (cs, cl, kh, kw) layout; tmp: WG_GROUPS more slabs
static int wgrad_reduce_run(const float* slabs, int nslab, int Cs, int Cl, float* dw, float* tmp, hipStream_t stream) {
  const long n = (long)Cs * 25 * Cl;
  if (nslab > 2 * WG_GROUPS) {     // long serial chains per output element: sum in WG_GROUPS parallel groups first
    const int per = (nslab + WG_GROUPS - 1) / WG_GROUPS;
    const int groups = (nslab + per - 1) / per;
    if (int rc = svs_reduce_slabs_run(slabs, nslab, per, groups, n, tmp, stream)) return rc;
    slabs = tmp;
    nslab = groups;
  }
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)Cs, (unsigned)((Cl + 63) / 64)), dim3(256), 0, stream, slabs, nslab, Cs,
                     Cl, dw);
  SVS_CHECK_LAUNCH("wgrad_reduce");
  return SVS_OK;
}

int svs_wgrad_gemm_run(const float* s, long lds, int B, int Hs, int Ws, int Cs, const float* l, long ldl, int Hl,
                       int Wl, int Cl, float* dw, void* ws, size_t ws_bytes, hipStream_t stream, const char* who) {
  SVS_REQUIRE(s && l && dw, "%s: null pointer", who);
  SVS_REQUIRE(Cs >= 32 && Cs % 32 == 0 && Cl >= 16 && Cl % 4 == 0, "%s: unsupported channels Cs=%d Cl=%d", who, Cs, Cl);
  SVS_REQUIRE(Hs == svs_conv_out(Hl) && Ws == svs_conv_out(Wl), "%s: small grid %dx%d does not match %dx%d", who, Hs, Ws, Hl, Wl);
  SVS_REQUIRE(lds >= Cs && lds % 4 == 0 && ldl >= Cl && ldl % 4 == 0, "%s: bad ld", who);
  SVS_REQUIRE(svs_aligned16(s) && svs_aligned16(l) && svs_aligned16(ws), "%s: pointers must be 16-byte aligned", who);
  WgradPlan pl = plan_wgrad(B, Hs, Ws, Cs, Cl);
  const size_t need = svs_wgrad_gemm_workspace(B, Hs, Ws, Cs, Cl);
  if (!ws || ws_bytes < need) {
    svs_set_error("%s: workspace too small (%zu < %zu)", who, ws_bytes, need);
    return SVS_ERR_WORKSPACE;
  }
  const WgWinPlan wp = plan_wgrad_window(B, Hs, Ws, Cs, Cl);
  if (wp.use && (long)Hs * Ws * lds * 4 < (1L << 31) && (long)Hl * Wl * ldl * 4 < (1L << 31)) {
    WgWinArgs wa{s, lds, l, ldl, B, Hs, Ws, Cs, Hl, Wl, Cl, (float*)ws, wp.ntiles, wp.tpb};
    if (wp.MT == 4) hipLaunchKernelGGL((wgrad_window_kernel<4>), dim3(wp.gx, wp.gy), dim3(256), 0, stream, wa);
    else hipLaunchKernelGGL((wgrad_window_kernel<2>), dim3(wp.gx, wp.gy), dim3(256), 0, stream, wa);
    SVS_CHECK_LAUNCH("wgrad_window");
    pl.ksplit = wp.nslab;
    if (svs_tune_flag(SVS_TUNE_SKIP_REDUCE)) return SVS_OK;
    return wgrad_reduce_run((const float*)ws, pl.ksplit, Cs, Cl, dw, (float*)ws + (size_t)pl.ksplit * Cs * 25 * Cl, stream);
  }
  WgradArgs a{};
  a.s = s; a.lds = lds; a.Hs = Hs; a.Ws = Ws; a.Cs = Cs;
  a.l = l; a.ldl = ldl; a.Hl = Hl; a.Wl = Wl; a.Cl = Cl;
  a.B = B; a.slab = (float*)ws; a.pix_per_split = pl.pps;
  auto log2_or_neg = [](int v) { int s = 0; while ((1 << s) < v) ++s; return (1 << s) == v ? s : -1; };
  a.ws_shift = log2_or_neg(Ws); a.hs_shift = log2_or_neg(Hs);
  SVS_REQUIRE((long)B * Hs * Ws < (1L << 31), "%s: pixel count exceeds 2^31", who);
  SVS_REQUIRE((long)B * Hl * Wl * ldl * 4 < (1L << 31) && (long)(pl.pps + 16) * lds * 4 < (1L << 31),
              "%s: operand views need 64-bit offsets; split the batch", who);
  dim3 grid((unsigned)(Cs / pl.BM), (unsigned)((25 * Cl + pl.BN - 1) / pl.BN), (unsigned)pl.ksplit);
  const int skip = use_wgrad_skip(B, Hs, Ws, Cl, lds, pl.cfg);
  if (skip) a.b_shift = log2_or_neg(B);
  // K-tiles requested ahead by the tap-skipping tiles: two (same-device A/B of tools/ab_tune.py WGRAD_PF 1 2 at batch 64: train step
  // 3.465 -> 3.447 ms); WGRAD_PF = 1 / 2 for A/B runs
  int pf = 2;
  if (svs_tune_on(SVS_TUNE_WGRAD_PF)) { const long f = svs_tune(SVS_TUNE_WGRAD_PF); pf = (f == 2 || (f == 3 && pl.cfg == 0) || (f == 4 && pl.cfg == 1)) ? 2 : 1; }   // 3 / 4: one tile shape only
#define SVS_WGRAD_LAUNCH(SPLIT_) \
  if (skip && pf == 2 && pl.cfg == 0) hipLaunchKernelGGL((wgrad_gemm_kernel<128, 128, 2, 2, true, SPLIT_, 2>), grid, dim3(256), 0, stream, a); \
  else if (skip && pf == 2 && pl.cfg == 1) hipLaunchKernelGGL((wgrad_gemm_kernel<64, 128, 1, 4, true, SPLIT_, 2>), grid, dim3(256), 0, stream, a); \
  else if (skip) { \
    if (pl.cfg == 0) hipLaunchKernelGGL((wgrad_gemm_kernel<128, 128, 2, 2, true, SPLIT_>), grid, dim3(256), 0, stream, a); \
    else if (pl.cfg == 1) hipLaunchKernelGGL((wgrad_gemm_kernel<64, 128, 1, 4, true, SPLIT_>), grid, dim3(256), 0, stream, a); \
    else hipLaunchKernelGGL((wgrad_gemm_kernel<32, 128, 1, 4, true, SPLIT_>), grid, dim3(256), 0, stream, a); \
  } else switch (pl.cfg) { \
    case 0: hipLaunchKernelGGL((wgrad_gemm_kernel<128, 128, 2, 2, false, SPLIT_>), grid, dim3(256), 0, stream, a); break; \
    case 1: hipLaunchKernelGGL((wgrad_gemm_kernel<64, 128, 1, 4, false, SPLIT_>), grid, dim3(256), 0, stream, a); break; \
    default: hipLaunchKernelGGL((wgrad_gemm_kernel<32, 128, 1, 4, false, SPLIT_>), grid, dim3(256), 0, stream, a); break; \
  }
  if (svs_tune(SVS_TUNE_MFMA_SPLIT) > 0) { SVS_WGRAD_LAUNCH(true) } else { SVS_WGRAD_LAUNCH(false) }      // optional mode: mfma_split.h
#undef SVS_WGRAD_LAUNCH
  SVS_CHECK_LAUNCH("wgrad_gemm");
  if (svs_tune_flag(SVS_TUNE_SKIP_REDUCE)) return SVS_OK;             // lets bench.py time the GEMM kernel alone
  return wgrad_reduce_run((const float*)ws, pl.ksplit, Cs, Cl, dw, (float*)ws + (size_t)pl.ksplit * Cs * 25 * Cl, stream);
}

int svs_wgrad_gemm_describe(int B, int Hs, int Ws, int Cs, int Cl, char* buf, size_t n) {
  const WgWinPlan wp = plan_wgrad_window(B, Hs, Ws, Cs, Cl);
  if (wp.use) { snprintf(buf, n, "wgrad_window_kernel<%d>", wp.MT); return wp.nslab; }
  const WgradPlan pl = plan_wgrad(B, Hs, Ws, Cs, Cl);
  const bool skip = use_wgrad_skip(B, Hs, Ws, Cl, Cs, pl.cfg) != 0;
  int pf = 2;
  if (svs_tune_on(SVS_TUNE_WGRAD_PF)) { const long f = svs_tune(SVS_TUNE_WGRAD_PF); pf = (f == 2 || (f == 3 && pl.cfg == 0) || (f == 4 && pl.cfg == 1)) ? 2 : 1; }
  if (!skip || pl.cfg > 1) pf = 1;           // (the two-ahead variant exists for the tap-skipping 128x128 and 64x128 tiles)
  snprintf(buf, n, "wgrad_gemm_kernel<%d, %d, %d, %d, %s, %s, %d>", pl.BM, pl.BN, pl.cfg == 0 ? 2 : 1, pl.cfg == 0 ? 2 : 4,
           skip ? "true" : "false", svs_tune(SVS_TUNE_MFMA_SPLIT) > 0 ? "true" : "false", pf);
  return pl.ksplit;
}
