// Implicit-GEMM 5x5 / stride-2 convolutions on the fp32 MFMA (v_mfma_f32_16x16x4_f32), gfx950.
//
// Two index maps share one kernel (MODE):
//   MODE_GATHER  y[b,oh,ow,n] = sum_{kh,kw,c} x[b,2oh-2+kh,2ow-2+kw,c] * wp[n][kh][kw][c]
//                -> Conv2d forward (reference model.py:48-74) and ConvTranspose2d backward-data.
//   MODE_PARITY  y[b,2a+ph,2c+pw,n] = sum_{th,tw,c} x[b,a+1-th,c+1-tw,c] * wp[p][n][th][tw][c]
//                (blockIdx.z = p = 2*ph+pw, 9/6/6/4 taps: the four stride-1 sub-convolutions of a
//                stride-2 transposed conv) -> ConvTranspose2d forward (model.py:79-109) and Conv2d
//                backward-data.
// GEMM view: M = output pixels of the (parity) grid, N = output channels, K = taps x input channels,
// K-tile = 16 consecutive input channels of one tap (NHWC makes them one 64-byte run in HBM).
//
// Workgroup: 256 threads = 4 waves arranged WM x WN; each wave owns (BM/WM) x (BN/WN) outputs as
// 16x16 MFMA tiles.  A (im2col rows) and B (weights) K-tiles are register-staged into LDS (16-byte
// global loads, XOR-swizzled 64-byte LDS rows so the ds_read_b128 fragment reads are conflict-free),
// double-buffered with one barrier per K-tile.  Each lane reads 4 consecutive k of its row with one
// ds_read_b128 and feeds them to 4 MFMAs: lane (row, q) supplies k = 4q+j to MFMA j on both the A and
// the B side, so the K order inside a tile is permuted consistently and no shuffles are needed.
//
// Bound: MFMA (fp32 157.3 TFLOP/s peak); algorithmic FLOPs = 2*M*N*K.
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <type_traits>
#include <vector>
#include "common.h"
#include "mfma_split.h"

enum { MODE_GATHER = 0, MODE_PARITY = 1 };

struct ConvGemmArgs {
  const float* x; long ldx;
  int B, H, W, C;                 // input geometry; C = channels reduced over
  const float* wp;                // packed weights (see header)
  const float* bias;              // [N] or null
  const float* scale;             // [N] or null: epilogue v*scale+shift then leaky(slope)
  const float* shift;
  float slope;
  float* y; long ldy;
  int Ho, Wo, N;
  int accumulate;
  int ksplit;                     // gridDim.z
  float* slab;                    // [ksplit][B*Ho*Wo][N] partial sums when ksplit > 1
  int tap_inner;                  // K-tile order: 1 = channel chunk outer / tap inner, 0 = tap outer / chunk inner
  int cpt_shift;                  // log2(C / 16) when that is a power of two, else -1
  float* stats;                   // non-split kernels: BatchNorm partials of the output, [row][2][N] (row = class * mtiles + mtile
                                  // for the GEMM, = block for the window kernel); null = off
};

// Balanced K-splits of the tap-skipping kernels (SKIP = true, batch a multiple of the tile height: an M-tile is then ONE pixel
// position of BM images).  On the deep levels a position has 9 .. 25 taps inside the image, so with one K-split count for the
// whole layer the blocks of a launch differ in length by up to 2.8x (and a split whose K range lies in the padding does
// nothing); all blocks of these launches are resident at once, so the launch lasts as long as the most loaded CU: 1.2-1.5x
// the mean (rocprofv3: MFMA-busy 0.39-0.46 of these launches against 0.65-0.72 of the ones with even blocks).  Here every
// position gets its OWN number of splits, proportional to its valid K-tiles, each split an equal share of the VALID K-tiles
// only; the host lays the blocks out as (class, position, M-tile, split, N-tile) in one 1-D grid and picks the split size that
// minimises the modelled load of the most loaded CU.  Block -> position by a 64-lane ballot over `first`.  Slab z of a row
// exists only for z < its position's split count, which the blocks record per output row (`rowsplit`) for the epilogue.
struct ConvBal {
  int enabled;                    // 0: uniform grid (tiles, K-splits, classes)
  int npos[4];                    // positions (= Ha * Wa) of each parity class (GATHER: class 0 only)
  unsigned first[4][65];          // first[c][pos]: first block of position pos of class c; first[c][npos[c]] = one past its last
  unsigned char nsplit[4][64];
  unsigned char* rowsplit;        // [B * Ho * Wo] (workspace, behind the slabs)
};
struct ConvNoBal {};

__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((row >> 1) & 3); }

// SKIP = true: rows are ordered (w, h, b) -- batch innermost -- instead of (b, h, w).  On the deep levels the images
// are tiny (8x2 .. 32x8 anchors) and a third of the taps of an edge pixel fall into the zero padding; with the batch
// innermost the rows of one M-tile are the SAME pixel position(s) of many images, so such a tap is out of the image
// for all of them at once and its K-tiles are skipped outright (no loads, no MFMAs): -30% work on the 8x2 level,
// -15% on 16x4.  Skipped products are exact zeros, so results do not change.  Needs tap-outer K order and C/16 a
// power of two.
template <int MODE, int BM, int BN, int WM, int WN, bool SKIP = false, bool SPLIT = false, int PF = 1>     // SPLIT: mfma_split.h (optional mode); PF: K-tiles requested ahead
__global__ __launch_bounds__(256) void conv_gemm_kernel(ConvGemmArgs p, std::conditional_t<SKIP, ConvBal, ConvNoBal> bal) {
  constexpr int TM = BM / WM / 16;
  constexpr int TN = BN / WN / 16;
  constexpr int RA = (BM + 63) / 64;   // A rows staged per thread
  constexpr int RB = (BN + 63) / 64;
  static_assert(WM * WN == 4, "4 waves");
  static_assert(TM >= 1 && TN >= 1, "tile");

  __shared__ __attribute__((aligned(16))) float As[2][BM * 16];
  __shared__ __attribute__((aligned(16))) float Bs[2][BN * 16];

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int lrow = lane & 15, q = lane >> 4;

  // ---- geometry of this block ---------------------------------------------------------------
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;     // (tile, K-split, parity class)
  int nsplit = p.ksplit;                                     // K-splits of this block's M-tile
  bool balanced = false;
  if constexpr (SKIP) {
    if (bal.enabled) {                                       // 1-D grid: (class, position, M-tile of the position, split, N-tile)
      balanced = true;
      const unsigned bid = blockIdx.x;
      int c = 0;
      if (MODE == MODE_PARITY) c = (bid >= bal.first[1][0]) + (bid >= bal.first[2][0]) + (bid >= bal.first[3][0]);
      const unsigned f = lane < bal.npos[c] ? bal.first[c][lane] : 0xFFFFFFFFu;
      const int pos = __builtin_amdgcn_readfirstlane(__builtin_popcountll(__ballot(f <= bid)) - 1);
      nsplit = bal.nsplit[c][pos];
      const int ntn = p.N / BN, per = nsplit * ntn;
      const int within = (int)(bid - bal.first[c][pos]);
      const int i = within / per, rem = within - i * per;
      by = rem / ntn;
      bx = (pos * (p.B / BM) + i) * ntn + (rem - by * ntn);
      bz = c;
    }
  }
  int ph = 0, pw = 0, nth = 5, ntw = 5;
  int Ha, Wa;                      // rows / cols of the M grid
  const float* wp = p.wp;
  if (MODE == MODE_PARITY) {
    const int par = bz;                     // classes outermost in dispatch order: all the long (9-tap) blocks start first
    ph = par >> 1; pw = par & 1;
    nth = 3 - ph; ntw = 3 - pw;
    Ha = (p.Ho - ph + 1) >> 1; Wa = (p.Wo - pw + 1) >> 1;
    const int poff = (par == 0) ? 0 : (par == 1) ? 9 : (par == 2) ? 15 : 21;
    wp += (long)poff * p.N * p.C;
  } else {
    Ha = p.Ho; Wa = p.Wo;
  }
  const int ntaps = nth * ntw;
  const long M = (long)p.B * Ha * Wa;
  const int ntile_n = p.N / BN;
  const long m0 = (long)(bx / ntile_n) * BM;
  const int n0 = (bx % ntile_n) * BN;
  if (m0 >= M) {                   // parity classes of odd-sized outputs are smaller
    if (p.stats && p.ksplit == 1) {                // their statistics row must still exist
      float* out = p.stats + ((long)bz * (gridDim.x / ntile_n) + bx / ntile_n) * 2 * p.N + n0;
      for (int c = threadIdx.x; c < BN; c += 256) { out[c] = 0.f; out[p.N + c] = 0.f; }
    }
    return;
  }
  const int cpt = p.C >> 4;        // K-tiles per tap
  const int nkt = ntaps * cpt;
  int kt_begin = (int)((long)nkt * by / nsplit);
  int kt_end = (int)((long)nkt * (by + 1) / nsplit);
  const long Kw = (long)ntaps * p.C;   // weight row length

  // ---- per-thread staging rows -----------------------------------------------------------------
  // fp32 MFMA and VALU instructions do not overlap on a SIMD (tools/mfma_valu_probe.hip: time ~ MFMA cycles +
  // VALU cycles whatever the occupancy), so the K-loop spends as few VALU instructions per MFMA as possible:
  // operands are fetched with buffer loads -- per row a 32-bit byte offset computed ONCE, per tile a scalar
  // (SGPR) offset for the tap / channel chunk, and a 25-bit per-row validity mask decided ONCE; an invalid row
  // (zero padding, M tail, unused B row) is pointed past num_records and the hardware range check returns
  // zeros.  Per A row and tile that is a bit test and a select -- no address arithmetic, no branches.
  const int chunk = t & 3;
  constexpr unsigned OOB = 0x80000000u;           // >= num_records of both descriptors
  unsigned a_voff[RA], a_mask[RA];
#pragma unroll
  for (int r = 0; r < RA; ++r) {
    const int row = (t >> 2) + 64 * r;
    const long m = m0 + row;
    const bool ok = (row < BM) && (m < M);
    const long mm = ok ? m : 0;
    int wq, hq; long b;
    if (SKIP) {                                              // 32-bit: M < 2^31 follows from the host's 2 GiB view check
      const unsigned pos = (unsigned)mm / (unsigned)p.B;
      b = (long)((unsigned)mm - pos * (unsigned)p.B);
      wq = (int)(pos / (unsigned)Ha);
      hq = (int)(pos - (unsigned)wq * (unsigned)Ha);
    } else {
      const unsigned um = (unsigned)mm, utmp = um / (unsigned)Wa;
      wq = (int)(um - utmp * (unsigned)Wa);
      const unsigned ub = utmp / (unsigned)Ha;
      hq = (int)(utmp - ub * (unsigned)Ha);
      b = ub;
    }
    const int h0 = (MODE == MODE_GATHER) ? 2 * hq : hq;      // anchor pixel of the row
    const int w0 = (MODE == MODE_GATHER) ? 2 * wq : wq;
    a_voff[r] = ok ? (unsigned)((((b * p.H + h0) * p.W + w0) * p.ldx + chunk * 4) * 4) : OOB;
    unsigned mask = 0;
    for (int th = 0; th < nth; ++th)
      for (int tw = 0; tw < ntw; ++tw) {
        const int ih = (MODE == MODE_GATHER) ? h0 - 2 + th : h0 + 1 - th;
        const int iw = (MODE == MODE_GATHER) ? w0 - 2 + tw : w0 + 1 - tw;
        if (ok && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W) mask |= 1u << (th * ntw + tw);
      }
    a_mask[r] = mask;
  }
  unsigned umask = 0x1FFFFFFu;                    // taps that at least one row of this tile needs (block-uniform)
  if (SKIP) {
    __shared__ unsigned umask_s;
    if (t == 0) umask_s = 0;
    __syncthreads();
    unsigned mine = 0;
#pragma unroll
    for (int r = 0; r < RA; ++r) mine |= a_mask[r];
    if (mine) atomicOr(&umask_s, mine);
    __syncthreads();
    umask = __builtin_amdgcn_readfirstlane(umask_s);
    if (balanced) {                // this split's equal share of the VALID K-tiles: ranks [vb, ve) among the set taps of umask
      const int nvk = __builtin_popcount(umask) << p.cpt_shift;
      const int vb = (int)((long)nvk * by / nsplit), ve = (int)((long)nvk * (by + 1) / nsplit);
      auto kt_of = [&](int v) -> int {               // v-th valid K-tile -> K-tile index (scalar loop over <= 25 bits)
        unsigned m = umask;
        for (int r = v >> p.cpt_shift; r > 0; --r) m &= m - 1;
        return (__builtin_ctz(m) << p.cpt_shift) | (v & (cpt - 1));
      };
      kt_begin = vb < nvk ? kt_of(vb) : nkt;
      kt_end = ve < nvk ? kt_of(ve) : nkt;
    }
  }
  unsigned b_voff[RB];
#pragma unroll
  for (int r = 0; r < RB; ++r) {
    const int row = (t >> 2) + 64 * r;
    b_voff[r] = (row < BN) ? (unsigned)(((long)(n0 + row) * Kw + chunk * 4) * 4) : OOB;
  }
  // base pointer shifted so that every per-tile scalar offset is >= 0: GATHER tap (th,tw) reads pixel
  // anchor + (th-2, tw-2) = [anchor - (2,2)] + (th,tw); PARITY reads anchor + (1-th, 1-tw) = [anchor - (1,1)] + (2-th, 2-tw)
  const long shift = (MODE == MODE_GATHER) ? (2L * p.W + 2) * p.ldx : (1L * p.W + 1) * p.ldx;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x - shift), 0, OOB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)wp, 0, OOB, 0x00020000);

  f32x4 ra[RA], rb[RB];
  f32x4 ra2[PF == 2 ? RA : 1], rb2[PF == 2 ? RB : 1];       // PF = 2: a second register set (requests two K-tiles ahead)
  // K-tile kt -> (tap, channel chunk), from block-uniform values only (plain scalar arithmetic: mutable loader
  // state captured by the lambdas below used to end up in scratch memory, with waterfall loops around the loads)
  auto load_tile_to = [&](int kt, f32x4 (&ra)[RA], f32x4 (&rb)[RB]) __attribute__((always_inline)) {
    int tap, cc;
    if (SKIP || !p.tap_inner) {            // tap outer, chunk inner
      if (p.cpt_shift >= 0) { tap = kt >> p.cpt_shift; cc = kt & (cpt - 1); }
      else { tap = kt / cpt; cc = kt - tap * cpt; }
    } else {                               // chunk outer, tap inner
      cc = (ntaps == 25) ? kt / 25 : (ntaps == 9) ? kt / 9 : (ntaps == 6) ? kt / 6 : kt >> 2;
      tap = kt - cc * ntaps;
    }
    const int th = (ntw == 5) ? tap / 5 : (ntw == 3) ? tap / 3 : tap >> 1;
    const int tw = tap - th * ntw;
    const int pix = (MODE == MODE_GATHER) ? th * p.W + tw : (2 - th) * p.W + (2 - tw);
    const int soff_a = __builtin_amdgcn_readfirstlane((int)((pix * p.ldx + (cc << 4)) * 4));
    const int soff_b = __builtin_amdgcn_readfirstlane((int)(((long)tap * p.C + (cc << 4)) * 4));
#pragma unroll
    for (int r = 0; r < RA; ++r) {
      const unsigned vo = ((a_mask[r] >> tap) & 1u) ? a_voff[r] : OOB;
      ra[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)vo, soff_a, 0));
    }
#pragma unroll
    for (int r = 0; r < RB; ++r)
      rb[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, (int)b_voff[r], soff_b, 0));
  };
  auto load_tile = [&](int kt) __attribute__((always_inline)) { load_tile_to(kt, ra, rb); };
  auto store_tile_from = [&](int buf, const f32x4 (&ra)[RA], const f32x4 (&rb)[RB]) __attribute__((always_inline)) {       // unconditional when the tile height is a multiple of 64 rows (no exec-mask branches)
#pragma unroll
    for (int r = 0; r < RA; ++r) {
      const int row = (t >> 2) + 64 * r;
      if (BM % 64 == 0 || row < BM) *(f32x4*)(&As[buf][row * 16 + swz(row, chunk) * 4]) = ra[r];
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int row = (t >> 2) + 64 * r;
      if (BN % 64 == 0 || row < BN) *(f32x4*)(&Bs[buf][row * 16 + swz(row, chunk) * 4]) = rb[r];
    }
  };

  auto store_tile = [&](int buf) __attribute__((always_inline)) { store_tile_from(buf, ra, rb); };
  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // first K-tile >= k (< kt_end) whose tap some row of this tile needs, else kt_end; scalar arithmetic only
  auto first_valid = [&](int k) -> int {
    if (!SKIP || k >= kt_end) return k;
    const int tp = k >> p.cpt_shift;
    if ((umask >> tp) & 1u) return k;
    const unsigned rest = umask >> (tp + 1);
    if (!rest) return kt_end;
    const int nk = (tp + 1 + __builtin_ctz(rest)) << p.cpt_shift;
    return nk < kt_end ? nk : kt_end;
  };
  auto multiply = [&](int buf) __attribute__((always_inline)) {
    f32x4 fa[TM], fb[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row = wm * (TM * 16) + i * 16 + lrow;
      fa[i] = *(const f32x4*)(&As[buf][row * 16 + swz(row, q) * 4]);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int row = wn * (TN * 16) + j * 16 + lrow;
      fb[j] = *(const f32x4*)(&Bs[buf][row * 16 + swz(row, q) * 4]);
    }
    if constexpr (SPLIT) {
      SvsSplitA sa[TM];
      SvsSplitB sb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) sa[i] = svs_split_a(fa[i][0], fa[i][1], fa[i][2], fa[i][3]);
#pragma unroll
      for (int j = 0; j < TN; ++j) sb[j] = svs_split_b(fb[j][0], fb[j][1], fb[j][2], fb[j][3]);
      svs_mma_split<TM, TN>(acc, sa, sb);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][k], fb[j][k], acc[i][j], 0, 0, 0);
    }
  };
  int kt_cur = first_valid(kt_begin);
  if constexpr (PF == 2) {
    // Two K-tiles in flight: the request for tile n + 2 goes out when tile n starts, so an operand has two tile times to arrive
    // before it is written to LDS: one tile time -- 512 MFMA cycles on the 64x64 tile -- is less than an L2 / Infinity-Cache
    // round trip under load (per layer at batch 64, one against two tiles ahead: deconv3 forward 130.7 -> 115.2 us, conv4
    // backward-data 70.1 -> 63.3, the deep layers -1 ... -5 %).  Written out as two phases so that register set and LDS buffer are
    // compile-time choices.  THREE tiles ahead (six phases) and a generic ring of PF sets with computed indices both measured
    // ~5 % SLOWER on the whole train step (3.60 / 3.63 against 3.43 / 3.445 ms, same device): the loop body triples.
    int kt_n1 = first_valid(kt_cur + 1);
    if (kt_cur < kt_end) load_tile_to(kt_cur, ra, rb);
    if (kt_n1 < kt_end) load_tile_to(kt_n1, ra2, rb2);
    if (kt_cur < kt_end) store_tile_from(0, ra, rb);
    __syncthreads();
    while (kt_cur < kt_end) {
      int kt_n2 = first_valid(kt_n1 + 1);                     // tile in LDS buffer 0; set 1 holds n1; set 0 is free
      if (kt_n2 < kt_end) load_tile_to(kt_n2, ra, rb);
      multiply(0);
      if (kt_n1 < kt_end) store_tile_from(1, ra2, rb2);
      __syncthreads();
      kt_cur = kt_n1; kt_n1 = kt_n2;
      if (kt_cur >= kt_end) break;
      kt_n2 = first_valid(kt_n1 + 1);                         // tile in LDS buffer 1; set 0 holds n1; set 1 is free
      if (kt_n2 < kt_end) load_tile_to(kt_n2, ra2, rb2);
      multiply(1);
      if (kt_n1 < kt_end) store_tile_from(0, ra, rb);
      __syncthreads();
      kt_cur = kt_n1; kt_n1 = kt_n2;
    }
  } else {
  if (kt_cur < kt_end) {
    load_tile(kt_cur);
    store_tile(0);
  }
  __syncthreads();
  for (int it = 0; kt_cur < kt_end; ++it) {
    const int buf = it & 1;
    const int kt_next = first_valid(kt_cur + 1);
    const bool more = kt_next < kt_end;
    if (more) load_tile(kt_next);
    kt_cur = kt_next;
    f32x4 fa[TM], fb[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row = wm * (TM * 16) + i * 16 + lrow;
      fa[i] = *(const f32x4*)(&As[buf][row * 16 + swz(row, q) * 4]);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int row = wn * (TN * 16) + j * 16 + lrow;
      fb[j] = *(const f32x4*)(&Bs[buf][row * 16 + swz(row, q) * 4]);
    }
    if constexpr (SPLIT) {
      SvsSplitA sa[TM];
      SvsSplitB sb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) sa[i] = svs_split_a(fa[i][0], fa[i][1], fa[i][2], fa[i][3]);
#pragma unroll
      for (int j = 0; j < TN; ++j) sb[j] = svs_split_b(fb[j][0], fb[j][1], fb[j][2], fb[j][3]);
      svs_mma_split<TM, TN>(acc, sa, sb);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][k], fb[j][k], acc[i][j], 0, 0, 0);
    }
    if (more) store_tile(buf ^ 1);
    __syncthreads();
  }
  }

  // ---- epilogue: C/D map of the 16x16 MFMA: col = lane & 15, row = 4*(lane>>4) + reg ------------
  const bool split = p.ksplit > 1;
  float ssum[TN], ssq[TN];               // per-lane column sums of the values written (BatchNorm statistics)
#pragma unroll
  for (int j = 0; j < TN; ++j) { ssum[j] = 0.f; ssq[j] = 0.f; }
  float* const slab = split ? p.slab + (long)by * ((long)p.B * p.Ho * p.Wo) * p.N : nullptr;
  bool mark_rows = false;                // balanced: split 0 of N-tile 0 records how many slabs its rows have
  if constexpr (SKIP) mark_rows = balanced && by == 0 && n0 == 0 && wn == 0;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long m = m0 + wm * (TM * 16) + i * 16 + q * 4 + r;
      if (m >= M) continue;
      long opix;
      if (SKIP) {
        const unsigned pos = (unsigned)m / (unsigned)p.B;
        const long b = (long)((unsigned)m - pos * (unsigned)p.B);
        const int wq = (int)(pos / (unsigned)Ha);
        const int hq = (int)(pos - (unsigned)wq * (unsigned)Ha);
        opix = (MODE == MODE_GATHER) ? (b * p.Ho + hq) * p.Wo + wq : (b * p.Ho + 2 * hq + ph) * p.Wo + 2 * wq + pw;
      } else if (MODE == MODE_GATHER) {
        opix = m;
      } else {                                     // (32-bit: M < 2^31, see the prologue; 64-bit divisions cost ~100 instructions each)
        const unsigned um = (unsigned)m, utmp = um / (unsigned)Wa;
        const int wq = (int)(um - utmp * (unsigned)Wa);
        const unsigned ub = utmp / (unsigned)Ha;
        const int hq = (int)(utmp - ub * (unsigned)Ha);
        const long b = ub;
        opix = (b * p.Ho + 2 * hq + ph) * p.Wo + 2 * wq + pw;
      }
      if constexpr (SKIP) { if (mark_rows && lrow == 0) bal.rowsplit[opix] = (unsigned char)nsplit; }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (TN * 16) + j * 16 + lrow;
        float v = acc[i][j][r];
        if (split) {
          slab[opix * p.N + n] = v;
        } else {
          if (p.bias) v += p.bias[n];
          if (p.scale) {
            v = v * p.scale[n] + p.shift[n];
            v = v > 0.f ? v : v * p.slope;
          }
          float* dst = p.y + opix * p.ldy + n;
          if (p.accumulate) v += *dst;
          *dst = v;
          ssum[j] += v;
          ssq[j] += v * v;
        }
      }
    }
  }
  if (p.stats && !split) {
    // column sums of this tile: over the four 4-row groups of a wave (lanes 16 apart), then over the WM waves
    __shared__ float st[2][WM][BN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      ssum[j] += __shfl_xor(ssum[j], 16, 64); ssum[j] += __shfl_xor(ssum[j], 32, 64);
      ssq[j] += __shfl_xor(ssq[j], 16, 64); ssq[j] += __shfl_xor(ssq[j], 32, 64);
      if (q == 0) { st[0][wm][wn * (TN * 16) + j * 16 + lrow] = ssum[j]; st[1][wm][wn * (TN * 16) + j * 16 + lrow] = ssq[j]; }
    }
    __syncthreads();
    float* out = p.stats + ((long)bz * (gridDim.x / ntile_n) + bx / ntile_n) * 2 * p.N + n0;
    for (int c = t; c < BN; c += 256) {
      float a = 0.f, b2 = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) { a += st[0][w][c]; b2 += st[1][w][c]; }
      out[c] = a;
      out[p.N + c] = b2;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// LDS-free variant for the layers with few output channels (N = 16 / 32: conv2, deconv4/5 and the
// backward-data of conv2/conv3).  With N that small every im2col element feeds only N MACs, so moving it
// global -> VGPR -> LDS -> VGPR (three wide data-movement instructions per 4 MFMAs, none of which overlaps the
// fp32 MFMA on a SIMD: tools/mfma_valu_probe.hip) costs more issue time than the MFMAs themselves.  Here each
// lane loads its MFMA fragment straight from global memory: lane (row r, quarter q) of row-tile i reads the 16
// bytes x[pixel(i, r)][c0 + 4q .. 4q+3], i.e. the four k it supplies to four consecutive MFMAs, with ONE
// buffer load (hardware range check = zero padding, 25-bit validity mask per row decided once, tap / chunk
// offset in an SGPR).  No LDS, no barriers, waves are independent; fragments are double-buffered in registers.
// ------------------------------------------------------------------------------------------------
template <int MODE, int TM, int TN>
__global__ __launch_bounds__(256) void conv_direct_kernel(ConvGemmArgs p) {
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int lrow = lane & 15, q = lane >> 4;
  int ph = 0, pw = 0, nth = 5, ntw = 5, Ha, Wa;
  const float* wp = p.wp;
  if (MODE == MODE_PARITY) {
    const int par = blockIdx.y;
    ph = par >> 1; pw = par & 1;
    nth = 3 - ph; ntw = 3 - pw;
    Ha = (p.Ho - ph + 1) >> 1; Wa = (p.Wo - pw + 1) >> 1;
    const int poff = (par == 0) ? 0 : (par == 1) ? 9 : (par == 2) ? 15 : 21;
    wp += (long)poff * p.N * p.C;
  } else {
    Ha = p.Ho; Wa = p.Wo;
  }
  const int ntaps = nth * ntw;
  const long M = (long)p.B * Ha * Wa;
  const long m0 = ((long)blockIdx.x * 4 + wave) * (TM * 16);
  if (m0 >= M) return;
  const int cpt = p.C >> 4;
  const int nkt = ntaps * cpt;
  const long Kw = (long)ntaps * p.C;
  constexpr unsigned OOB = 0x80000000u;

  unsigned a_voff[TM], a_mask[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const long m = m0 + i * 16 + lrow;
    const bool ok = m < M;
    const long mm = ok ? m : 0;
    const unsigned um = (unsigned)mm, utmp = um / (unsigned)Wa;      // (M < 2^31: 32-bit divisions)
    const int wq = (int)(um - utmp * (unsigned)Wa);
    const unsigned ub = utmp / (unsigned)Ha;
    const int hq = (int)(utmp - ub * (unsigned)Ha);
    const long b = ub;
    const int h0 = (MODE == MODE_GATHER) ? 2 * hq : hq;
    const int w0 = (MODE == MODE_GATHER) ? 2 * wq : wq;
    a_voff[i] = ok ? (unsigned)((((b * p.H + h0) * p.W + w0) * p.ldx + q * 4) * 4) : OOB;
    unsigned mask = 0;
    for (int th = 0; th < nth; ++th)
      for (int tw = 0; tw < ntw; ++tw) {
        const int ih = (MODE == MODE_GATHER) ? h0 - 2 + th : h0 + 1 - th;
        const int iw = (MODE == MODE_GATHER) ? w0 - 2 + tw : w0 + 1 - tw;
        if (ok && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W) mask |= 1u << (th * ntw + tw);
      }
    a_mask[i] = mask;
  }
  unsigned b_voff[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) b_voff[j] = (unsigned)(((long)(j * 16 + lrow) * Kw + q * 4) * 4);
  // base shifted so that every per-tile scalar offset is >= 0 (see conv_gemm_kernel for the tap geometry)
  const long shift = (MODE == MODE_GATHER) ? (2L * p.W + 2) * p.ldx : (1L * p.W + 1) * p.ldx;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x - shift), 0, OOB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)wp, 0, OOB, 0x00020000);

  int l_th = 0, l_tw = 0, l_cc = 0;        // chunk outer, tap inner
  auto load_frags = [&](f32x4 (&fa)[TM], f32x4 (&fb)[TN]) {
    const int th = l_th, tw = l_tw, tap = th * ntw + tw;
    const int pix = (MODE == MODE_GATHER) ? th * p.W + tw : (2 - th) * p.W + (2 - tw);
    const int soff_a = (int)((pix * p.ldx + (l_cc << 4)) * 4);
    const int soff_b = (int)(((long)tap * p.C + (l_cc << 4)) * 4);
    if (++l_tw == ntw) { l_tw = 0; if (++l_th == nth) { l_th = 0; ++l_cc; } }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const unsigned vo = ((a_mask[i] >> tap) & 1u) ? a_voff[i] : OOB;
      fa[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)vo, soff_a, 0));
    }
#pragma unroll
    for (int j = 0; j < TN; ++j)
      fb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, (int)b_voff[j], soff_b, 0));
  };
  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto mma = [&](const f32x4 (&fa)[TM], const f32x4 (&fb)[TN]) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][k], fb[j][k], acc[i][j], 0, 0, 0);
  };
  f32x4 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
  load_frags(fa0, fb0);
  for (int kt = 0; kt < nkt; kt += 2) {
    if (kt + 1 < nkt) load_frags(fa1, fb1);
    mma(fa0, fb0);
    if (kt + 1 < nkt) {
      if (kt + 2 < nkt) load_frags(fa0, fb0);
      mma(fa1, fb1);
    }
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long m = m0 + i * 16 + q * 4 + r;
      if (m >= M) continue;
      long opix;
      if (MODE == MODE_GATHER) {
        opix = m;
      } else {                                     // (32-bit: M < 2^31, see the prologue; 64-bit divisions cost ~100 instructions each)
        const unsigned um = (unsigned)m, utmp = um / (unsigned)Wa;
        const int wq = (int)(um - utmp * (unsigned)Wa);
        const unsigned ub = utmp / (unsigned)Ha;
        const int hq = (int)(utmp - ub * (unsigned)Ha);
        const long b = ub;
        opix = (b * p.Ho + 2 * hq + ph) * p.Wo + 2 * wq + pw;
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = j * 16 + lrow;
        float v = acc[i][j][r];
        if (p.bias) v += p.bias[n];
        if (p.scale) {
          v = v * p.scale[n] + p.shift[n];
          v = v > 0.f ? v : v * p.slope;
        }
        float* dst = p.y + opix * p.ldy + n;
        if (p.accumulate) v += *dst;
        *dst = v;
      }
    }
  }
}

template <int N_, int I_ = 0, class F>
__device__ __forceinline__ void svs_static_for(F&& f) {        // f(integral_constant<int, I>) for I = 0 .. N-1, unrolled
  if constexpr (I_ < N_) {
    f(std::integral_constant<int, I_>{});
    svs_static_for<N_, I_ + 1>(f);
  }
}

// ------------------------------------------------------------------------------------------------
// LDS-window variant of the PARITY mode for the shallow layers (few channels, many pixels: deconv5 forward,
// conv2 backward-data).  A block owns TH x TW anchor pixels of one image (= a 2TH x 2TW output patch, all four
// output parities), stages that window plus a one-pixel halo ONCE in LDS (zero-filled outside the image) and
// every tap of every parity class then reads its MFMA A-fragment from LDS at base + compile-time offset: no
// per-tap address arithmetic, no validity masks, no re-fetch of the input through L1/L2 (the direct kernel
// fetched 5-6x the algorithmic bytes on these layers, profiles/r01_pmc_traffic.json).  Each wave owns two anchor
// rows (two 16-pixel row tiles); the weight fragments come straight from global memory (they are shared by
// every block, so they live in L2/L1) and are prefetched one tap ahead.
// ------------------------------------------------------------------------------------------------
template <int C, int CW, int TN, int NT = TN>  // C input channels, CW of them per staging phase; N = 16 * NT, of which a block
                                               // computes 16 * TN (blockIdx.y picks them: small batches get NT / TN x the blocks)
__global__ __launch_bounds__(256) void parity_window_kernel(ConvGemmArgs p) {
  constexpr int TH = 8, TW = 16, TM = 2;
  constexpr int LP = CW + 4;                   // floats per staged pixel: 16-byte aligned, conflict-free b128 reads
  constexpr int WW = TW + 2, NPX = (TH + 2) * WW;
  constexpr int CQ = CW / 4, CC = CW / 16;
  constexpr int NST = (NPX * CQ + 255) / 256;
  constexpr unsigned OOB = 0x80000000u;
  constexpr int POFF[4] = {0, 9, 15, 21};
  __shared__ __attribute__((aligned(16))) float win[NPX * LP];
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int lrow = lane & 15, q = lane >> 4;
  const int tiles_w = (p.W + TW - 1) / TW, tiles_h = (p.H + TH - 1) / TH;
  const int tile = blockIdx.x;
  const int tw0 = (tile % tiles_w) * TW;
  const int th0 = ((tile / tiles_w) % tiles_h) * TH;
  const long b = tile / (tiles_w * tiles_h);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + b * p.H * p.W * p.ldx), 0, OOB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, OOB, 0x00020000);
  // A-fragment base of row tile i: anchor (2*wave + i, lrow), channel quarter q, at window coordinate (+1, +1)
  const float* abase = &win[((2 * wave + 1) * WW + lrow + 1) * LP + q * 4];
  auto stage_window = [&](int phase) {                             // global -> registers -> LDS, zero outside the image
    f32x4 stage[NST];
#pragma unroll
    for (int k = 0; k < NST; ++k) {
      const int e = t + k * 256;
      const int cq = e % CQ, px = e / CQ;
      const int lw = px % WW, lh = px / WW;
      const int ih = th0 - 1 + lh, iw = tw0 - 1 + lw;
      const bool ok = px < NPX && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
      const unsigned vo = ok ? (unsigned)(((ih * p.W + iw) * (int)p.ldx + phase * CW + cq * 4) * 4) : OOB;
      stage[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)vo, 0, 0));
    }
    if (phase) __syncthreads();                                    // the previous phase's readers are done
#pragma unroll
    for (int k = 0; k < NST; ++k) {
      const int e = t + k * 256;
      const int cq = e % CQ, px = e / CQ;
      if (px < NPX) *(f32x4*)(&win[px * LP + cq * 4]) = stage[k];
    }
    __syncthreads();
  };
  // The 25 taps in class order (9 + 6 + 6 + 4).  Step s = (class par, tap within class); the weight fragments of
  // step s + 1 (possibly the next class's first tap) are requested before step s's MFMAs, and a scheduling barrier
  // after every step keeps the compiler from hoisting later steps' loads on top (which costs a third of the occupancy).
  unsigned b_voff[TN];
  const int n0 = NT > TN ? (int)blockIdx.y * (TN * 16) : 0;
#pragma unroll
  for (int j = 0; j < TN; ++j) b_voff[j] = (unsigned)((n0 + j * 16 + lrow) * C * 4);   // one tap's row pitch; x ntaps of the class below
  f32x4 fb[2][CC][TN];
  auto load_b = [&](auto sc, int phase) {
    constexpr int s_ = decltype(sc)::value;
    constexpr int par = s_ < 9 ? 0 : s_ < 15 ? 1 : s_ < 21 ? 2 : 3;
    constexpr int tap = s_ - POFF[par];
    constexpr int ntaps = (3 - (par >> 1)) * (3 - (par & 1));
    // class weights start at POFF*N*C floats; row n of the class holds ntaps*C floats
#pragma unroll
    for (int cc = 0; cc < CC; ++cc)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        fb[s_ & 1][cc][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
            rw, (int)(b_voff[j] * ntaps + q * 16 + phase * (CW * 4)), (POFF[par] * (NT * 16) * C + tap * C + cc * 16) * 4, 0));
  };
  auto step = [&](auto sc, f32x4 (&acc)[TM][TN]) {
    constexpr int s_ = decltype(sc)::value;
    constexpr int par = s_ < 9 ? 0 : s_ < 15 ? 1 : s_ < 21 ? 2 : 3;
    constexpr int tap = s_ - POFF[par];
    constexpr int ntw = 3 - (par & 1);
    constexpr int th = tap / ntw, tw = tap % ntw;
    constexpr int aoff = ((1 - th) * WW + (1 - tw)) * LP;
#pragma unroll
    for (int cc = 0; cc < CC; ++cc) {
      f32x4 fa[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = *(const f32x4*)(abase + aoff + i * WW * LP + cc * 16);
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][k], fb[s_ & 1][cc][j][k], acc[i][j], 0, 0, 0);
    }
  };
  struct ColStat { float s[TN], q[TN]; };     // per-lane channel sums of what a class stores (BatchNorm statistics); returned
                                              // by value: an array captured by reference in a lambda ends up in scratch
  ColStat wst;
#pragma unroll
  for (int j = 0; j < TN; ++j) { wst.s[j] = 0.f; wst.q[j] = 0.f; }
  // rows q*4 + r of row tile i = anchor (th0 + 2*wave + i, tw0 + q*4 + r); column lrow (+16j) = output channel
  auto store_class = [&](int par, const f32x4 (&acc)[TM][TN]) -> ColStat {
    ColStat cs;
#pragma unroll
    for (int j = 0; j < TN; ++j) { cs.s[j] = 0.f; cs.q[j] = 0.f; }
    const int ph = par >> 1, pw = par & 1;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int a = th0 + 2 * wave + i;
      const int oh = 2 * a + ph;
      if (a >= p.H || oh >= p.Ho) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = tw0 + q * 4 + r;
        const int ow = 2 * c + pw;
        if (c >= p.W || ow >= p.Wo) continue;
        const long opix = (b * p.Ho + oh) * p.Wo + ow;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int n = n0 + j * 16 + lrow;
          float v = acc[i][j][r];
          if (p.bias) v += p.bias[n];
          if (p.scale) {
            v = v * p.scale[n] + p.shift[n];
            v = v > 0.f ? v : v * p.slope;
          }
          float* dst = p.y + opix * p.ldy + n;
          if (p.accumulate) v += *dst;
          *dst = v;
          cs.s[j] += v;
          cs.q[j] += v * v;
        }
      }
    }
    return cs;
  };
  auto add_stat = [](ColStat a, const ColStat& b) -> ColStat {
#pragma unroll
    for (int j = 0; j < TN; ++j) { a.s[j] += b.s[j]; a.q[j] += b.q[j]; }
    return a;
  };
  auto zero = [&](f32x4 (&acc)[TM][TN]) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  };
  if constexpr (C == CW) {                     // one phase: one class at a time, 2*TN accumulators live
    stage_window(0);
    f32x4 acc[TM][TN];
    load_b(std::integral_constant<int, 0>{}, 0);
    svs_static_for<25>([&](auto sc) {
      constexpr int s_ = decltype(sc)::value;
      constexpr int par = s_ < 9 ? 0 : s_ < 15 ? 1 : s_ < 21 ? 2 : 3;
      if constexpr (s_ == POFF[par]) zero(acc);
      if constexpr (s_ + 1 < 25) load_b(std::integral_constant<int, (s_ + 1 < 25 ? s_ + 1 : 0)>{}, 0);
      if constexpr (CC >= 4) __builtin_amdgcn_sched_barrier(0);
      step(sc, acc);
      if constexpr (s_ == 24 || s_ + 1 == POFF[par < 3 ? par + 1 : 3]) wst = add_stat(wst, store_class(par, acc));
      if constexpr (CC >= 4) __builtin_amdgcn_sched_barrier(0);     // (32-channel steps are too short to fence)
    });
  } else {                                     // several phases share the window buffer: all four classes stay live
    f32x4 acc[4][TM][TN];
#pragma unroll
    for (int par = 0; par < 4; ++par) zero(acc[par]);
#pragma unroll 1
    for (int phase = 0; phase < C / CW; ++phase) {
      stage_window(phase);
      load_b(std::integral_constant<int, 0>{}, phase);
      svs_static_for<25>([&](auto sc) {
        constexpr int s_ = decltype(sc)::value;
        constexpr int par = s_ < 9 ? 0 : s_ < 15 ? 1 : s_ < 21 ? 2 : 3;
        if constexpr (s_ + 1 < 25) load_b(std::integral_constant<int, (s_ + 1 < 25 ? s_ + 1 : 0)>{}, phase);
        __builtin_amdgcn_sched_barrier(0);
        step(sc, acc[par]);
        __builtin_amdgcn_sched_barrier(0);
      });
    }
#pragma unroll
    for (int par = 0; par < 4; ++par) wst = add_stat(wst, store_class(par, acc[par]));
  }
  if (p.stats) {                               // this block's row of BatchNorm partials: [2][N]
    __shared__ float st[2][4][TN * 16];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float a = wst.s[j], b2 = wst.q[j];
      a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
      b2 += __shfl_xor(b2, 16, 64); b2 += __shfl_xor(b2, 32, 64);
      if (q == 0) { st[0][wave][j * 16 + lrow] = a; st[1][wave][j * 16 + lrow] = b2; }
    }
    __syncthreads();
    if (t < TN * 16) {
      float* out = p.stats + (long)blockIdx.x * 2 * (NT * 16) + n0;
      out[t] = (st[0][0][t] + st[0][1][t]) + (st[0][2][t] + st[0][3][t]);
      out[NT * 16 + t] = (st[1][0][t] + st[1][1][t]) + (st[1][2][t] + st[1][3][t]);
    }
  }
}

// out[pix][n] = epi(sum_z slab[z][pix][n] + bias[n]).  With `stats` the block also leaves the per-channel sum and
// sum of squares of the values it wrote in stats[blk][2][N] (the layout of svs_bn_stats' partials), so the
// BatchNorm statistics of a split-K layer need no pass of their own over the output (requires 256 % (N/4) == 0:
// a thread then keeps the same four channels for the whole grid-stride loop).
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const float* __restrict__ slab, int ksplit, long P, int N,
                                                              const float* __restrict__ bias,
                                                              const float* __restrict__ scale,
                                                              const float* __restrict__ shift, float slope,
                                                              float* y, long ldy, int accumulate, float* __restrict__ stats,
                                                              int n_shift,        // log2(N) when N is a power of two (every layer here), else -1
                                                              const unsigned char* __restrict__ rowsplit) {   // balanced splits: slabs per row, else null
  __shared__ f32x4 red[2][256];
  const long total4 = P * N / 4;
  const long stride = P * N;
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const long e = i * 4;
    const long pix = n_shift >= 0 ? (e >> n_shift) : e / N;         // (a 64-bit division per float4 otherwise)
    const int n = (int)(e - pix * N);
    f32x4 s = *(const f32x4*)(slab + e);
    const int nz = rowsplit ? (int)rowsplit[pix] : ksplit;                              // (ConvBal: a row has as many slabs as its position has splits)
#pragma unroll 4
    for (int z = 1; z < nz; ++z) s += *(const f32x4*)(slab + z * stride + e);         // (unrolled: four slabs' loads in flight, same order)
    float* dst = y + pix * ldy + n;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float v = s[k];
      if (bias) v += bias[n + k];
      if (scale) {
        v = v * scale[n + k] + shift[n + k];
        v = v > 0.f ? v : v * slope;
      }
      if (accumulate) v += dst[k];
      s[k] = v;
    }
    *(f32x4*)dst = s;
    s0 += s;
    s1 += s * s;
  }
  if (!stats) return;
  const int G = N >> 2, PL = 256 / G, t = threadIdx.x;
  red[0][t] = s0;
  red[1][t] = s1;
  __syncthreads();
  if (t < G) {
    f32x4 a = red[0][t], b = red[1][t];
    for (int j = 1; j < PL; ++j) { a += red[0][j * G + t]; b += red[1][j * G + t]; }
    float* out = stats + (long)blockIdx.x * 2 * N;
    *(f32x4*)(out + t * 4) = a;
    *(f32x4*)(out + N + t * 4) = b;
  }
}

// The same sum for a layer whose BatchNorm statistics are wanted, in the slab form of elementwise.hip's RedPlan: a block owns
// 32 adjacent channels (one 128-byte line per pixel and slab) and a range of `ppr` pixels, block -> (slab, range) with the slab
// innermost, and leaves stats[range][2][N] -- as many partial rows per channel as there are pixel ranges (8-64 on the deep
// levels that run split-K), few enough for the BatchNorm apply kernel to fold them itself.  N % 32 == 0.
__global__ __launch_bounds__(256) void splitk_epilogue_stats_kernel(const float* __restrict__ slab, int ksplit, long P, int N,
                                                                    const float* __restrict__ bias, float* __restrict__ y, long ldy,
                                                                    float* __restrict__ stats, long ppr,
                                                                    const unsigned char* __restrict__ rowsplit) {
  __shared__ f32x4 red[2][256];
  const int nslab = N >> 5, t = threadIdx.x;
  const int sl = blockIdx.x % nslab, row = blockIdx.x / nslab;
  const int l = t & 7, pr = t >> 3;
  const int n = sl * 32 + l * 4;
  const long stride = P * N;
  const long p0 = (long)row * ppr;
  long p1 = p0 + ppr;
  if (p1 > P) p1 = P;
  f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
  if (bias) b4 = *(const f32x4*)(bias + n);
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
  for (long pix = p0 + pr; pix < p1; pix += 32) {
    const long e = pix * N + n;
    f32x4 s = *(const f32x4*)(slab + e);
    const int nz = rowsplit ? (int)rowsplit[pix] : ksplit;
#pragma unroll 4
    for (int z = 1; z < nz; ++z) s += *(const f32x4*)(slab + z * stride + e);         // (same order as splitk_epilogue_kernel)
    if (bias) s += b4;
    *(f32x4*)(y + pix * ldy + n) = s;
    s0 += s;
    s1 += s * s;
  }
  red[0][t] = s0;
  red[1][t] = s1;
  __syncthreads();
  if (t < 8) {
    f32x4 a = red[0][t], b = red[1][t];
    for (int j = 1; j < 32; ++j) { a += red[0][j * 8 + t]; b += red[1][j * 8 + t]; }
    float* out = stats + (long)row * 2 * N + n;
    *(f32x4*)out = a;
    *(f32x4*)(out + N) = b;
  }
}

// ------------------------------------------------------------------------------------------------
// LDS-window form of the GATHER mode for conv2 forward (16 -> 32 channels, stride 2).  On the GEMM kernel every input pixel goes
// through L2 / LDS 25/4 times with 256 x 32 tiles (PMC: 271 MB per launch at batch 64 for 100 MB of input + output, 78.6 us:
// the least efficient forward GEMM).  Here a block owns 8 x 16 OUTPUT pixels of one image, stages their 19 x 35 input window
// once (even / odd columns in separate planes, so that the 16 pixels of a fragment -- two input columns apart -- are adjacent
// slots; 80-byte slot pitch: conflict-free ds_read_b128), and every tap is 16 MFMAs per wave with the pixel fragments at base +
// compile-time offset.  Waves (nt, rh): channel tile nt (16 of the 32), output rows 4 rh .. 4 rh + 3.  Weights are the FIRST
// MFMA operand (D = [channel][pixel]: a lane's four registers are four consecutive channels of one pixel -- one 16-byte store);
// their fragments come from the gather-packed matrix [n][25][16] in L2, one tap ahead.  K order inside a tap: step s of lane
// group q is channel 4 q + s, so a fragment is one 16-byte read.  A block walks `tpb` consecutive tiles and leaves ONE row of
// BatchNorm partials.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_window_kernel(ConvGemmArgs p, int ntiles, int tpb) {
  constexpr int TH = 8, TW = 16, WR = 2 * TH + 3, WC = 2 * TW + 3, PW = TW + 2, LP = 20;       // LP: floats per slot
  constexpr int NCH = WR * WC * 4, NST = (NCH + 255) / 256;                                    // 16-byte pieces of the window
  constexpr unsigned OOB = 0x80000000u;
  __shared__ __attribute__((aligned(16))) float win[WR * 2 * PW * LP];
  __shared__ float st[2][2][32];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, lrow = lane & 15, q = lane >> 4;
  const int nt = wave >> 1, rh = wave & 1;
  const int tiles_w = (p.Wo + TW - 1) / TW, tiles_h = (p.Ho + TH - 1) / TH;
  const int n = nt * 16 + 4 * q;                                     // this lane's four output channels
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, OOB, 0x00020000);
  const unsigned w_voff = (unsigned)(((nt * 16 + lrow) * 25 * 16 + q * 4) * 4);                // row n = nt*16 + lrow of the weight matrix
  f32x4 b4 = {0.f, 0.f, 0.f, 0.f}, sc4 = b4, sh4 = b4;
  if (p.bias) b4 = *(const f32x4*)(p.bias + n);
  if (p.scale) { sc4 = *(const f32x4*)(p.scale + n); sh4 = *(const f32x4*)(p.shift + n); }
  const float* const pbase = &win[((16 * rh) * PW + lrow) * LP + q * 4];                       // window row 8 rh (+ 2 i + kh), plane 0, slot lrow
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
  const int tile_lo = blockIdx.x * tpb, tile_hi = tile_lo + tpb < ntiles ? tile_lo + tpb : ntiles;
  for (int tile = tile_lo; tile < tile_hi; ++tile) {
    const int ow0 = (tile % tiles_w) * TW, oh0 = ((tile / tiles_w) % tiles_h) * TH;
    const long b = tile / (tiles_w * tiles_h);
    // (the window of the NEXT tile in registers during this tile's MFMAs, as the bf16 window kernels have it, measured slower here:
    // 76.8 against 72.8 us at batch 64 -- 44 more registers; the second resident block covers the load latency)
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + b * p.H * p.W * p.ldx), 0, OOB, 0x00020000);
    f32x4 stage[NST];
#pragma unroll
    for (int k = 0; k < NST; ++k) {
      const int e = t + k * 256, c4 = e & 3, px = e >> 2;
      const int wr = px / WC, wc = px - wr * WC;
      const int ih = 2 * oh0 - 2 + wr, iw = 2 * ow0 - 2 + wc;
      const bool ok = e < NCH && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
      const unsigned vo = ok ? (unsigned)(((ih * p.W + iw) * (int)p.ldx + c4 * 4) * 4) : OOB;
      stage[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)vo, 0, 0));
    }
    __syncthreads();                                                 // the previous tile's readers are done
#pragma unroll
    for (int k = 0; k < NST; ++k) {
      const int e = t + k * 256, c4 = e & 3, px = e >> 2;
      const int wr = px / WC, wc = px - wr * WC;
      if (e < NCH) *(f32x4*)(&win[((wr * 2 + (wc & 1)) * PW + (wc >> 1)) * LP + c4 * 4]) = stage[k];
    }
    __syncthreads();
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 fw[2];
    fw[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, (int)w_voff, 0, 0));
    svs_static_for<25>([&](auto tc) {
      constexpr int tap = decltype(tc)::value, kh = tap / 5, kw = tap % 5;
      if constexpr (tap + 1 < 25) fw[(tap + 1) & 1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, (int)w_voff, (tap + 1) * 16 * 4, 0));
      f32x4 fp[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fp[i] = *(const f32x4*)(pbase + (((2 * i + kh) * 2 + (kw & 1)) * PW + (kw >> 1)) * LP);
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(fw[tap & 1][k], fp[i][k], acc[i], 0, 0, 0);
    });
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int oh = oh0 + 4 * rh + i, ow = ow0 + lrow;
      if (oh >= p.Ho || ow >= p.Wo) continue;
      f32x4 v = acc[i];
      if (p.bias) v += b4;
      if (p.scale) {
        v = v * sc4 + sh4;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = v[k] > 0.f ? v[k] : v[k] * p.slope;
      }
      *(f32x4*)(p.y + ((b * p.Ho + oh) * p.Wo + ow) * p.ldy + n) = v;
      s0 += v;
      s1 += v * v;
    }
  }
  if (p.stats) {                                 // this block's row of BatchNorm partials [2][32]: over the 16 pixels of a row group, then the two row halves
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) { s0[k] += __shfl_xor(s0[k], o, 64); s1[k] += __shfl_xor(s1[k], o, 64); }
    if (lrow == 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) { st[0][rh][n + k] = s0[k]; st[1][rh][n + k] = s1[k]; }
    }
    __syncthreads();
    if (t < 64) {
      const int which = t >> 5, c = t & 31;
      p.stats[(long)blockIdx.x * 64 + which * 32 + c] = st[which][0][c] + st[which][1][c];
    }
  }
}

// ---- host side ---------------------------------------------------------------------------------
struct ConvPlan { int cfg; int BM, BN; int ksplit; long mtiles; int grid_y; int pf; };

// `inference`: the call has the folded-BatchNorm epilogue (eval forward).  Those calls run at the serving batch sizes (1..16
// tiles, BASELINE configs[0..1]) and take the rules of the batch-16 sweep; the training calls keep the batch-64 table (same-device
// A/B: the batch-16 rules cost the batch-64 train step 0.6 %, and gain the batch-16 forward 3.6 %).
static ConvPlan plan_conv(int mode, long Mmax, int N, int nkt_min, bool narrow = false, bool inference = false) {
  ConvPlan pl{};
  const bool v2 = inference && svs_tune(SVS_TUNE_CONV_PLAN) != 0;
  if (N % 128 == 0) {
    if (Mmax <= 96) { pl.cfg = 4; pl.BM = 32; pl.BN = 128; }
    else { pl.cfg = 0; pl.BM = 128; pl.BN = 128; }
  } else if (N == 64) {
    if (Mmax <= 512 || nkt_min <= 25 || (v2 && Mmax <= 16384)) { pl.cfg = 5; pl.BM = 64; pl.BN = 64; }      // (16 input channels: deconv5 backward-data, -9 % in the sweep; batch 16: conv3 forward 33 -> 27 us)
    else { pl.cfg = 1; pl.BM = 128; pl.BN = 64; }
  } else if (N == 32) { pl.cfg = 2; pl.BM = 256; pl.BN = 32; }
  else { pl.cfg = 3; pl.BM = 256; pl.BN = 16; }
  // Parity blocks differ in length (9/6/6/4 taps): many small tiles balance better than few big ones
  // (tools/gemm_sweep.py, B=64: 64x64 tiles beat 128-wide ones by 10-20% whenever M per parity > 1024).
  if (mode == MODE_PARITY && N % 64 == 0 && Mmax > 1024) { pl.cfg = 5; pl.BM = 64; pl.BN = 64; }
  // Narrow (deep) levels with batch-innermost rows: a 64-row tile is one pixel position of 64 images, so the padding
  // skip is exact per position, and 4x the tiles need a quarter of the K-splits (same-device sweep at B=64: 64x64
  // beats 128x128 on every such layer, by 20 % on the 8x2 parity layers).
  if (narrow && N % 64 == 0) { pl.cfg = 5; pl.BM = 64; pl.BN = 64; }
  if (narrow && N == 128 && !(v2 && Mmax < (mode == MODE_PARITY ? 2048 : 8192))) { pl.cfg = 6; pl.BM = 64; pl.BN = 128; }     // (batch 64: deconv2 forward -9 %, conv4 fwd / conv5 bwd-data -2 %; batch 16: 64x64 is 10-15 % ahead)
  // conv5 forward (16x4 anchors, 128 -> 256 channels) with balanced K-splits (ConvBal): the 64x128 tile at 3 blocks per CU is
  // 13 % ahead of 64x64 in either form (68 vs 78 us at batch 64)
  if (narrow && mode == MODE_GATHER && N == 256 && nkt_min == 200 && Mmax >= 4096 && Mmax % 64 == 0 && !v2 && svs_tune(SVS_TUNE_CONV_BALANCE) != 0) { pl.cfg = 6; pl.BM = 64; pl.BN = 128; }
  // Optional split-bf16 product mode (mfma_split.h): the MFMA part of a K-tile is 2.7x shorter, the per-fragment limb split is
  // paid once per (row tile + column tile) of a wave, so LARGER wave tiles win (batch-64 sweep in that mode: 128x128 / 64x128
  // ahead of 64x64 by 8-30 % on every N >= 128 layer, 128x64 on the N = 64 ones), at two blocks per CU
  const bool split = svs_tune(SVS_TUNE_MFMA_SPLIT) > 0 && svs_tune(SVS_TUNE_CONV_PLAN) != 0;
  if (split && Mmax > 96) {
    if (N % 128 == 0) {
      if (mode == MODE_GATHER && Mmax >= 4096) { pl.cfg = 0; pl.BM = 128; pl.BN = 128; }
      else { pl.cfg = 6; pl.BM = 64; pl.BN = 128; }
    } else if (N == 64 && Mmax >= 16384) { pl.cfg = 1; pl.BM = 128; pl.BN = 64; }
  }
  if (svs_tune_on(SVS_TUNE_CONV_CFG)) {      // sweeps only
    static const int bm[7] = {128, 128, 256, 256, 32, 64, 64}, bn[7] = {128, 64, 32, 16, 128, 64, 128};
    const int c = (int)svs_tune(SVS_TUNE_CONV_CFG);
    if (c >= 0 && c < 7 && N % bn[c] == 0) { pl.cfg = c; pl.BM = bm[c]; pl.BN = bn[c]; }
  }
  pl.mtiles = (Mmax + pl.BM - 1) / pl.BM;
  pl.grid_y = (mode == MODE_PARITY) ? 4 : 1;
  const long blocks = pl.mtiles * (N / pl.BN) * pl.grid_y;
  int ks = 1;
  const long target = (mode == MODE_PARITY) ? 1024 : ((v2 || split) ? 512 : 768);     // (batch-16 sweep: two gather blocks per CU beat three on every layer)
  if (blocks < ((mode == MODE_PARITY) ? 768 : 384)) {
    ks = (int)((target + blocks - 1) / blocks);
    // keep >= 8 K-tiles per split (16 in gather mode: with K = 25 or 50 tiles -- conv2 / conv3 at batch 16 -- finer
    // splits lose more in the epilogue than they win in occupancy)
    const int per = (mode == MODE_GATHER) ? 16 : 8;
    const int cap = nkt_min / per > 1 ? nkt_min / per : 1;
    if (ks > cap) ks = cap;
    if (ks > 64) ks = 64;
    if (ks < 1) ks = 1;
  }
  if (svs_tune_on(SVS_TUNE_CONV_KSPLIT)) { int f = (int)svs_tune(SVS_TUNE_CONV_KSPLIT); if (f >= 1 && f <= nkt_min) ks = f; }
  pl.ksplit = ks;
  // K-tiles requested ahead: two on the 64-row tiles (same-device A/B of tools/ab_tune.py CONV_PF 1 2: train step at batch 64
  // 3.530 -> 3.454 ms, eval forward 0.388 -> 0.379 ms at batch 16 and 1.030 -> 0.994 ms at batch 64: with 60-100 registers these
  // tiles keep their occupancy, and one tile time -- 512 MFMA cycles -- is less than an L2 / Infinity-Cache round trip under load)
  // (the 128x64 / 256x32 / 256x16 tiles: another -6 us on the train step, CONV_PF 2 against 3; the 128x128 and 32x128 tiles keep
  //  one tile ahead: at 164 registers a second set costs the 128x128 tile its third resident block)
  pl.pf = (pl.cfg == 5 || pl.cfg == 6 || (pl.cfg >= 1 && pl.cfg <= 3)) ? 2 : 1;
  // CONV_PF (A/B runs): 1 = one tile ahead everywhere; 2 = two ahead on the 64-row tiles only; 3 = on every tile that has the variant
  if (svs_tune_on(SVS_TUNE_CONV_PF)) {
    const long f = svs_tune(SVS_TUNE_CONV_PF);
    pl.pf = (f >= 2 && (pl.cfg == 5 || pl.cfg == 6)) || (f == 3 && pl.cfg >= 1 && pl.cfg <= 3) ? 2 : 1;
  }
  return pl;
}

template <int MODE, bool SPLIT>
static void launch_conv_gemm_cfg(const ConvGemmArgs& a, const ConvPlan& pl, dim3 grid, hipStream_t stream, bool skip, const ConvBal& bal) {
  dim3 block(256);
  if (skip) {                                // batch-innermost rows + padding-tap skipping (deep levels)
    switch (pl.cfg) {
      case 0: hipLaunchKernelGGL((conv_gemm_kernel<MODE, 128, 128, 2, 2, true, SPLIT>), grid, block, 0, stream, a, bal); break;
      case 1: if (pl.pf == 2) hipLaunchKernelGGL((conv_gemm_kernel<MODE, 128, 64, 2, 2, true, SPLIT, 2>), grid, block, 0, stream, a, bal);
              else hipLaunchKernelGGL((conv_gemm_kernel<MODE, 128, 64, 2, 2, true, SPLIT>), grid, block, 0, stream, a, bal);
              break;
      case 4: hipLaunchKernelGGL((conv_gemm_kernel<MODE, 32, 128, 1, 4, true, SPLIT>), grid, block, 0, stream, a, bal); break;
      case 6: if (pl.pf == 2) hipLaunchKernelGGL((conv_gemm_kernel<MODE, 64, 128, 2, 2, true, SPLIT, 2>), grid, block, 0, stream, a, bal);
              else hipLaunchKernelGGL((conv_gemm_kernel<MODE, 64, 128, 2, 2, true, SPLIT>), grid, block, 0, stream, a, bal);
              break;
      default: if (pl.pf == 2) hipLaunchKernelGGL((conv_gemm_kernel<MODE, 64, 64, 2, 2, true, SPLIT, 2>), grid, block, 0, stream, a, bal);
               else hipLaunchKernelGGL((conv_gemm_kernel<MODE, 64, 64, 2, 2, true, SPLIT>), grid, block, 0, stream, a, bal);
               break;
    }
    return;
  }
  const ConvNoBal nb{};
  switch (pl.cfg) {
    case 0: hipLaunchKernelGGL((conv_gemm_kernel<MODE, 128, 128, 2, 2, false, SPLIT>), grid, block, 0, stream, a, nb); break;
    case 1: if (pl.pf == 2) hipLaunchKernelGGL((conv_gemm_kernel<MODE, 128, 64, 2, 2, false, SPLIT, 2>), grid, block, 0, stream, a, nb);
            else hipLaunchKernelGGL((conv_gemm_kernel<MODE, 128, 64, 2, 2, false, SPLIT>), grid, block, 0, stream, a, nb);
            break;
    case 2: if (pl.pf == 2) hipLaunchKernelGGL((conv_gemm_kernel<MODE, 256, 32, 4, 1, false, SPLIT, 2>), grid, block, 0, stream, a, nb);
            else hipLaunchKernelGGL((conv_gemm_kernel<MODE, 256, 32, 4, 1, false, SPLIT>), grid, block, 0, stream, a, nb);
            break;
    case 3: if (pl.pf == 2) hipLaunchKernelGGL((conv_gemm_kernel<MODE, 256, 16, 4, 1, false, SPLIT, 2>), grid, block, 0, stream, a, nb);
            else hipLaunchKernelGGL((conv_gemm_kernel<MODE, 256, 16, 4, 1, false, SPLIT>), grid, block, 0, stream, a, nb);
            break;
    case 4: hipLaunchKernelGGL((conv_gemm_kernel<MODE, 32, 128, 1, 4, false, SPLIT>), grid, block, 0, stream, a, nb); break;
    case 6: if (pl.pf == 2) hipLaunchKernelGGL((conv_gemm_kernel<MODE, 64, 128, 2, 2, false, SPLIT, 2>), grid, block, 0, stream, a, nb);
            else hipLaunchKernelGGL((conv_gemm_kernel<MODE, 64, 128, 2, 2, false, SPLIT>), grid, block, 0, stream, a, nb);
            break;
    default: if (pl.pf == 2) hipLaunchKernelGGL((conv_gemm_kernel<MODE, 64, 64, 2, 2, false, SPLIT, 2>), grid, block, 0, stream, a, nb);
             else hipLaunchKernelGGL((conv_gemm_kernel<MODE, 64, 64, 2, 2, false, SPLIT>), grid, block, 0, stream, a, nb);
             break;
  }
}
template <int MODE>
static int launch_conv_gemm(const ConvGemmArgs& a, const ConvPlan& pl, hipStream_t stream, bool skip, const ConvBal& bal) {
  dim3 grid((unsigned)(pl.mtiles * (a.N / pl.BN)), (unsigned)pl.ksplit, (unsigned)pl.grid_y);      // (tiles, K-splits, parity classes)
  if (skip && bal.enabled) grid = dim3(bal.first[pl.grid_y - 1][bal.npos[pl.grid_y - 1]], 1, 1);     // (class, position, M-tile, split, N-tile)
  if (svs_tune(SVS_TUNE_MFMA_SPLIT) > 0) launch_conv_gemm_cfg<MODE, true>(a, pl, grid, stream, skip, bal);     // optional mode: mfma_split.h
  else launch_conv_gemm_cfg<MODE, false>(a, pl, grid, stream, skip, bal);
  SVS_CHECK_LAUNCH("conv_gemm");
  return SVS_OK;
}

// Host side of ConvBal (see the struct): per-position split counts for a tap-skipping launch.  Returns the number of slabs
// (the largest split count), 0 = not applicable / not worth it -> uniform grid.  Purely a function of the shape, so the
// workspace query and the launch agree.
static int plan_balance_search(int mode, int B, int H, int W, int C, int Ho, int Wo, int N, const ConvPlan& pl, ConvBal* out);
// memoised: the search walks a few hundred candidate plans of ~1000 blocks each (0.2-0.7 ms of host time -- per LAUNCH it would
// cost more than the kernel it plans); a handful of distinct shapes per process
static int plan_balance(int mode, int B, int H, int W, int C, int Ho, int Wo, int N, const ConvPlan& pl, ConvBal* out) {
  struct Key { int v[12]; bool operator==(const Key& o) const { return !memcmp(v, o.v, sizeof(v)); } };
  struct Entry { Key k; int slabs; ConvBal bal; };
  static std::mutex mu;
  static std::vector<Entry> cache;
  const Key key{{mode, B, H, W, C, Ho, Wo, N, pl.BM, pl.BN, pl.ksplit, (int)svs_tune(SVS_TUNE_CONV_BALANCE)}};
  std::lock_guard<std::mutex> lock(mu);
  for (const Entry& e : cache)
    if (e.k == key) { if (out) { unsigned char* rs = out->rowsplit; *out = e.bal; out->rowsplit = rs; } return e.slabs; }
  Entry e{};
  e.k = key;
  e.slabs = plan_balance_search(mode, B, H, W, C, Ho, Wo, N, pl, &e.bal);
  if (cache.size() > 256) cache.clear();
  cache.push_back(e);
  if (out) { unsigned char* rs = out->rowsplit; *out = e.bal; out->rowsplit = rs; }
  return e.slabs;
}
static int plan_balance_search(int mode, int B, int H, int W, int C, int Ho, int Wo, int N, const ConvPlan& pl, ConvBal* out) {
  if (out) out->enabled = 0;
  if (svs_tune(SVS_TUNE_CONV_BALANCE) == 0 || pl.ksplit <= 1 || B % pl.BM != 0 || (C & (C - 1)) != 0) return 0;
  const int ncls = (mode == MODE_PARITY) ? 4 : 1, cpt = C / 16, r = B / pl.BM, ntn = N / pl.BN;
  int npos[4] = {0, 0, 0, 0}, nvk[4][64];
  long total = 0;
  for (int c = 0; c < ncls; ++c) {
    const int ph = c >> 1, pw = c & 1;
    const int Ha = (mode == MODE_PARITY) ? (Ho - ph + 1) / 2 : Ho, Wa = (mode == MODE_PARITY) ? (Wo - pw + 1) / 2 : Wo;
    const int nth = (mode == MODE_PARITY) ? 3 - ph : 5, ntw = (mode == MODE_PARITY) ? 3 - pw : 5;
    if (Ha * Wa > 64 || Ha * Wa < 1) return 0;
    npos[c] = Ha * Wa;
    for (int pos = 0; pos < npos[c]; ++pos) {                     // rows are ordered (w, h, b): pos = wq * Ha + hq
      const int wq = pos / Ha, hq = pos - wq * Ha;
      int vh = 0, vw = 0;
      for (int th = 0; th < nth; ++th) { const int ih = (mode == MODE_GATHER) ? 2 * hq - 2 + th : hq + 1 - th; vh += (ih >= 0 && ih < H); }
      for (int tw = 0; tw < ntw; ++tw) { const int iw = (mode == MODE_GATHER) ? 2 * wq - 2 + tw : wq + 1 - tw; vw += (iw >= 0 && iw < W); }
      nvk[c][pos] = vh * vw * cpt;
      total += (long)nvk[c][pos] * r * ntn;
    }
  }
  // candidates: target work per block T (in K-tiles) and a cap on the splits; cost = load of the most loaded CU when the
  // blocks go to the 256 CUs round-robin in launch order (how the dispatcher is observed to place a grid that is resident at
  // once; only speed depends on it), each block paying a fixed prologue / epilogue share, plus the epilogue's slab traffic
  const int NCU = 256;
  // (units: one K-tile of one block = 4 * TM * TN MFMAs per wave; a 64x64 tile's is 512 cycles.  Fixed cost per block ~ 3 such;
  //  a slab tile's write + read in the epilogue ~ 0.03 of them chip-wide, whatever the tile: bytes and unit both scale with it)
  const double OVH = 3.0 * 4096.0 / (pl.BM * pl.BN), SLAB = 0.03;
  const int smax_hi = pl.ksplit * 2 < 16 ? pl.ksplit * 2 : 16;
  double best = 1e30;
  int best_cap = 0; double best_T = 0, best_load = 0;
  static thread_local double load[256];
  // Same-device sweep at batch 64 (tools/gemm_sweep.py --ab-env CONV_BALANCE, targets 512 .. 1536): multiples of 256 blocks win
  // (the round-robin model), 3 per CU for the GATHER direction and 5 per CU for the PARITY one; the 8x2-anchor layers gain
  // 10-15 % (conv6 / deconv1, both directions), conv5 forward 13 % with the 64x128 tile, the 16x4-anchor PARITY layers LOSE
  // 5-10 % (their uniform grid is 256 blocks per parity class, which round-robin already spreads evenly): rule below.
  int max_pos = 0;
  for (int c = 0; c < ncls; ++c) max_pos = npos[c] > max_pos ? npos[c] : max_pos;
  const bool tuned = max_pos <= 16 || (mode == MODE_GATHER && max_pos <= 64 && pl.BN == 128);
  long force_nb = svs_tune(SVS_TUNE_CONV_BALANCE) >= 16 ? svs_tune(SVS_TUNE_CONV_BALANCE) : 0;     // sweeps: aim at this many blocks
  if (!force_nb && svs_tune(SVS_TUNE_CONV_BALANCE) != 3) {      // (3: the cost-model search below, for comparison)
    if (!tuned) return 0;
    force_nb = mode == MODE_GATHER ? 768 : 1280;
  }
  const int caps[2] = {force_nb ? 16 : pl.ksplit, force_nb ? 16 : smax_hi};
  for (int ci = 0; ci < 2; ++ci) {
    const int cap = caps[ci];
    if (ci == 1 && cap == caps[0]) break;
    // (block counts around the uniform plan's: that count -- 3 to 4 resident blocks per CU -- came out of the tile / split sweeps,
    //  and the model knows nothing about latency hiding)
    const long nb_uniform = pl.mtiles * ntn * pl.grid_y * pl.ksplit;
    // a target count: the largest plan that does NOT exceed it (one block too many puts a whole extra block on some CUs);
    // no target: every size around the uniform plan's, by the cost model
    const long nb_lo = force_nb ? force_nb / 2 : nb_uniform * 7 / 8, nb_hi = force_nb ? force_nb : nb_uniform * 11 / 8;
    for (long nb = nb_hi; nb >= nb_lo; nb -= 4) {
      const double T = (double)total / nb;
      for (int i = 0; i < NCU; ++i) load[i] = 0.0;
      long idx = 0, slabs = 0;
      for (int c = 0; c < ncls; ++c)
        for (int pos = 0; pos < npos[c]; ++pos) {
          int sp = (int)(nvk[c][pos] / T + 0.5);
          sp = sp < 1 ? 1 : sp > cap ? cap : sp;
          if (sp > nvk[c][pos]) sp = nvk[c][pos];
          slabs += (long)sp * r * ntn;
          for (int i = 0; i < r; ++i)
            for (int j = 0; j < sp; ++j) {
              const double w = (double)((long)nvk[c][pos] * (j + 1) / sp - (long)nvk[c][pos] * j / sp) + OVH;
              for (int n = 0; n < ntn; ++n) load[idx++ % NCU] += w;
            }
        }
      double mx = 0.0;
      for (int i = 0; i < NCU; ++i) mx = load[i] > mx ? load[i] : mx;
      const double cost = mx + SLAB * (double)slabs;
      if (force_nb) {
        if (idx <= force_nb) { best = cost; best_cap = cap; best_T = T; best_load = mx; break; }
        continue;
      }
      if (cost < best) { best = cost; best_cap = cap; best_T = T; best_load = mx; }
    }
  }
  if (best_cap == 0) return 0;
  // coarse cases (few splits per position, e.g. batch 128 with two M-tiles per position) do not come out even: keep the
  // uniform grid unless the modelled load of the most loaded CU is within 12 % of the mean (sweeps with a forced target excepted)
  {
    long nblk = 0;
    for (int c = 0; c < ncls; ++c)
      for (int pos = 0; pos < npos[c]; ++pos) {
        int sp = (int)(nvk[c][pos] / best_T + 0.5);
        sp = sp < 1 ? 1 : sp > best_cap ? best_cap : sp;
        if (sp > nvk[c][pos]) sp = nvk[c][pos];
        nblk += (long)sp * r * ntn;
      }
    const double mean = ((double)total + OVH * (double)nblk) / NCU;
    if (svs_tune(SVS_TUNE_CONV_BALANCE) < 16 && best_load > 1.12 * mean) return 0;
  }
  int S = 1;
  unsigned at = 0;
  for (int c = 0; c < 4; ++c) {
    if (out) out->npos[c] = npos[c];
    for (int pos = 0; pos < npos[c]; ++pos) {
      int sp = (int)(nvk[c][pos] / best_T + 0.5);
      sp = sp < 1 ? 1 : sp > best_cap ? best_cap : sp;
      if (sp > nvk[c][pos]) sp = nvk[c][pos];
      S = sp > S ? sp : S;
      if (out) { out->first[c][pos] = at; out->nsplit[c][pos] = (unsigned char)sp; }
      at += (unsigned)(sp * r * ntn);
    }
    if (out) out->first[c][npos[c]] = at;
    if (out && c + 1 < 4 && c + 1 >= ncls) { out->first[c + 1][0] = at; }
  }
  if (S <= 1) return 0;                       // (a layer that needs no slabs stays on its direct epilogue)
  if (out) out->enabled = 1;
  if (out && svs_tune(SVS_TUNE_CONV_BALANCE) == 2)      // sweeps: show the plan
    fprintf(stderr, "[svs] balanced splits mode %d B %d in %dx%dx%d N %d tile %dx%d: uniform ks %d -> slabs %d, %u blocks (%.2f per CU), "
            "%.1f K-tiles per block, most loaded CU %.0f (+ slabs: %.0f) vs mean %.0f\n", mode, B, H, W, C, N, pl.BM, pl.BN, pl.ksplit, S, at,
            at / 256.0, best_T, best_load, best, (double)total / NCU);
  return S;
}

static int check_gemm_args(const char* who, const float* x, long ldx, int B, int H, int W, int C, const float* wp,
                           float* y, long ldy, int Ho, int Wo, int N) {
  SVS_REQUIRE(x && wp && y, "%s: null pointer", who);
  SVS_REQUIRE(B > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, "%s: bad geometry B=%d H=%d W=%d", who, B, H, W);
  SVS_REQUIRE(C >= 16 && C % 16 == 0, "%s: C=%d must be a multiple of 16", who, C);
  SVS_REQUIRE(N >= 16 && (N == 16 || N == 32 || N == 64 || N % 128 == 0), "%s: unsupported N=%d", who, N);
  SVS_REQUIRE(ldx >= C && ldx % 4 == 0 && ldy >= N && ldy % 4 == 0, "%s: bad ld (ldx=%ld ldy=%ld)", who, ldx, ldy);
  SVS_REQUIRE(svs_aligned16(x) && svs_aligned16(wp) && svs_aligned16(y), "%s: pointers must be 16-byte aligned", who);
  return SVS_OK;
}

// LDS-window kernel for the shallow parity layers (N = 16 / 32, C = 32 / 64 / 128, large images)
static int use_parity_window(int mode, int B, int H, int W, int C, int N, long ldx) {
  const bool eligible = mode == MODE_PARITY && (N == 16 || N == 32) && (C == 32 || C == 64 || C == 128) &&
                        ((long)H * W * ldx) * 4 < (1L << 31);
  const bool fills_gpu = H >= 8 && W >= 16 && (long)B * ((H + 7) / 8) * ((W + 15) / 16) >= 128;   // (B=16 sweep: still ahead of the direct kernel at 128 blocks)
  int window = eligible && fills_gpu;
  if (svs_tune_on(SVS_TUNE_CONV_WINDOW)) {     // sweeps and tests: 0 = never, 2 = whenever the shape is eligible, 3 = same and
    const int f = (int)svs_tune(SVS_TUNE_CONV_WINDOW);          // never with the channel halves in separate blocks
    window = (f == 0) ? 0 : (f == 2 || f == 3) ? eligible : window;
  }
  return window;
}

// 32 output channels and at most one tile per CU (batch <= 32): the two 16-channel halves go to different blocks, each
// staging the window itself (same-device A/B of the eval forward: batch 16 -7 %, batch 32 -1 %, batch 64 +2 %)
static bool parity_window_halves(long tiles) {
  return tiles <= 256 && !(svs_tune_on(SVS_TUNE_CONV_WINDOW) && svs_tune(SVS_TUNE_CONV_WINDOW) == 3);
}

static bool narrow_level(int mode, int B, int C, int Wo, int N) {
  if (svs_tune_on(SVS_TUNE_CONV_SKIP) || svs_tune_on(SVS_TUNE_CONV_KORDER)) return false;          // sweeps and tests keep the generic tiles
  return N > 32 && (C & (C - 1)) == 0 && B >= 16 && ((mode == MODE_GATHER) ? Wo : (Wo + 1) / 2) <= 8;
}

// narrow levels: batch-innermost rows so that whole taps of an M-tile fall into the padding and are skipped
static int use_tap_skip(int mode, int B, int C, int Wo, int N, int cfg) {
  const bool can = N > 32 /* tap-outer K order */ && (C & (C - 1)) == 0 && (cfg == 0 || cfg == 1 || cfg == 4 || cfg == 5 || cfg == 6);
  int skip = can && B >= 16 && ((mode == MODE_GATHER) ? Wo : (Wo + 1) / 2) <= 8;
  if (svs_tune_on(SVS_TUNE_CONV_SKIP)) {     // sweeps and tests: 0 = never, 2 = whenever the kernel supports it
    const int f = (int)svs_tune(SVS_TUNE_CONV_SKIP);
    skip = (f == 0) ? 0 : (f == 2) ? can : skip;
  }
  if (svs_tune_on(SVS_TUNE_CONV_KORDER)) skip = 0;           // (the K-order sweep switch may select tap-inner order)
  return skip;
}

// Shared by enc fwd / dec bwd_data (GATHER) and dec fwd / enc bwd_data (PARITY).
int svs_conv_gemm_run(int mode, const float* x, long ldx, int B, int H, int W, int C, const float* wp,
                      const float* bias, const float* scale, const float* shift, float slope, float* y, long ldy,
                      int Ho, int Wo, int N, int accumulate, void* ws, size_t ws_bytes, hipStream_t stream,
                      const char* who, float* stats, int stats_cap, int* stats_nblk) {   // stats_cap: capacity of `stats` in floats
  if (stats_nblk) *stats_nblk = 0;
  int rc = check_gemm_args(who, x, ldx, B, H, W, C, wp, y, ldy, Ho, Wo, N);
  if (rc) return rc;
  long Mmax;
  int nkt_min;
  if (mode == MODE_GATHER) {
    SVS_REQUIRE(Ho == svs_conv_out(H) && Wo == svs_conv_out(W), "%s: output %dx%d does not match input %dx%d", who, Ho, Wo, H, W);
    Mmax = (long)B * Ho * Wo;
    nkt_min = 25 * (C / 16);
  } else {
    SVS_REQUIRE((Ho == 2 * H || Ho == 2 * H - 1) && (Wo == 2 * W || Wo == 2 * W - 1),
                "%s: output %dx%d unreachable from input %dx%d", who, Ho, Wo, H, W);
    Mmax = (long)B * ((Ho + 1) / 2) * ((Wo + 1) / 2);
    nkt_min = 4 * (C / 16);
  }
  // operands are addressed with 32-bit byte offsets (buffer loads): each view must stay below 2 GiB
  SVS_REQUIRE(((long)B * H * W * ldx + 4L * (W + 2) * ldx) * 4 < (1L << 31) && (long)N * C * 25 * 4 < (1L << 31),
              "%s: input view of %ld bytes needs 64-bit offsets; split the batch", who, (long)B * H * W * ldx * 4);
  ConvPlan pl = plan_conv(mode, Mmax, N, nkt_min, narrow_level(mode, B, C, Wo, N), scale != nullptr);
  ConvGemmArgs a{};
  a.x = x; a.ldx = ldx; a.B = B; a.H = H; a.W = W; a.C = C; a.wp = wp;
  a.bias = bias; a.scale = scale; a.shift = shift; a.slope = slope;
  a.y = y; a.ldy = ldy; a.Ho = Ho; a.Wo = Wo; a.N = N; a.accumulate = accumulate;
  a.slab = nullptr;
  // With few output channels the im2col operand dominates the traffic; consuming all taps of a 16-channel
  // chunk before the next chunk keeps a block's re-read window in cache (same-device A/B: 5% faster for the
  // N<=32 layers, 1-2% slower for the deep ones, hence the switch).
  a.cpt_shift = -1;
  if ((C & (C - 1)) == 0) { a.cpt_shift = 0; while ((16 << a.cpt_shift) < C) ++a.cpt_shift; }
  a.tap_inner = N <= 32;
  if (svs_tune_on(SVS_TUNE_CONV_KORDER)) a.tap_inner = svs_tune(SVS_TUNE_CONV_KORDER) != 0;     // sweeps only
  const long P = (long)B * Ho * Wo;
  const int window = use_parity_window(mode, B, H, W, C, N, ldx);
  if (window) pl.ksplit = 1;
  ConvBal bal{};
  const int nslab_bal = (!window && use_tap_skip(mode, B, C, Wo, N, pl.cfg)) ? plan_balance(mode, B, H, W, C, Ho, Wo, N, pl, &bal) : 0;
  if (nslab_bal) pl.ksplit = nslab_bal;          // slabs = the largest split count of any position
  a.ksplit = pl.ksplit;
  if (pl.ksplit > 1) {
    const size_t need = (size_t)pl.ksplit * P * N * sizeof(float) + (nslab_bal ? svs_align_up((size_t)P, 16) : 0);
    if (!ws || ws_bytes < need || !svs_aligned16(ws)) {
      svs_set_error("%s: workspace too small (%zu < %zu)", who, ws_bytes, need);
      return SVS_ERR_WORKSPACE;
    }
    a.slab = (float*)ws;
    if (nslab_bal) bal.rowsplit = (unsigned char*)ws + (size_t)pl.ksplit * P * N * sizeof(float);
  }
  // LDS-free kernel for the 16-channel outputs (same-device A/B: 1.2-1.3x there; N = 32 is mixed, so it stays on
  // the LDS kernel except in parity mode with a deep reduction)
  int direct = 0;
  if (Mmax >= 16384 && ((long)B * H * W * ldx + 4L * (W + 2) * ldx) * 4 < (1L << 31)) {
    if (N == 16) direct = 1;                                    // 64 rows per wave (128 measured slower)
    else if (N == 32 && mode == MODE_PARITY && C >= 128) direct = 1;
  }
  if (svs_tune_on(SVS_TUNE_CONV_DIRECT)) { const int f = (int)svs_tune(SVS_TUNE_CONV_DIRECT); if (f == 0 || (N <= 32 && Mmax >= 16384)) direct = f; }  // sweeps
  const bool want_stats = stats && stats_nblk && !scale && !accumulate;
  // conv2 forward: LDS-window form of the GATHER mode (SVS_CONV_GWINDOW=0: the GEMM kernel)
  if (mode == MODE_GATHER && C == 16 && N == 32 && !accumulate && ldy % 4 == 0 && svs_aligned16(y) && svs_tune(SVS_TUNE_CONV_GWINDOW) != 0 &&
      ((long)H * W * ldx) * 4 < (1L << 31)) {
    const long gtiles = (long)B * ((Ho + 7) / 8) * ((Wo + 15) / 16);
    const bool forced = svs_tune(SVS_TUNE_CONV_GWINDOW) == 2;          // tests: whenever the layer is eligible
    if ((forced || (Ho >= 8 && Wo >= 16 && gtiles >= 256)) && gtiles < (1L << 30)) {
      const long tblocks = svs_tune(SVS_TUNE_CONV_GWINDOW) >= 16 ? svs_tune(SVS_TUNE_CONV_GWINDOW) : 512;     // (>= 16: sweeps)
      int tpb = (int)(gtiles / tblocks);         // one round of two resident blocks per CU; >= 1 tile per block
      if (tpb < 1) tpb = 1;
      if (tpb > 8) tpb = 8;
      const long gblocks = (gtiles + tpb - 1) / tpb;
      a.ksplit = 1; a.slab = nullptr;
      if (want_stats && gblocks * 2 * N <= stats_cap) { a.stats = stats; *stats_nblk = (int)gblocks; }
      hipLaunchKernelGGL(gather_window_kernel, dim3((unsigned)gblocks), dim3(256), 0, stream, a, (int)gtiles, tpb);
      SVS_CHECK_LAUNCH("gather_window");
      return SVS_OK;
    }
  }
  if (window) {
    a.ksplit = 1; a.slab = nullptr;
    dim3 grid((unsigned)((long)B * ((H + 7) / 8) * ((W + 15) / 16)));
    if (want_stats && (long)grid.x * 2 * N <= stats_cap) { a.stats = stats; *stats_nblk = (int)grid.x; }
#define SVS_LAUNCH_WINDOW(C_, CW_, TN_) hipLaunchKernelGGL((parity_window_kernel<C_, CW_, TN_>), grid, dim3(256), 0, stream, a)
    if (N == 16) {
      if (C == 32) SVS_LAUNCH_WINDOW(32, 32, 1); else if (C == 64) SVS_LAUNCH_WINDOW(64, 64, 1); else SVS_LAUNCH_WINDOW(128, 64, 1);
    } else if (parity_window_halves(grid.x)) {
      grid.y = 2;
#define SVS_LAUNCH_WINDOW_HALF(C_, CW_) hipLaunchKernelGGL((parity_window_kernel<C_, CW_, 1, 2>), grid, dim3(256), 0, stream, a)
      if (C == 32) SVS_LAUNCH_WINDOW_HALF(32, 32); else if (C == 64) SVS_LAUNCH_WINDOW_HALF(64, 64); else SVS_LAUNCH_WINDOW_HALF(128, 64);
#undef SVS_LAUNCH_WINDOW_HALF
    } else {
      if (C == 32) SVS_LAUNCH_WINDOW(32, 32, 2); else if (C == 64) SVS_LAUNCH_WINDOW(64, 64, 2); else SVS_LAUNCH_WINDOW(128, 64, 2);
    }
#undef SVS_LAUNCH_WINDOW
    SVS_CHECK_LAUNCH("parity_window");
    return SVS_OK;
  }
  if (direct && (N == 16 || N == 32)) {
    a.ksplit = 1; a.slab = nullptr;
    const int rows = (direct == 2 ? 8 : 4) * 64;      // 4 independent waves per block, TM*16 output rows each
    dim3 grid((unsigned)((Mmax + rows - 1) / rows), (unsigned)pl.grid_y, 1);
#define SVS_LAUNCH_DIRECT(MODE_, TM_, TN_) hipLaunchKernelGGL((conv_direct_kernel<MODE_, TM_, TN_>), grid, dim3(256), 0, stream, a)
    if (mode == MODE_GATHER) {
      if (N == 16) { if (direct == 2) SVS_LAUNCH_DIRECT(MODE_GATHER, 8, 1); else SVS_LAUNCH_DIRECT(MODE_GATHER, 4, 1); }
      else { if (direct == 2) SVS_LAUNCH_DIRECT(MODE_GATHER, 8, 2); else SVS_LAUNCH_DIRECT(MODE_GATHER, 4, 2); }
    } else {
      if (N == 16) { if (direct == 2) SVS_LAUNCH_DIRECT(MODE_PARITY, 8, 1); else SVS_LAUNCH_DIRECT(MODE_PARITY, 4, 1); }
      else { if (direct == 2) SVS_LAUNCH_DIRECT(MODE_PARITY, 8, 2); else SVS_LAUNCH_DIRECT(MODE_PARITY, 4, 2); }
    }
#undef SVS_LAUNCH_DIRECT
    SVS_CHECK_LAUNCH("conv_direct");
    return SVS_OK;
  }
  const int skip = use_tap_skip(mode, B, C, Wo, N, pl.cfg);
  if (want_stats && pl.ksplit == 1 && pl.mtiles * pl.grid_y * 2 * N <= stats_cap) {
    a.stats = stats;                         // one row of partials per (parity class, M-tile)
    *stats_nblk = (int)(pl.mtiles * pl.grid_y);
  }
  rc = (mode == MODE_GATHER) ? launch_conv_gemm<MODE_GATHER>(a, pl, stream, skip, bal) : launch_conv_gemm<MODE_PARITY>(a, pl, stream, skip, bal);
  if (rc) return rc;
  if (pl.ksplit > 1 && !svs_tune_flag(SVS_TUNE_SKIP_REDUCE)) {      // (the switch lets bench.py time the GEMM kernel alone)
    const long total4 = P * N / 4;
    int grid = (int)((total4 + 255) / 256);
    if (grid > 2048) grid = 2048;
    // fused BatchNorm statistics: stats[grid][2][N] must fit the caller's buffer (stats_cap rows)
    const bool fuse = stats && stats_nblk && stats_cap >= 2 * N && N % 4 == 0 && N <= 1024 && 256 % (N / 4) == 0 && !scale && !accumulate;
    if (fuse && grid > 512) grid = 512;
    if (fuse && (long)grid * 2 * N > stats_cap) grid = stats_cap / (2 * N);
    int n_shift = -1;
    if ((N & (N - 1)) == 0) { n_shift = 0; while ((1 << n_shift) < N) ++n_shift; }
    if (fuse && N % 32 == 0 && svs_tune(SVS_TUNE_BN_INLINE) != 0) {
      // slab form: blocks = slabs x pixel ranges, at most 512 (two per CU), a range no shorter than one pass of 32 pixels
      const int nslab = N / 32;
      long rows = 512 / nslab;
      if (rows > (P + 31) / 32) rows = (P + 31) / 32;
      if (rows * 2 * N > stats_cap) rows = stats_cap / (2 * N);
      if (rows < 1) rows = 1;
      const long ppr = (P + rows - 1) / rows;
      rows = (P + ppr - 1) / ppr;
      hipLaunchKernelGGL(splitk_epilogue_stats_kernel, dim3((unsigned)(rows * nslab)), dim3(256), 0, stream, a.slab, pl.ksplit, P, N, bias,
                         y, ldy, stats, ppr, bal.enabled ? bal.rowsplit : nullptr);
      SVS_CHECK_LAUNCH("splitk_epilogue_stats");
      *stats_nblk = (int)rows;
      return SVS_OK;
    }
    hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(grid), dim3(256), 0, stream, a.slab, pl.ksplit, P, N, bias, scale,
                       shift, slope, y, ldy, accumulate, fuse ? stats : nullptr, n_shift, bal.enabled ? bal.rowsplit : nullptr);
    SVS_CHECK_LAUNCH("splitk_epilogue");
    if (fuse) *stats_nblk = grid;
  }
  return SVS_OK;
}

size_t svs_conv_gemm_workspace(int mode, int B, int H, int W, int C, int Ho, int Wo, int N) {
  if (C < 16 || N < 16) return 0;
  long Mmax;
  int nkt_min;
  if (mode == MODE_GATHER) { Mmax = (long)B * Ho * Wo; nkt_min = 25 * (C / 16); }
  else { Mmax = (long)B * ((Ho + 1) / 2) * ((Wo + 1) / 2); nkt_min = 4 * (C / 16); }
  const int ks_train = plan_conv(mode, Mmax, N, nkt_min, narrow_level(mode, B, C, Wo, N), false).ksplit;
  const int ks_eval = plan_conv(mode, Mmax, N, nkt_min, narrow_level(mode, B, C, Wo, N), true).ksplit;
  int ks = ks_train > ks_eval ? ks_train : ks_eval;               // (one workspace serves either kind of call, balanced or not)
  size_t extra = 0;
  for (int inference = 0; inference < 2; ++inference) {            // balanced splits (ConvBal) may need more slabs, and the row table
    const ConvPlan pl = plan_conv(mode, Mmax, N, nkt_min, narrow_level(mode, B, C, Wo, N), inference != 0);
    const bool window = use_parity_window(mode, B, H, W, C, N, C) != 0;
    const int sb = (!window && use_tap_skip(mode, B, C, Wo, N, pl.cfg)) ? plan_balance(mode, B, H, W, C, Ho, Wo, N, pl, nullptr) : 0;
    if (sb) { ks = sb > ks ? sb : ks; extra = svs_align_up((size_t)B * Ho * Wo, 16); }
  }
  if (ks <= 1) return 0;
  return (size_t)ks * B * Ho * Wo * N * sizeof(float) + extra;
}

// Name (as rocprofv3 prints it) and K-split of the kernel the planner picks for a conv GEMM -- bench.py groups its
// live per-layer timings by this name so that they can be matched against the rocprofv3 kernel statistics.
int svs_conv_gemm_describe(int mode, int B, int H, int W, int C, int Ho, int Wo, int N, long ldx, char* buf, size_t n) {
  long Mmax;
  int nkt_min;
  if (mode == MODE_GATHER) { Mmax = (long)B * Ho * Wo; nkt_min = 25 * (C / 16); }
  else { Mmax = (long)B * ((Ho + 1) / 2) * ((Wo + 1) / 2); nkt_min = 4 * (C / 16); }
  int direct = 0;
  if (Mmax >= 16384 && ((long)B * H * W * ldx + 4L * (W + 2) * ldx) * 4 < (1L << 31)) {
    if (N == 16) direct = 1;
    else if (N == 32 && mode == MODE_PARITY && C >= 128) direct = 1;
  }
  if (mode == MODE_GATHER && C == 16 && N == 32 && svs_tune(SVS_TUNE_CONV_GWINDOW) != 0 && Ho >= 8 && Wo >= 16 &&
      (long)B * ((Ho + 7) / 8) * ((Wo + 15) / 16) >= 256) {
    snprintf(buf, n, "gather_window_kernel");
    return 1;
  }
  if (use_parity_window(mode, B, H, W, C, N, ldx)) {
    const bool halves = N == 32 && parity_window_halves((long)B * ((H + 7) / 8) * ((W + 15) / 16));
    snprintf(buf, n, "parity_window_kernel<%d, %d, %d, %d>", C, C == 32 ? 32 : 64, halves ? 1 : N / 16, N / 16);
    return 1;
  }
  if (direct) { snprintf(buf, n, "conv_direct_kernel<%d, 4, %d>", mode, N / 16); return 1; }
  static const int wm[10] = {2, 2, 4, 4, 1, 2, 2, 4, 4, 4}, wn[10] = {2, 2, 1, 1, 4, 2, 2, 1, 1, 1};     // (6 = 64x128, 2x2 waves)
  const ConvPlan pl = plan_conv(mode, Mmax, N, nkt_min, narrow_level(mode, B, C, Wo, N));
  const bool skip = use_tap_skip(mode, B, C, Wo, N, pl.cfg) != 0;
  // (the last template argument, K-tiles requested ahead, exists as 2 only for the tile shapes launch_conv_gemm_cfg has it for)
  const int pf = (pl.pf == 2 && (pl.cfg == 5 || pl.cfg == 6 || pl.cfg == 1 || (!skip && (pl.cfg == 2 || pl.cfg == 3)))) ? 2 : 1;
  snprintf(buf, n, "conv_gemm_kernel<%d, %d, %d, %d, %d, %s, %s, %d>", mode, pl.BM, pl.BN, wm[pl.cfg], wn[pl.cfg],
           skip ? "true" : "false", svs_tune(SVS_TUNE_MFMA_SPLIT) > 0 ? "true" : "false", pf);
  ConvBal show{};
  const int sb = skip ? plan_balance(mode, B, H, W, C, Ho, Wo, N, pl, svs_tune(SVS_TUNE_CONV_BALANCE) == 2 ? &show : nullptr) : 0;      // balanced: the largest split count
  return sb ? sb : pl.ksplit;
}
