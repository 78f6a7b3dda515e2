// Bandwidth-bound pieces of the training / inference step (gfx950): BatchNorm statistics, finalise,
// apply (+ LeakyReLU/ReLU + Dropout2d), BatchNorm backward, the L1 mask loss, Adam, weight packing,
// eval-mode BN folding, synthetic data.  All tensors are fp32 NHWC (pixel-major), read and written
// as float4 by C/4 adjacent lanes per pixel.  Bound: HBM (each tensor once per pass).
#include "internal.h"

// ------------------------------------------------------------------------------------------------
// per-channel block reduction shared by bn_stats (sum x, sum x^2) and bn_bwd (sum dz, sum dz*xhat)
// partial layout: ws[blk][2][C]
// ------------------------------------------------------------------------------------------------
static inline int red_blocks(long P, int C) {     // upper bound of the partial rows (sizes the workspaces)
  long nb = (P * C) / 4096;              // 16 float4 per thread: the small (deep-level) tensors are latency-bound, not bandwidth-bound
  if (nb > 1024) nb = 1024;
  if (nb < 1) nb = 1;
  return (int)nb;
}

// A reducing / applying block owns a SLAB of min(C, 32) adjacent channels (one 128-byte line per pixel: LP = 8 float4 lanes,
// 32 pixels per pass of the block) and a range of pixels; block -> (slab, range) with the slab innermost, so the blocks that
// run together read the same lines.  The partials of a channel then have as many rows as there are pixel RANGES (32-512), not
// blocks: few enough that the kernel which consumes the statistics sums them itself (bn_block_sums) instead of waiting
// for a finalise launch of its own -- 22 launches of 5-8 us each on the train step's critical path.
struct RedPlan { int sw, nslab, lp, rows; long ppr; };     // slab width (channels), slabs, float4 lanes per pixel, rows, pixels per row
static RedPlan red_plan(long P, int C) {
  RedPlan r;
  r.sw = C < 32 ? C : 32;
  r.nslab = C / r.sw;
  r.lp = r.sw / 4;
  long cap = svs_tune(SVS_TUNE_BN_BLOCKS) > 0 ? svs_tune(SVS_TUNE_BN_BLOCKS) : 512;
  if (cap > 1024) cap = 1024;                       // (red_blocks: the workspaces' bound)
  long nb = (P * C) / 4096;
  if (nb > cap) nb = cap;
  long rows = nb / r.nslab;
  if (rows < 1) rows = 1;
  r.ppr = (P + rows - 1) / rows;
  r.rows = (int)((P + r.ppr - 1) / r.ppr);          // (no empty rows)
  return r;
}
static inline int red_rows(long P, int C) { return red_plan(P, C).rows; }

struct BnCtx {
  const float* raw; long ldr; long P; int C; long pps;    // pps: pixels per sample (dropout index)
  const float* gamma; const float* beta; const float* mean; const float* invstd;
  float slope; const float* drop;
  const float* dy; long lddy;
  int nslab, lp; long ppr;                                // RedPlan
};
static inline void set_plan(BnCtx& p, const RedPlan& r) { p.nslab = r.nslab; p.lp = r.lp; p.ppr = r.ppr; }

// MODE 0: stats of raw.  MODE 1: backward sums.  partial[row][2][C]
template <int MODE>
__global__ __launch_bounds__(256) void channel_reduce_kernel(BnCtx p, float* __restrict__ partial) {
  __shared__ f32x4 red[2][256];
  const int t = threadIdx.x;
  const int slab = blockIdx.x % p.nslab, row = blockIdx.x / p.nslab;
  const int l = t % p.lp, pr = t / p.lp, PPB = 256 / p.lp;
  const int c = slab * (p.lp * 4) + l * 4;
  const long p0 = (long)row * p.ppr;
  long p1 = p0 + p.ppr;
  if (p1 > p.P) p1 = p.P;
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
  f32x4 mean4, k4, beta4, inv4;
  if (MODE == 1) {
    mean4 = *(const f32x4*)(p.mean + c);
    inv4 = *(const f32x4*)(p.invstd + c);
    k4 = *(const f32x4*)(p.gamma + c) * inv4;
    beta4 = *(const f32x4*)(p.beta + c);
  }
#pragma unroll 4
  for (long pix = p0 + pr; pix < p1; pix += PPB) {        // (unrolled: four pixels' loads in flight per thread)
    const f32x4 x = *(const f32x4*)(p.raw + pix * p.ldr + c);
    if (MODE == 0) {
      s0 += x;
      s1 += x * x;
    } else {
      f32x4 dz = *(const f32x4*)(p.dy + pix * p.lddy + c);
      if (p.drop) dz *= *(const f32x4*)(p.drop + (long)((unsigned)pix / (unsigned)p.pps) * p.C + c);     // (P < 2^31: 32-bit division)
      const f32x4 xm = x - mean4;
      const f32x4 z = xm * k4 + beta4;
#pragma unroll
      for (int k = 0; k < 4; ++k) dz[k] = z[k] > 0.f ? dz[k] : dz[k] * p.slope;
      s0 += dz;
      s1 += dz * (xm * inv4);
    }
  }
  red[0][t] = s0;
  red[1][t] = s1;
  __syncthreads();
  if (t < p.lp) {
    f32x4 a = red[0][t], b = red[1][t];
    for (int j = 1; j < PPB; ++j) { a += red[0][j * p.lp + t]; b += red[1][j * p.lp + t]; }
    float* out = partial + (long)row * 2 * p.C + c;
    *(f32x4*)out = a;
    *(f32x4*)(out + p.C) = b;
  }
}

// Sums the per-workgroup partials ws[blk][2][C] of 4 adjacent channels with one 256-thread block
// (double accumulation, fixed order: reproducible).  Result: lane k (<4) of wave 0 gets (s, ss) of channel c0+k.
__device__ __forceinline__ void reduce_partials4(const float* __restrict__ partial, int nblk, int C, int c0, double* s, double* ss) {
  __shared__ double sh[2][4][4];     // [which][wave][channel]
  double a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
  for (int r = threadIdx.x; r < nblk; r += 256) {
    const f32x4 x = *(const f32x4*)(partial + (long)r * 2 * C + c0);
    const f32x4 y = *(const f32x4*)(partial + (long)r * 2 * C + C + c0);
#pragma unroll
    for (int k = 0; k < 4; ++k) { a[k] += (double)x[k]; b[k] += (double)y[k]; }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a[k] += __shfl_xor(a[k], o, 64); b[k] += __shfl_xor(b[k], o, 64); }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) { sh[0][wave][k] = a[k]; sh[1][wave][k] = b[k]; }
  }
  __syncthreads();
  const int k = threadIdx.x & 3;
  *s = (sh[0][0][k] + sh[0][1][k]) + (sh[0][2][k] + sh[0][3][k]);
  *ss = (sh[1][0][k] + sh[1][1][k]) + (sh[1][2][k] + sh[1][3][k]);
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ partial, int nblk, long P, int C, float eps,
                                                          float momentum, float* running_mean, float* running_var,
                                                          long long* nbt, float* save_mean, float* save_invstd) {
  const int c0 = blockIdx.x * 4;
  double s, ss;
  reduce_partials4(partial, nblk, C, c0, &s, &ss);
  if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) nbt[0] += 1;
  if (threadIdx.x >= 4) return;
  const int c = c0 + threadIdx.x;
  const double mean = s / (double)P;
  double var = ss / (double)P - mean * mean;
  if (var < 0.0) var = 0.0;
  save_mean[c] = (float)mean;
  save_invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) {
    const double unb = P > 1 ? var * ((double)P / (double)(P - 1)) : var;
    running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
    running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
  }
}

// Block-local finalise.  The block's threads are (l, pr) = (float4 lane of the slab, pixel row of a pass); the 256 / lp threads
// of a lane share the R partial rows of its four channels and fold them in a fixed tree (double accumulation: every block,
// whatever its index, gets the same bits).  Afterwards sums[which][l][k] holds the totals of channel c0 + 4 l + k.
__device__ __forceinline__ void bn_block_sums(const float* __restrict__ partial, int R, int C, int c, int lp, double (*sums)[256][4]) {
  const int t = threadIdx.x, pr = t / lp, PPB = 256 / lp;
  double a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
#pragma unroll 2
  for (int r = pr; r < R; r += PPB) {
    const f32x4 x = *(const f32x4*)(partial + (long)r * 2 * C + c);
    const f32x4 y = *(const f32x4*)(partial + (long)r * 2 * C + C + c);
#pragma unroll
    for (int k = 0; k < 4; ++k) { a[k] += (double)x[k]; b[k] += (double)y[k]; }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) { sums[0][t][k] = a[k]; sums[1][t][k] = b[k]; }
  __syncthreads();
  for (int st = PPB >> 1; st > 0; st >>= 1) {
    if (pr < st) {
#pragma unroll
      for (int k = 0; k < 4; ++k) { sums[0][t][k] += sums[0][t + st * lp][k]; sums[1][t][k] += sums[1][t + st * lp][k]; }
    }
    __syncthreads();
  }
}

struct BnFin {                        // what bn_finalize_kernel takes, for the apply kernel that does its work
  const float* partial; int rows; float eps, momentum;
  float* running_mean; float* running_var; long long* nbt; float* save_mean; float* save_invstd;
};

// bn_finalize + bn_act_apply in one launch, slab form (see RedPlan).  grid = nslab * ranges; the range-0 block of each slab
// also writes the saved statistics and the running buffers of its channels.
__global__ __launch_bounds__(256) void bn_fin_act_apply_kernel(BnCtx p, BnFin f, float* __restrict__ y, long ldy) {
  __shared__ double sums[2][256][4];
  __shared__ __attribute__((aligned(16))) float fin[2][32];
  const int t = threadIdx.x;
  const int slab = blockIdx.x % p.nslab, row = blockIdx.x / p.nslab;
  const int l = t % p.lp, pr = t / p.lp, PPB = 256 / p.lp;
  const int c0 = slab * (p.lp * 4), c = c0 + l * 4;
  bn_block_sums(f.partial, f.rows, p.C, c, p.lp, sums);
  if (t < p.lp * 4) {
    const double mean = sums[0][t >> 2][t & 3] / (double)p.P;
    double var = sums[1][t >> 2][t & 3] / (double)p.P - mean * mean;
    if (var < 0.0) var = 0.0;
    const float mf = (float)mean, inv = (float)(1.0 / sqrt(var + (double)f.eps));
    fin[0][t] = mf;
    fin[1][t] = inv;
    if (row == 0) {
      const int ch = c0 + t;
      f.save_mean[ch] = mf;
      f.save_invstd[ch] = inv;
      if (f.running_mean) {
        const double unb = p.P > 1 ? var * ((double)p.P / (double)(p.P - 1)) : var;
        f.running_mean[ch] = (float)((1.0 - f.momentum) * f.running_mean[ch] + f.momentum * mean);
        f.running_var[ch] = (float)((1.0 - f.momentum) * f.running_var[ch] + f.momentum * unb);
      }
      if (f.nbt && blockIdx.x == 0 && t == 0) f.nbt[0] += 1;
    }
  }
  __syncthreads();
  const f32x4 mean4 = *(const f32x4*)&fin[0][l * 4];
  const f32x4 k4 = *(const f32x4*)(p.gamma + c) * *(const f32x4*)&fin[1][l * 4];
  const f32x4 beta4 = *(const f32x4*)(p.beta + c);
  const long p0 = (long)row * p.ppr;
  long p1 = p0 + p.ppr;
  if (p1 > p.P) p1 = p.P;
  const unsigned pps = (unsigned)p.pps;
#pragma unroll 4
  for (long pix = p0 + pr; pix < p1; pix += PPB) {
    const f32x4 x = *(const f32x4*)(p.raw + pix * p.ldr + c);
    f32x4 z = (x - mean4) * k4 + beta4;
#pragma unroll
    for (int k = 0; k < 4; ++k) z[k] = z[k] > 0.f ? z[k] : z[k] * p.slope;
    if (p.drop) z *= *(const f32x4*)(p.drop + (long)((unsigned)pix / pps) * p.C + c);
    *(f32x4*)(y + pix * ldy + c) = z;
  }
}

// (index arithmetic in 32 bits -- P * C / 4 < 2^32 is checked on the host -- and by shift when C / 4 is a power of two, which it is
// for every layer of this network: the 64-bit `%` and `/` per float4 made this stream VALU-bound at 4.2 TB/s)
__global__ __launch_bounds__(256) void bn_act_apply_kernel(BnCtx p, float* __restrict__ y, long ldy, int g_shift) {
  const unsigned G = (unsigned)p.C >> 2;
  const unsigned total = (unsigned)(p.P * G);
  const unsigned pps = (unsigned)p.pps;
#pragma unroll 4
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    unsigned cg, pixu;
    if (g_shift >= 0) { cg = i & (G - 1); pixu = i >> g_shift; }
    else { pixu = i / G; cg = i - pixu * G; }
    const long pix = pixu;
    const f32x4 x = *(const f32x4*)(p.raw + pix * p.ldr + cg * 4);
    const f32x4 k4 = *(const f32x4*)(p.gamma + cg * 4) * *(const f32x4*)(p.invstd + cg * 4);
    f32x4 z = (x - *(const f32x4*)(p.mean + cg * 4)) * k4 + *(const f32x4*)(p.beta + cg * 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) z[k] = z[k] > 0.f ? z[k] : z[k] * p.slope;
    if (p.drop) z *= *(const f32x4*)(p.drop + (long)(pixu / pps) * p.C + cg * 4);
    *(f32x4*)(y + pix * ldy + cg * 4) = z;
  }
}

// coef[0][c] = gamma*invstd, coef[1][c] = mean(dz), coef[2][c] = mean(dz*xhat)
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int nblk, long P, int C,
                                                              const float* __restrict__ gamma, const float* __restrict__ invstd,
                                                              float* dgamma, float* dbeta, float* coef) {
  const int c0 = blockIdx.x * 4;
  double s, sx;
  reduce_partials4(partial, nblk, C, c0, &s, &sx);
  if (threadIdx.x >= 4) return;
  const int c = c0 + threadIdx.x;
  if (dbeta) dbeta[c] = (float)s;
  if (dgamma) dgamma[c] = (float)sx;
  coef[c] = gamma[c] * invstd[c];
  coef[C + c] = (float)(s / (double)P);
  coef[2 * C + c] = (float)(sx / (double)P);
}

// d_raw = k * (dz - mean(dz) - xhat * mean(dz*xhat)); when `partial` is given the block also leaves the per-channel
// sum of its d_raw values there ([row][2][C] layout, first half) -- that sum over all rows is the gradient of the
// conv bias in front of this BatchNorm (the reference gets it from autograd; true value 0, what remains is
// rounding noise), obtained here without a second pass over d_raw.  Slab form (RedPlan).  FIN: the block takes the two
// means from the reduce pass's partial rows itself (bn_block_sums) instead of from a bn_bwd_finalize launch; the range-0
// block of each slab then also writes dgamma / dbeta.  `partial` must not be the buffer `sum_rows` points to.
template <bool FIN>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(BnCtx p, const float* __restrict__ coef, float* __restrict__ d_raw,
                                                           float* __restrict__ partial, const float* __restrict__ sum_rows, int nrows,
                                                           float* dgamma, float* dbeta) {
  __shared__ double sums[2][FIN ? 256 : 1][4];
  __shared__ __attribute__((aligned(16))) float fin[2][32];
  __shared__ f32x4 red[256];
  const int t = threadIdx.x;
  const int slab = blockIdx.x % p.nslab, row = blockIdx.x / p.nslab;
  const int l = t % p.lp, pr = t / p.lp, PPB = 256 / p.lp;
  const int c0 = slab * (p.lp * 4), c = c0 + l * 4;
  const f32x4 inv4 = *(const f32x4*)(p.invstd + c);
  const f32x4 mean4 = *(const f32x4*)(p.mean + c);
  const f32x4 beta4 = *(const f32x4*)(p.beta + c);
  f32x4 k4, c1, c2;
  if constexpr (FIN) {
    bn_block_sums(sum_rows, nrows, p.C, c, p.lp, sums);
    if (t < p.lp * 4) {
      const double s = sums[0][t >> 2][t & 3], sx = sums[1][t >> 2][t & 3];
      fin[0][t] = (float)(s / (double)p.P);
      fin[1][t] = (float)(sx / (double)p.P);
      if (row == 0) {
        if (dbeta) dbeta[c0 + t] = (float)s;
        if (dgamma) dgamma[c0 + t] = (float)sx;
      }
    }
    __syncthreads();
    k4 = *(const f32x4*)(p.gamma + c) * inv4;
    c1 = *(const f32x4*)&fin[0][l * 4];
    c2 = *(const f32x4*)&fin[1][l * 4];
  } else {
    k4 = *(const f32x4*)(coef + c);
    c1 = *(const f32x4*)(coef + p.C + c);
    c2 = *(const f32x4*)(coef + 2 * p.C + c);
  }
  const long p0 = (long)row * p.ppr;
  long p1 = p0 + p.ppr;
  if (p1 > p.P) p1 = p.P;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (long pix = p0 + pr; pix < p1; pix += PPB) {
    const f32x4 x = *(const f32x4*)(p.raw + pix * p.ldr + c);
    f32x4 dz = *(const f32x4*)(p.dy + pix * p.lddy + c);
    if (p.drop) dz *= *(const f32x4*)(p.drop + (long)((unsigned)pix / (unsigned)p.pps) * p.C + c);
    const f32x4 xm = x - mean4;
    const f32x4 z = xm * k4 + beta4;
#pragma unroll
    for (int k = 0; k < 4; ++k) dz[k] = z[k] > 0.f ? dz[k] : dz[k] * p.slope;
    const f32x4 r = k4 * (dz - c1 - (xm * inv4) * c2);
    *(f32x4*)(d_raw + pix * p.C + c) = r;
    acc += r;
  }
  if (partial) {
    red[t] = acc;
    __syncthreads();
    if (t < p.lp) {
      f32x4 a = red[t];
      for (int j = 1; j < PPB; ++j) a += red[j * p.lp + t];
      *(f32x4*)(partial + (long)row * 2 * p.C + c) = a;
    }
  }
}

static int check_bn(const char* who, const float* raw, long ldr, long P, int C) {
  SVS_REQUIRE(raw, "%s: null pointer", who);
  SVS_REQUIRE(C >= 4 && C % 4 == 0 && C <= 1024 && (256 % (C / 4) == 0 || (C / 4) % 256 == 0), "%s: unsupported C=%d", who, C);
  SVS_REQUIRE(P > 0 && P < (1L << 31) && ldr >= C && ldr % 4 == 0 && svs_aligned16(raw), "%s: bad view", who);
  return SVS_OK;
}

static int grid_for(long total) {
  long g = (total + 255) / 256;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}

extern "C" size_t svs_bn_workspace_bytes(int64_t P, int C) {
  return (size_t)red_blocks(P, C) * 2 * C * sizeof(float) + (size_t)3 * C * sizeof(float);
}

extern "C" int svs_bn_stats(const float* raw, int64_t ldr, int64_t P, int C, void* ws, size_t ws_bytes, hipStream_t stream) {
  int rc = check_bn("svs_bn_stats", raw, ldr, P, C);
  if (rc) return rc;
  if (!ws || ws_bytes < svs_bn_workspace_bytes(P, C)) { svs_set_error("svs_bn_stats: workspace too small"); return SVS_ERR_WORKSPACE; }
  const RedPlan rp = red_plan(P, C);
  BnCtx p{}; p.raw = raw; p.ldr = ldr; p.P = P; p.C = C; set_plan(p, rp);
  hipLaunchKernelGGL(channel_reduce_kernel<0>, dim3(rp.rows * rp.nslab), dim3(256), 0, stream, p, (float*)ws);
  SVS_CHECK_LAUNCH("bn_stats");
  return SVS_OK;
}

extern "C" int svs_bn_finalize(const void* ws, int64_t P, int C, float eps, float momentum, float* running_mean,
                               float* running_var, int64_t* num_batches_tracked, float* save_mean, float* save_invstd,
                               hipStream_t stream) {
  return svs_bn_finalize_run(ws, red_rows(P, C), P, C, eps, momentum, running_mean, running_var,
                             (long long*)num_batches_tracked, save_mean, save_invstd, stream);
}

int svs_bn_finalize_run(const void* partial, int nblk, long P, int C, float eps, float momentum, float* running_mean,
                        float* running_var, long long* nbt, float* save_mean, float* save_invstd, hipStream_t stream) {
  SVS_REQUIRE(partial && save_mean && save_invstd && nblk > 0 && C % 4 == 0, "svs_bn_finalize: bad arguments");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C / 4), dim3(256), 0, stream, (const float*)partial, nblk, P, C, eps,
                     momentum, running_mean, running_var, nbt, save_mean, save_invstd);
  SVS_CHECK_LAUNCH("bn_finalize");
  return SVS_OK;
}

size_t svs_bn_partial_floats(long P, int C) { return (size_t)red_blocks(P, C) * 2 * C; }     // (capacity: an upper bound of the rows)
int svs_bn_partial_rows(long P, int C) { return red_rows(P, C); }

// Inline finalise (bn_block_sums) pays when the rows a block has to fold are few: rows * 2 * min(C, 32) floats per block.
// SVS_BN_INLINE: 0 = never (finalise launches as before), n > 0 = row limit; default 128 (same-device A/B of the batch-64 train
// step: off 3.409 ms, 128 rows 3.381, 512 rows 3.384, 2048 rows 3.404 -- the shallow levels' 512-1024 rows cost a block more than
// the launch they save).
static bool fin_inline(int rows, int C) {
  const long lim = svs_tune(SVS_TUNE_BN_INLINE) >= 0 ? svs_tune(SVS_TUNE_BN_INLINE) : 128;
  return rows <= lim && C >= 4 && (C <= 32 ? 256 % (C / 4) == 0 : C % 32 == 0);
}

// svs_bn_finalize + svs_bn_act_apply; one launch when the partial rows are few enough (fin_inline)
int svs_bn_fin_act_apply_run(const void* partial, int rows, const float* raw, long ldr, long P, int C, long pixels_per_sample,
                             const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                             float* running_var, long long* nbt, float* save_mean, float* save_invstd, float slope,
                             const float* drop, float* y, long ldy, hipStream_t stream) {
  int rc = check_bn("svs_bn_fin_act_apply", raw, ldr, P, C);
  if (rc) return rc;
  SVS_REQUIRE(partial && rows > 0 && save_mean && save_invstd, "svs_bn_fin_act_apply: bad arguments");
  SVS_REQUIRE(y && ldy >= C && ldy % 4 == 0 && svs_aligned16(y), "svs_bn_fin_act_apply: bad output view");
  if (!fin_inline(rows, C)) {
    if ((rc = svs_bn_finalize_run(partial, rows, P, C, eps, momentum, running_mean, running_var, nbt, save_mean, save_invstd, stream))) return rc;
    return svs_bn_act_apply(raw, ldr, P, C, pixels_per_sample, gamma, beta, save_mean, save_invstd, slope, drop, y, ldy, stream);
  }
  SVS_REQUIRE(pixels_per_sample > 0 && pixels_per_sample < (1L << 31), "svs_bn_fin_act_apply: bad pixels_per_sample");
  const RedPlan rp = red_plan(P, C);
  BnCtx p{}; p.raw = raw; p.ldr = ldr; p.P = P; p.C = C; p.pps = pixels_per_sample; set_plan(p, rp);
  p.gamma = gamma; p.beta = beta; p.slope = slope; p.drop = drop;
  BnFin f{(const float*)partial, rows, eps, momentum, running_mean, running_var, nbt, save_mean, save_invstd};
  hipLaunchKernelGGL(bn_fin_act_apply_kernel, dim3(rp.rows * rp.nslab), dim3(256), 0, stream, p, f, y, ldy);
  SVS_CHECK_LAUNCH("bn_fin_act_apply");
  return SVS_OK;
}

extern "C" int svs_bn_act_apply(const float* raw, int64_t ldr, int64_t P, int C, int64_t pixels_per_sample,
                                const float* gamma, const float* beta, const float* save_mean, const float* save_invstd,
                                float slope, const float* drop, float* y, int64_t ldy, hipStream_t stream) {
  int rc = check_bn("svs_bn_act_apply", raw, ldr, P, C);
  if (rc) return rc;
  SVS_REQUIRE(y && ldy >= C && ldy % 4 == 0 && svs_aligned16(y), "svs_bn_act_apply: bad output view");
  BnCtx p{}; p.raw = raw; p.ldr = ldr; p.P = P; p.C = C; p.pps = pixels_per_sample;
  p.gamma = gamma; p.beta = beta; p.mean = save_mean; p.invstd = save_invstd; p.slope = slope; p.drop = drop;
  SVS_REQUIRE(P * (C / 4) < (1L << 32) - 4096L * 256 && pixels_per_sample > 0 && pixels_per_sample < (1L << 31),
              "svs_bn_act_apply: %ld pixels x %d channels need 64-bit indices; split the batch", (long)P, C);
  const int G = C / 4;
  int g_shift = -1;
  if ((G & (G - 1)) == 0) { g_shift = 0; while ((1 << g_shift) < G) ++g_shift; }
  hipLaunchKernelGGL(bn_act_apply_kernel, dim3(grid_for(P * (C / 4))), dim3(256), 0, stream, p, y, (long)ldy, g_shift);
  SVS_CHECK_LAUNCH("bn_act_apply");
  return SVS_OK;
}

int svs_bn_bwd_run(const float* dy, long lddy, const float* raw, long ldr, long P, int C, long pixels_per_sample,
                   const float* gamma, const float* beta, const float* save_mean, const float* save_invstd, float slope,
                   const float* drop, float* d_raw, float* dgamma, float* dbeta, float* dbias, void* ws, size_t ws_bytes,
                   hipStream_t stream, float* dbias_partial, SvsSumJobs* defer);
__global__ void channel_sum_finalize_kernel(const float* __restrict__ partial, int nblk, int C, float* out);

extern "C" int svs_bn_bwd(const float* dy, int64_t lddy, const float* raw, int64_t ldr, int64_t P, int C,
                          int64_t pixels_per_sample, const float* gamma, const float* beta, const float* save_mean,
                          const float* save_invstd, float slope, const float* drop, float* d_raw, float* dgamma,
                          float* dbeta, void* ws, size_t ws_bytes, hipStream_t stream) {
  return svs_bn_bwd_run(dy, lddy, raw, ldr, P, C, pixels_per_sample, gamma, beta, save_mean, save_invstd, slope, drop, d_raw,
                        dgamma, dbeta, nullptr, ws, ws_bytes, stream, nullptr, nullptr);
}

int svs_bn_bwd_run(const float* dy, long lddy, const float* raw, long ldr, long P, int C, long pixels_per_sample,
                   const float* gamma, const float* beta, const float* save_mean, const float* save_invstd, float slope,
                   const float* drop, float* d_raw, float* dgamma, float* dbeta, float* dbias, void* ws, size_t ws_bytes,
                   hipStream_t stream, float* dbias_partial, SvsSumJobs* defer) {
  int rc = check_bn("svs_bn_bwd", raw, ldr, P, C);
  if (rc) return rc;
  SVS_REQUIRE(dy && d_raw && lddy >= C && lddy % 4 == 0 && svs_aligned16(dy) && svs_aligned16(d_raw), "svs_bn_bwd: bad gradient view");
  if (!ws || ws_bytes < svs_bn_workspace_bytes(P, C)) { svs_set_error("svs_bn_bwd: workspace too small"); return SVS_ERR_WORKSPACE; }
  const RedPlan rp = red_plan(P, C);
  const int nb = rp.rows, grid = rp.rows * rp.nslab;
  float* partial = (float*)ws;
  float* coef = partial + (size_t)nb * 2 * C;
  BnCtx p{}; p.raw = raw; p.ldr = ldr; p.P = P; p.C = C; p.pps = pixels_per_sample; set_plan(p, rp);
  p.gamma = gamma; p.beta = beta; p.mean = save_mean; p.invstd = save_invstd; p.slope = slope; p.drop = drop;
  p.dy = dy; p.lddy = lddy;
  hipLaunchKernelGGL(channel_reduce_kernel<1>, dim3(grid), dim3(256), 0, stream, p, partial);
  SVS_CHECK_LAUNCH("bn_bwd_reduce");
  const bool deferred = dbias && dbias_partial && defer && defer->njobs < 12;
  // the apply pass reuses the partial buffer of the reduce pass (already consumed by the finalize kernel) unless the
  // caller keeps the bias-gradient partials for a deferred, batched final pass -- which the inline finalise needs too:
  // its blocks still read the reduce pass's rows while others write theirs
  float* bpart = deferred ? dbias_partial : partial;
  if (fin_inline(nb, C) && (!dbias || deferred)) {
    hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, dim3(grid), dim3(256), 0, stream, p, (const float*)nullptr, d_raw, dbias ? bpart : nullptr,
                       (const float*)partial, nb, dgamma, dbeta);
  } else {
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C / 4), dim3(256), 0, stream, (const float*)partial, nb, (long)P, C,
                       gamma, save_invstd, dgamma, dbeta, coef);
    SVS_CHECK_LAUNCH("bn_bwd_finalize");
    hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, dim3(grid), dim3(256), 0, stream, p, (const float*)coef, d_raw, dbias ? bpart : nullptr,
                       (const float*)nullptr, 0, (float*)nullptr, (float*)nullptr);
  }
  SVS_CHECK_LAUNCH("bn_bwd_apply");
  if (deferred) {
    const int j = defer->njobs++;
    defer->partial[j] = bpart; defer->nblk[j] = nb; defer->C[j] = C; defer->out[j] = dbias;
  } else if (dbias) {
    hipLaunchKernelGGL(channel_sum_finalize_kernel, dim3(C / 4), dim3(256), 0, stream, (const float*)partial, nb, C, dbias);
    SVS_CHECK_LAUNCH("channel_sum_finalize");
  }
  return SVS_OK;
}

// per-channel sum of a (P, C) view -> out[C]   (bias gradients).  Uses the stats kernel's partials.
int svs_channel_sum_run(const float* x, long ldx, long P, int C, float* out, void* ws, size_t ws_bytes, hipStream_t stream);

__global__ __launch_bounds__(256) void channel_sum_finalize_kernel(const float* __restrict__ partial, int nblk, int C, float* out) {
  double s, ss;
  reduce_partials4(partial, nblk, C, blockIdx.x * 4, &s, &ss);
  if (threadIdx.x < 4) out[blockIdx.x * 4 + threadIdx.x] = (float)s;
}

// the same for several (partial, C, out) jobs in one launch: block -> (job, channel quad)
__global__ __launch_bounds__(256) void channel_sum_finalize_multi_kernel(SvsSumJobs jobs) {
  int blk = blockIdx.x, j = 0;
  while (j < jobs.njobs - 1 && blk >= jobs.C[j] / 4) { blk -= jobs.C[j] / 4; ++j; }
  double s, ss;
  reduce_partials4(jobs.partial[j], jobs.nblk[j], jobs.C[j], blk * 4, &s, &ss);
  if (threadIdx.x < 4) jobs.out[j][blk * 4 + threadIdx.x] = (float)s;
}

int svs_channel_sum_finalize_multi_run(const SvsSumJobs& jobs, hipStream_t stream) {
  if (jobs.njobs <= 0) return SVS_OK;
  int blocks = 0;
  for (int j = 0; j < jobs.njobs; ++j) blocks += jobs.C[j] / 4;
  hipLaunchKernelGGL(channel_sum_finalize_multi_kernel, dim3(blocks), dim3(256), 0, stream, jobs);
  SVS_CHECK_LAUNCH("channel_sum_finalize_multi");
  return SVS_OK;
}

int svs_channel_sum_run(const float* x, long ldx, long P, int C, float* out, void* ws, size_t ws_bytes, hipStream_t stream) {
  int rc = svs_bn_stats(x, ldx, P, C, ws, ws_bytes, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(channel_sum_finalize_kernel, dim3(C / 4), dim3(256), 0, stream, (const float*)ws, red_rows(P, C), C, out);
  SVS_CHECK_LAUNCH("channel_sum_finalize");
  return SVS_OK;
}

// ------------------------------------------------------------------------------------------------
// scalar reductions: two-stage, fixed order
// ------------------------------------------------------------------------------------------------
#define SCALAR_BLOCKS 512

__device__ __forceinline__ float block_sum(float v, float* sh) {
  v = svs_wave_sum(v);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = (sh[0] + sh[1]) + (sh[2] + sh[3]);
  __syncthreads();
  return r;
}

__global__ __launch_bounds__(256) void l1_mask_loss_kernel(const float* __restrict__ mask, const float* __restrict__ mix,
                                                           const float* __restrict__ voc, long n, float gscale,
                                                           float* __restrict__ d_logit, float* __restrict__ partial) {
  __shared__ float sh[4];
  float s = 0.f;
  auto one = [&](float m, float x, float v, float& dl) __attribute__((always_inline)) {
    const float d1 = m * x - v;
    const float ta = fmaxf(x - v, 0.f);
    const float d2 = (1.f - m) * x - ta;
    s += fabsf(d1) + fabsf(d2);
    const float s1 = d1 > 0.f ? 1.f : (d1 < 0.f ? -1.f : 0.f);
    const float s2 = d2 > 0.f ? 1.f : (d2 < 0.f ? -1.f : 0.f);
    const float dm = (s1 - s2) * x * gscale;
    dl = dm * m * (1.f - m);
  };
  const long n4 = n >> 2;                              // float4 body (all three inputs and d_logit are 16-byte aligned), scalar tail
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const f32x4 m = ((const f32x4*)mask)[i], x = ((const f32x4*)mix)[i], v = ((const f32x4*)voc)[i];
    f32x4 d;
#pragma unroll
    for (int k = 0; k < 4; ++k) { float dl; one(m[k], x[k], v[k], dl); d[k] = dl; }
    ((f32x4*)d_logit)[i] = d;
  }
  for (long i = (n4 << 2) + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float dl; one(mask[i], mix[i], voc[i], dl); d_logit[i] = dl;
  }
  s = block_sum(s, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ void sum_partials_kernel(const float* __restrict__ partial, int nb, double scale, float* out) {
  __shared__ double sh[64];
  double s = 0.0;
  for (int i = threadIdx.x; i < nb; i += 64) s += (double)partial[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double tsum = 0.0;
    for (int i = 0; i < 64; ++i) tsum += sh[i];
    out[0] = (float)(tsum * scale);
  }
}

extern "C" size_t svs_l1_mask_loss_workspace_bytes(int64_t n) { (void)n; return SCALAR_BLOCKS * sizeof(float); }

extern "C" int svs_l1_mask_loss_fwd_bwd(const float* mask, const float* mix, const float* voc, int64_t n, float loss_scale,
                                        float* d_logit, float* loss, void* ws, size_t ws_bytes, hipStream_t stream) {
  SVS_REQUIRE(mask && mix && voc && d_logit && loss && n > 0, "svs_l1_mask_loss_fwd_bwd: null pointer");
  SVS_REQUIRE(svs_aligned16(mask) && svs_aligned16(mix) && svs_aligned16(voc) && svs_aligned16(d_logit), "svs_l1_mask_loss_fwd_bwd: pointers must be 16-byte aligned");
  if (!ws || ws_bytes < SCALAR_BLOCKS * sizeof(float)) { svs_set_error("svs_l1_mask_loss_fwd_bwd: workspace too small"); return SVS_ERR_WORKSPACE; }
  int nb = (int)((n + 255) / 256);
  if (nb > SCALAR_BLOCKS) nb = SCALAR_BLOCKS;
  hipLaunchKernelGGL(l1_mask_loss_kernel, dim3(nb), dim3(256), 0, stream, mask, mix, voc, (long)n, loss_scale / (float)n, d_logit, (float*)ws);
  SVS_CHECK_LAUNCH("l1_mask_loss");
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(64), 0, stream, (const float*)ws, nb, 1.0 / (double)n, loss);
  SVS_CHECK_LAUNCH("sum_partials");
  return SVS_OK;
}

// d_logit = d_mask * mask * (1 - mask)    (autograd path: the loss was built by the caller in torch)
__global__ __launch_bounds__(256) void sigmoid_bwd_kernel(const float* __restrict__ mask, const float* __restrict__ dmask,
                                                          long n, float* __restrict__ d_logit) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float m = mask[i];
    d_logit[i] = dmask[i] * m * (1.f - m);
  }
}
int svs_sigmoid_bwd_run(const float* mask, const float* dmask, long n, float* d_logit, hipStream_t stream) {
  hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, stream, mask, dmask, n, d_logit);
  SVS_CHECK_LAUNCH("sigmoid_bwd");
  return SVS_OK;
}

__global__ __launch_bounds__(256) void sum_kernel(const float* __restrict__ x, long n, float* __restrict__ partial) {
  __shared__ float sh[4];
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) s += x[i];
  s = block_sum(s, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
// out[0] = sum(x); ws: SCALAR_BLOCKS floats
int svs_sum_run(const float* x, long n, float* out, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (!ws || ws_bytes < SCALAR_BLOCKS * sizeof(float)) { svs_set_error("svs_sum: workspace too small"); return SVS_ERR_WORKSPACE; }
  int nb = (int)((n + 255) / 256);
  if (nb > SCALAR_BLOCKS) nb = SCALAR_BLOCKS;
  hipLaunchKernelGGL(sum_kernel, dim3(nb), dim3(256), 0, stream, x, n, (float*)ws);
  SVS_CHECK_LAUNCH("sum");
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(64), 0, stream, (const float*)ws, nb, 1.0, out);
  SVS_CHECK_LAUNCH("sum_partials");
  return SVS_OK;
}

__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long n, float* __restrict__ partial) {
  __shared__ float sh[4];
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) s = fmaxf(s, fabsf(x[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s = fmaxf(s, __shfl_xor(s, o, 64));
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}
__global__ void max_partials_kernel(const float* __restrict__ partial, int nb, float* out) {
  float s = 0.f;
  for (int i = threadIdx.x; i < nb; i += 64) s = fmaxf(s, partial[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s = fmaxf(s, __shfl_xor(s, o, 64));
  if (threadIdx.x == 0) out[0] = s;
}
extern "C" int svs_absmax(const float* x, int64_t n, float* out, void* ws, size_t ws_bytes, hipStream_t stream) {
  SVS_REQUIRE(x && out && n > 0, "svs_absmax: null pointer");
  if (!ws || ws_bytes < SCALAR_BLOCKS * sizeof(float)) { svs_set_error("svs_absmax: workspace too small"); return SVS_ERR_WORKSPACE; }
  int nb = (int)((n + 255) / 256);
  if (nb > SCALAR_BLOCKS) nb = SCALAR_BLOCKS;
  hipLaunchKernelGGL(absmax_kernel, dim3(nb), dim3(256), 0, stream, x, (long)n, (float*)ws);
  SVS_CHECK_LAUNCH("absmax");
  hipLaunchKernelGGL(max_partials_kernel, dim3(1), dim3(64), 0, stream, (const float*)ws, nb, out);
  SVS_CHECK_LAUNCH("max_partials");
  return SVS_OK;
}

extern "C" int svs_max(const float* x, int64_t n, float* out, hipStream_t stream) {
  SVS_REQUIRE(x && out && n > 0 && n < (1L << 31), "svs_max: bad arguments");
  hipLaunchKernelGGL(max_partials_kernel, dim3(1), dim3(64), 0, stream, x, (int)n, out);
  SVS_CHECK_LAUNCH("max_partials");
  return SVS_OK;
}

__global__ __launch_bounds__(256) void scale_by_inv_kernel(float* x, long n, const float* denom, float numer) {
  float d = denom[0];
  if (d == 0.f) d = 1.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) x[i] = x[i] / d * numer;
}
extern "C" int svs_scale_by_inv(float* x, int64_t n, const float* denom, float numer, hipStream_t stream) {
  SVS_REQUIRE(x && denom && n > 0, "svs_scale_by_inv: null pointer");
  hipLaunchKernelGGL(scale_by_inv_kernel, dim3(grid_for(n)), dim3(256), 0, stream, x, (long)n, denom, numer);
  SVS_CHECK_LAUNCH("scale_by_inv");
  return SVS_OK;
}

// ------------------------------------------------------------------------------------------------
// Adam (torch.optim.Adam single-tensor update, model.py:116)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long n, float b1, float b2, float omb1,
                                                   float omb2, float eps, float step_size, float bc2_sqrt, float gscale) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float gi = g[i] * gscale;
    const float mi = m[i] * b1 + omb1 * gi;
    const float vi = v[i] * b2 + omb2 * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] -= step_size * (mi / denom);
  }
}
extern "C" int svs_adam_step(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2,
                             double eps, int step, float grad_scale, hipStream_t stream) {
  SVS_REQUIRE(p && g && m && v && n > 0 && step >= 1, "svs_adam_step: bad arguments");
  const double bc1 = 1.0 - pow(beta1, step);
  const double bc2 = 1.0 - pow(beta2, step);
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, stream, p, g, m, v, (long)n, (float)beta1,
                     (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, (float)(lr / bc1), (float)sqrt(bc2),
                     grad_scale);
  SVS_CHECK_LAUNCH("adam");
  return SVS_OK;
}

__global__ __launch_bounds__(256) void apply_mask_kernel(const float* __restrict__ mix, const float* __restrict__ mask,
                                                         float* __restrict__ out, long n, int invert) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float m = mask[i];
    if (invert) m = 1.f - m;
    out[i] = mix[i] * m;
  }
}
extern "C" int svs_apply_mask(const float* mix, const float* mask, float* out, int64_t n, int invert, hipStream_t stream) {
  SVS_REQUIRE(mix && mask && out && n > 0, "svs_apply_mask: null pointer");
  hipLaunchKernelGGL(apply_mask_kernel, dim3(grid_for(n)), dim3(256), 0, stream, mix, mask, out, (long)n, invert);
  SVS_CHECK_LAUNCH("apply_mask");
  return SVS_OK;
}

// ------------------------------------------------------------------------------------------------
// weight packing / BN fold / synthetic data
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_gather_kernel(const float* __restrict__ w, float* __restrict__ wp, int N, int C) {
  const long total = (long)N * C * 25;
  for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < total; o += (long)gridDim.x * 256) {
    const int c = (int)(o % C);
    const long r = o / C;
    const int tap = (int)(r % 25);
    const long n = r / 25;
    wp[o] = w[(n * C + c) * 25 + tap];
  }
}
extern "C" int svs_pack_weight_gather(const float* w, float* wp, int N, int C, hipStream_t stream) {
  SVS_REQUIRE(w && wp && N > 0 && C > 0, "svs_pack_weight_gather: bad arguments");
  hipLaunchKernelGGL(pack_gather_kernel, dim3(grid_for((long)N * C * 25)), dim3(256), 0, stream, w, wp, N, C);
  SVS_CHECK_LAUNCH("pack_gather");
  return SVS_OK;
}

__global__ __launch_bounds__(256) void pack_parity_kernel(const float* __restrict__ w, float* __restrict__ wp, int C, int N) {
  const long total = (long)N * C * 25;
  const long NC = (long)N * C;
  for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < total; o += (long)gridDim.x * 256) {
    int par; long rem;
    if (o < 9 * NC) { par = 0; rem = o; }
    else if (o < 15 * NC) { par = 1; rem = o - 9 * NC; }
    else if (o < 21 * NC) { par = 2; rem = o - 15 * NC; }
    else { par = 3; rem = o - 21 * NC; }
    const int ph = par >> 1, pw = par & 1;
    const int ntw = 3 - pw, ntaps = (3 - ph) * ntw;
    const int c = (int)(rem % C);
    const long r = rem / C;
    const int tap = (int)(r % ntaps);
    const long n = r / ntaps;
    const int th = tap / ntw, tw = tap - th * ntw;
    const int kh = ph + 2 * th, kw = pw + 2 * tw;
    wp[o] = w[((long)c * N + n) * 25 + kh * 5 + kw];
  }
}
extern "C" int svs_pack_weight_parity(const float* w, float* wp, int C, int N, hipStream_t stream) {
  SVS_REQUIRE(w && wp && N > 0 && C > 0, "svs_pack_weight_parity: bad arguments");
  hipLaunchKernelGGL(pack_parity_kernel, dim3(grid_for((long)N * C * 25)), dim3(256), 0, stream, w, wp, C, N);
  SVS_CHECK_LAUNCH("pack_parity");
  return SVS_OK;
}

// All weight packings of a step in ONE launch (10 layers x 2 layouts used to be 20 launches of ~8 us each).
// One block per output-channel row; the row's C x 25 weights go through LDS so that both the global reads
// (25- or 25*C-float runs) and the global writes (C-float runs) are contiguous.
__global__ __launch_bounds__(256) void pack_all_kernel(SvsPackJobs jobs) {
  __shared__ float tile[512 * 25];
  int ji = 0;
  while (ji + 1 < jobs.n && (int)blockIdx.x >= jobs.j[ji + 1].first_block) ++ji;
  const SvsPackJob jb = jobs.j[ji];
  const int n = blockIdx.x - jb.first_block;
  const int C = jb.C, N = jb.N, tot = C * 25;
  // C is a power of two for every layer (16 .. 512): index splits by shift / mask (the runtime divisions were most of the
  // kernel's instructions); lc < 0 keeps the general form
  const int lc = (C & (C - 1)) == 0 ? 31 - __builtin_clz((unsigned)C) : -1;
  if (jb.kind == 0) {           // gather: w[n][c][tap] -> wp[n][tap][c]
    const float* src = jb.w + (long)n * tot;
    for (int e = threadIdx.x; e < tot; e += 256) tile[e] = src[e];
    __syncthreads();
    float* dst = jb.wp + (long)n * tot;
    for (int e = threadIdx.x; e < tot; e += 256) {
      const int tap = lc >= 0 ? e >> lc : e / C, c = e - tap * C;
      dst[e] = tile[c * 25 + tap];
    }
  } else {                      // parity: w[c][n][tap] -> wp[p][n][th][tw][c]
    for (int e = threadIdx.x; e < tot; e += 256) {
      const int c = e / 25, tap = e - c * 25;
      tile[e] = jb.w[((long)c * N + n) * 25 + tap];
    }
    __syncthreads();
    const long NC = (long)N * C;
#pragma unroll
    for (int par = 0; par < 4; ++par) {
      const int ph = par >> 1, pw = par & 1, ntw = 3 - pw, ntaps = (3 - ph) * ntw;
      const int poff = (par == 0) ? 0 : (par == 1) ? 9 : (par == 2) ? 15 : 21;
      float* dst = jb.wp + poff * NC + (long)n * ntaps * C;
      for (int e = threadIdx.x; e < ntaps * C; e += 256) {
        const int t2 = lc >= 0 ? e >> lc : e / C, c = e - t2 * C;
        const int th = t2 / ntw, tw = t2 - th * ntw;            // (ntw is 3 or 2: compile-time after unrolling)
        dst[e] = tile[c * 25 + (ph + 2 * th) * 5 + pw + 2 * tw];
      }
    }
  }
}

int svs_pack_all_run(SvsPackJobs& jobs, hipStream_t stream) {
  int blocks = 0;
  for (int i = 0; i < jobs.n; ++i) {
    SVS_REQUIRE(jobs.j[i].C <= 512 && jobs.j[i].w && jobs.j[i].wp, "svs_pack_all: bad job %d", i);
    jobs.j[i].first_block = blocks;
    blocks += jobs.j[i].N;
  }
  if (!blocks) return SVS_OK;
  hipLaunchKernelGGL(pack_all_kernel, dim3(blocks), dim3(256), 0, stream, jobs);
  SVS_CHECK_LAUNCH("pack_all");
  return SVS_OK;
}

__global__ void bn_fold_kernel(const float* gamma, const float* beta, const float* rm, const float* rv, const float* bias,
                               float eps, float* scale, float* shift, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float s = gamma[c] / sqrtf(rv[c] + eps);
  scale[c] = s;
  shift[c] = beta[c] + ((bias ? bias[c] : 0.f) - rm[c]) * s;
}
extern "C" int svs_bn_fold(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                           const float* conv_bias, float eps, float* scale, float* shift, int C, hipStream_t stream) {
  SVS_REQUIRE(gamma && beta && running_mean && running_var && scale && shift && C > 0, "svs_bn_fold: bad arguments");
  hipLaunchKernelGGL(bn_fold_kernel, dim3((C + 63) / 64), dim3(64), 0, stream, gamma, beta, running_mean, running_var,
                     conv_bias, eps, scale, shift, C);
  SVS_CHECK_LAUNCH("bn_fold");
  return SVS_OK;
}

__global__ __launch_bounds__(256) void fill_uniform_kernel(float* out, long n, uint32_t seed, uint64_t offset, float scale, float shift) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    out[i] = (float)(svs_u32(seed, offset + (uint64_t)i) >> 8) * 5.9604644775390625e-08f * scale + shift;
}
extern "C" int svs_fill_uniform(float* out, int64_t n, uint32_t seed, uint64_t offset, float scale, float shift, hipStream_t stream) {
  SVS_REQUIRE(out && n > 0, "svs_fill_uniform: bad arguments");
  hipLaunchKernelGGL(fill_uniform_kernel, dim3(grid_for(n)), dim3(256), 0, stream, out, (long)n, seed, offset, scale, shift);
  SVS_CHECK_LAUNCH("fill_uniform");
  return SVS_OK;
}

__global__ __launch_bounds__(256) void fill_tiles_kernel(float* mix, float* voc, long per, long total, long first_tile) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long b = i / per;
    const uint64_t ctr = (uint64_t)(i - b * per) + ((uint64_t)(first_tile + b) << 32);
    const float m = (float)(svs_u32(0u, ctr) >> 8) * 5.9604644775390625e-08f;
    const float r = (float)(svs_u32(1u, ctr) >> 8) * 5.9604644775390625e-08f;
    mix[i] = m;
    voc[i] = m * r;
  }
}
extern "C" int svs_fill_tiles(float* mix, float* voc, int B, int H, int W, int64_t first_tile, hipStream_t stream) {
  SVS_REQUIRE(mix && voc && B > 0 && H > 0 && W > 0, "svs_fill_tiles: bad arguments");
  const long per = (long)H * W, total = per * B;
  hipLaunchKernelGGL(fill_tiles_kernel, dim3(grid_for(total)), dim3(256), 0, stream, mix, voc, per, total, (long)first_tile);
  SVS_CHECK_LAUNCH("fill_tiles");
  return SVS_OK;
}

__global__ void dropout_mask_kernel(float* out, int n, uint32_t seed, uint64_t offset) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (float)(svs_u32(seed, offset + (uint64_t)i) >> 31) * 2.f;
}
// all five decoder masks in one launch: out = [B*256 | B*128 | B*64 | B*32 | B*16], each as svs_dropout_mask draws it
__global__ void dropout_masks_all_kernel(float* out, int B, uint32_t seed, int step, int rank) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * 496) return;
  int layer = 0, base = 0, c = 256;
  while (i >= base + B * c) { base += B * c; c >>= 1; ++layer; }
  const uint64_t off = ((uint64_t)layer << 56) | ((uint64_t)(rank & 0xFF) << 48) | ((uint64_t)(step & 0xFFFFFF) << 24);
  out[i] = (float)(svs_u32(seed, off + (uint64_t)(i - base)) >> 31) * 2.f;
}
extern "C" int svs_dropout_masks_all(float* out, int B, uint32_t seed, int step, int rank, hipStream_t stream) {
  SVS_REQUIRE(out && B > 0 && (long)B * 496 < (1L << 31), "svs_dropout_masks_all: bad arguments");
  const int n = B * 496;
  hipLaunchKernelGGL(dropout_masks_all_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, out, B, seed, step, rank);
  SVS_CHECK_LAUNCH("dropout_masks_all");
  return SVS_OK;
}
extern "C" int svs_dropout_mask(float* out, int B, int C, int layer, uint32_t seed, int step, int rank, hipStream_t stream) {
  SVS_REQUIRE(out && B > 0 && C > 0 && layer >= 0 && layer < 5, "svs_dropout_mask: bad arguments");
  const uint64_t off = ((uint64_t)layer << 56) | ((uint64_t)(rank & 0xFF) << 48) | ((uint64_t)(step & 0xFFFFFF) << 24);
  const int n = B * C;
  hipLaunchKernelGGL(dropout_mask_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, out, n, seed, off);
  SVS_CHECK_LAUNCH("dropout_mask");
  return SVS_OK;
}

// ------------------------------------------------------------------------------------------------
// Training tiles from spectrograms resident in HBM (train.py:86-143, SpectrogramDataset.__getitem__): sample b is
// rows 1..F of song[b]'s (F+1, T) magnitude file -- the DC row is already dropped when the song is uploaded -- columns
// [start[b], start[b] + seg), right zero-padded when the song is shorter; mixture and vocal share the start.
// songs_*: one flat buffer per track type, song s at offset[s], (F, T_s) row-major.  One thread = 4 frames.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void crop_tiles_kernel(const float* __restrict__ mix_songs, const float* __restrict__ voc_songs,
                                                         const long long* __restrict__ offset, const int* __restrict__ frames,
                                                         const int* __restrict__ song, const int* __restrict__ start, int B,
                                                         int F, int seg, float* __restrict__ mix, float* __restrict__ voc) {
  const long total = (long)B * F * (seg / 4);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int t4 = (int)(i % (seg / 4));
    const long row = i / (seg / 4);
    const int f = (int)(row % F), b = (int)(row / F);
    const int s = song[b], T = frames[s], st = start[b];
    const long src = offset[s] + (long)f * T + st + 4 * t4;
    f32x4 m, v;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bool in = st + 4 * t4 + k < T;
      m[k] = in ? mix_songs[src + k] : 0.f;
      v[k] = in ? voc_songs[src + k] : 0.f;
    }
    *(f32x4*)(mix + row * seg + 4 * t4) = m;
    *(f32x4*)(voc + row * seg + 4 * t4) = v;
  }
}

// np.angle of the unit-phasor files (train.py:103-104): angle[i] = atan2(im, re), float32
__global__ __launch_bounds__(256) void phase_angle_kernel(const float2* __restrict__ z, float* __restrict__ out, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = atan2f(z[i].y, z[i].x);
}
extern "C" int svs_phase_angle(const float* phasor, float* angle, int64_t n, hipStream_t stream) {
  SVS_REQUIRE(phasor && angle && n > 0 && (((uintptr_t)phasor) & 7u) == 0, "svs_phase_angle: bad arguments");
  hipLaunchKernelGGL(phase_angle_kernel, dim3(grid_for(n)), dim3(256), 0, stream, (const float2*)phasor, angle, (long)n);
  SVS_CHECK_LAUNCH("phase_angle");
  return SVS_OK;
}

extern "C" int svs_crop_tiles(const float* mix_songs, const float* voc_songs, const int64_t* offset, const int32_t* frames,
                              const int32_t* song, const int32_t* start, int B, int F, int seg, float* mix, float* voc,
                              hipStream_t stream) {
  SVS_REQUIRE(mix_songs && voc_songs && offset && frames && song && start && mix && voc, "svs_crop_tiles: null pointer");
  SVS_REQUIRE(B > 0 && F > 0 && seg > 0 && seg % 4 == 0 && svs_aligned16(mix) && svs_aligned16(voc), "svs_crop_tiles: bad geometry");
  hipLaunchKernelGGL(crop_tiles_kernel, dim3(grid_for((long)B * F * (seg / 4))), dim3(256), 0, stream, mix_songs, voc_songs,
                     (const long long*)offset, frames, song, start, B, F, seg, mix, voc);
  SVS_CHECK_LAUNCH("crop_tiles");
  return SVS_OK;
}

