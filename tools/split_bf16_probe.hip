// Probe: fp32-accurate GEMM products on the bf16 MFMA pipe (gfx950).
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/split_bf16_probe.hip -o tools/bin/split_bf16_probe && tools/bin/split_bf16_probe
//
// C[M][N] = sum_k A[M][K] * B[N][K], fp32 operands, LDS-staged 16-deep K-tiles exactly like csrc/gemm_conv.hip
// (f32x4 fragment per lane = 4 consecutive k of one row).  Two inner loops on the same staging:
//   MODE 0: 4 x v_mfma_f32_16x16x4_f32 per 16x16 tile and K-tile                       (what the library does today)
//   MODE 1: each fragment split exactly into three bf16 limbs (x = x0 + x1 + x2, 8 significant bits each), and the six
//           products a0b0, a0b1, a1b0, a1b1, a0b2, a2b0 (everything above 2^-24 relative) issued as THREE
//           v_mfma_f32_16x16x32_bf16: the instruction's K = 32 holds two limb pairs of the lane's 4 k-values,
//           A = [a1|a0] B = [b1|b0],  A = [a1|a0] B = [b0|b1],  A = [a0|a2] B = [b2|b0].
// Reports time, fp32-equivalent TFLOP/s and the error against a float64 host reference for both.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((row >> 1) & 3); }
// limb planes: 8 dwords per row; a 32-lane ds_read_b64 group is 16 rows x 2 chunks -> rows 8..15 take the other half row
__device__ __forceinline__ int swz2(int row, int chunk) { return chunk ^ (((row >> 3) & 1) << 1); }
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// three bf16 limbs of 4 floats, as packed pairs: p[l][0] = (x0, x1), p[l][1] = (x2, x3) of limb l
struct Limbs { unsigned p[3][2]; };
__device__ __forceinline__ unsigned cvt_pk(float a, float b) {          // v_cvt_pk_bf16_f32 (round to nearest even)
  f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ Limbs split3(f32x4 x) {
  Limbs L;
  float r[4] = {x[0], x[1], x[2], x[3]};
#pragma unroll
  for (int l = 0; l < 3; ++l) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const unsigned pk = cvt_pk(r[2 * h], r[2 * h + 1]);
      L.p[l][h] = pk;
      if (l < 2) {
        r[2 * h] -= __builtin_bit_cast(float, pk << 16);                 // exact: the limb is the leading part of r
        r[2 * h + 1] -= __builtin_bit_cast(float, pk & 0xffff0000u);
      }
    }
  }
  return L;
}
__device__ __forceinline__ bf16x8 pair(const Limbs& L, int lo, int hi) {
  u32x4 v = {L.p[lo][0], L.p[lo][1], L.p[hi][0], L.p[hi][1]};
  return __builtin_bit_cast(bf16x8, v);
}

template <int MODE, int BM, int BN, int WM, int WN, int DEPTH = 1>
__global__ __launch_bounds__(256) void gemm_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K) {
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
  constexpr int RA = BM / 64, RB = BN / 64;
  // MODE 0/1: fp32 tiles.  MODE 2: three bf16 limb planes per operand, [plane][row][16 k] = 8 dwords per row and plane
  __shared__ __attribute__((aligned(16))) float As[2][MODE == 2 ? 3 * BM * 8 : BM * 16];
  __shared__ __attribute__((aligned(16))) float Bs[2][MODE == 2 ? 3 * BN * 8 : BN * 16];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lrow = lane & 15, q = lane >> 4;
  const int wm = wave / WN, wn = wave % WN;
  const int ntn = N / BN;
  const long m0 = (long)(blockIdx.x / ntn) * BM, n0 = (long)(blockIdx.x % ntn) * BN;
  const int chunk = t & 3;
  f32x4 ra[RA], rb[RB];
  f32x4 ra2[RA], rb2[RB];                   // DEPTH 2: second register set (tile kt + 2 in flight while kt + 1 waits to be stored)
  auto load_tile = [&](int kt) {
#pragma unroll
    for (int r = 0; r < RA; ++r) ra[r] = *(const f32x4*)(A + (m0 + (t >> 2) + 64 * r) * K + kt * 16 + chunk * 4);
#pragma unroll
    for (int r = 0; r < RB; ++r) rb[r] = *(const f32x4*)(B + (n0 + (t >> 2) + 64 * r) * K + kt * 16 + chunk * 4);
  };
  auto store_tile = [&](int buf) {
    if (MODE == 2) {                          // split once per staged element; ds_write_b64 per limb
#pragma unroll
      for (int r = 0; r < RA; ++r) {
        const int row = (t >> 2) + 64 * r;
        const Limbs L = split3(ra[r]);
#pragma unroll
        for (int l = 0; l < 3; ++l) *(u32x2*)(&As[buf][l * BM * 8 + row * 8 + swz2(row, chunk) * 2]) = (u32x2){L.p[l][0], L.p[l][1]};
      }
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        const int row = (t >> 2) + 64 * r;
        const Limbs L = split3(rb[r]);
#pragma unroll
        for (int l = 0; l < 3; ++l) *(u32x2*)(&Bs[buf][l * BN * 8 + row * 8 + swz2(row, chunk) * 2]) = (u32x2){L.p[l][0], L.p[l][1]};
      }
      return;
    }
#pragma unroll
    for (int r = 0; r < RA; ++r) { const int row = (t >> 2) + 64 * r; *(f32x4*)(&As[buf][row * 16 + swz(row, chunk) * 4]) = ra[r]; }
#pragma unroll
    for (int r = 0; r < RB; ++r) { const int row = (t >> 2) + 64 * r; *(f32x4*)(&Bs[buf][row * 16 + swz(row, chunk) * 4]) = rb[r]; }
  };
  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int nkt = K / 16;
  if (DEPTH == 2 && MODE == 1) {
    auto load2 = [&](int kt) {
#pragma unroll
      for (int r = 0; r < RA; ++r) ra2[r] = *(const f32x4*)(A + (m0 + (t >> 2) + 64 * r) * K + kt * 16 + chunk * 4);
#pragma unroll
      for (int r = 0; r < RB; ++r) rb2[r] = *(const f32x4*)(B + (n0 + (t >> 2) + 64 * r) * K + kt * 16 + chunk * 4);
    };
    auto store2 = [&](int buf) {
#pragma unroll
      for (int r = 0; r < RA; ++r) { const int row = (t >> 2) + 64 * r; *(f32x4*)(&As[buf][row * 16 + swz(row, chunk) * 4]) = ra2[r]; }
#pragma unroll
      for (int r = 0; r < RB; ++r) { const int row = (t >> 2) + 64 * r; *(f32x4*)(&Bs[buf][row * 16 + swz(row, chunk) * 4]) = rb2[r]; }
    };
    auto compute = [&](int buf) {
      f32x4 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) { const int row = wm * (TM * 16) + i * 16 + lrow; fa[i] = *(const f32x4*)(&As[buf][row * 16 + swz(row, q) * 4]); }
#pragma unroll
      for (int j = 0; j < TN; ++j) { const int row = wn * (TN * 16) + j * 16 + lrow; fb[j] = *(const f32x4*)(&Bs[buf][row * 16 + swz(row, q) * 4]); }
      bf16x8 a10[TM], a02[TM], b10[TN], b01[TN], b20[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) { const Limbs L = split3(fa[i]); a10[i] = pair(L, 1, 0); a02[i] = pair(L, 0, 2); }
#pragma unroll
      for (int j = 0; j < TN; ++j) { const Limbs L = split3(fb[j]); b10[j] = pair(L, 1, 0); b01[j] = pair(L, 0, 1); b20[j] = pair(L, 2, 0); }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a02[i], b20[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a10[i], b01[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a10[i], b10[j], acc[i][j], 0, 0, 0);
    };
    // tiles kt (even) travel through set 1 (ra/rb), odd ones through set 2; nkt is even here
    load_tile(0);
    load2(1);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; kt += 2) {
      if (kt + 2 < nkt) load_tile(kt + 2);        // set 1 is free: tile kt went to LDS one iteration ago
      compute(0);
      store2(1);                                  // tile kt + 1, requested a whole iteration ago
      __syncthreads();
      if (kt + 3 < nkt) load2(kt + 3);
      compute(1);
      if (kt + 2 < nkt) store_tile(0);
      __syncthreads();
    }
  } else
  {
  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    const bool more = kt + 1 < nkt;
    if (more) load_tile(kt + 1);
    if (MODE == 2) {
      bf16x8 a10[TM], a02[TM], b10[TN], b01[TN], b20[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = wm * (TM * 16) + i * 16 + lrow;
        Limbs L;
#pragma unroll
        for (int l = 0; l < 3; ++l) { const u32x2 v = *(const u32x2*)(&As[buf][l * BM * 8 + row * 8 + swz2(row, q) * 2]); L.p[l][0] = v[0]; L.p[l][1] = v[1]; }
        a10[i] = pair(L, 1, 0); a02[i] = pair(L, 0, 2);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = wn * (TN * 16) + j * 16 + lrow;
        Limbs L;
#pragma unroll
        for (int l = 0; l < 3; ++l) { const u32x2 v = *(const u32x2*)(&Bs[buf][l * BN * 8 + row * 8 + swz2(row, q) * 2]); L.p[l][0] = v[0]; L.p[l][1] = v[1]; }
        b10[j] = pair(L, 1, 0); b01[j] = pair(L, 0, 1); b20[j] = pair(L, 2, 0);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a02[i], b20[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a10[i], b01[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a10[i], b10[j], acc[i][j], 0, 0, 0);
        }
      if (more) store_tile(buf ^ 1);
      __syncthreads();
      continue;
    }
    f32x4 fa[TM], fb[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) { const int row = wm * (TM * 16) + i * 16 + lrow; fa[i] = *(const f32x4*)(&As[buf][row * 16 + swz(row, q) * 4]); }
#pragma unroll
    for (int j = 0; j < TN; ++j) { const int row = wn * (TN * 16) + j * 16 + lrow; fb[j] = *(const f32x4*)(&Bs[buf][row * 16 + swz(row, q) * 4]); }
    if (MODE == 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][k], fb[j][k], acc[i][j], 0, 0, 0);
    } else {
      bf16x8 a10[TM], a02[TM], b10[TN], b01[TN], b20[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) { const Limbs L = split3(fa[i]); a10[i] = pair(L, 1, 0); a02[i] = pair(L, 0, 2); }
#pragma unroll
      for (int j = 0; j < TN; ++j) { const Limbs L = split3(fb[j]); b10[j] = pair(L, 1, 0); b01[j] = pair(L, 0, 1); b20[j] = pair(L, 2, 0); }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a02[i], b20[j], acc[i][j], 0, 0, 0);      // small terms first
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a10[i], b01[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a10[i], b10[j], acc[i][j], 0, 0, 0);
        }
    }
    if (more) store_tile(buf ^ 1);
    __syncthreads();
  }
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long m = m0 + wm * (TM * 16) + i * 16 + q * 4 + r, n = n0 + wn * (TN * 16) + j * 16 + lrow;
        C[m * N + n] = acc[i][j][r];
      }
}

template <int MODE, int BM, int BN, int WM, int WN, int DEPTH = 1>
static void run(const char* name, const float* dA, const float* dB, float* dC, int M, int N, int K, const std::vector<float>& hA,
                const std::vector<float>& hB) {
  dim3 grid((M / BM) * (N / BN));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((gemm_kernel<MODE, BM, BN, WM, WN, DEPTH>), grid, dim3(256), 0, 0, dA, dB, dC, M, N, K);
  CHECK(hipEventRecord(e0));
  const int reps = 10;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((gemm_kernel<MODE, BM, BN, WM, WN, DEPTH>), grid, dim3(256), 0, 0, dA, dB, dC, M, N, K);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  std::vector<float> hC((size_t)64 * N);
  CHECK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
  double emax = 0, esum = 0, scale = 0; long cnt = 0;
  for (int m = 0; m < 64; m += 7)
    for (int n = 0; n < N; n += 37) {
      double ref = 0, mag = 0;
      for (int k = 0; k < K; ++k) { const double pr = (double)hA[(size_t)m * K + k] * hB[(size_t)n * K + k]; ref += pr; mag += fabs(pr); }
      const double e = fabs(hC[(size_t)m * N + n] - ref) / mag;        // relative to sum |a b| (the rounding-error scale)
      emax = fmax(emax, e); esum += e; scale = fmax(scale, mag); ++cnt;
    }
  printf("%-34s %8.3f ms  %7.1f TFLOP/s (fp32-equivalent)   err / sum|ab|: max %.3e mean %.3e\n", name, ms, 2.0 * M * N * K / ms * 1e-9, emax, esum / cnt);
}

int main(int argc, char** argv) {
  const int M = argc > 2 ? atoi(argv[2]) : 65536, N = argc > 3 ? atoi(argv[3]) : 256, K = argc > 1 ? atoi(argv[1]) : 1024;   // conv-like: few columns, B stays in L2
  std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) * (1.0f / 16777216.0f) - 0.5f); };
  for (auto& v : hA) v = rnd() * expf(4.0f * rnd());      // a few orders of magnitude of dynamic range
  for (auto& v : hB) v = rnd() * expf(4.0f * rnd());
  float *dA, *dB, *dC;
  CHECK(hipMalloc(&dA, hA.size() * 4)); CHECK(hipMalloc(&dB, hB.size() * 4)); CHECK(hipMalloc(&dC, (size_t)M * N * 4));
  CHECK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
  printf("C[%d][%d] = A[%d][%d] B^T, fp32 operands, fp32 accumulation\n", M, N, M, K);
  run<0, 128, 128, 2, 2>("fp32 MFMA 16x16x4, tile 128x128", dA, dB, dC, M, N, K, hA, hB);
  run<1, 128, 128, 2, 2>("3 x bf16 MFMA 16x16x32, 128x128", dA, dB, dC, M, N, K, hA, hB);
  run<0, 64, 64, 2, 2>("fp32 MFMA 16x16x4, tile 64x64", dA, dB, dC, M, N, K, hA, hB);
  run<1, 64, 64, 2, 2>("3 x bf16 MFMA 16x16x32, 64x64", dA, dB, dC, M, N, K, hA, hB);
  run<0, 128, 64, 2, 2>("fp32 MFMA 16x16x4, tile 128x64", dA, dB, dC, M, N, K, hA, hB);
  run<1, 128, 64, 2, 2>("3 x bf16 MFMA 16x16x32, 128x64", dA, dB, dC, M, N, K, hA, hB);
  run<1, 128, 128, 2, 2, 2>("3 x bf16, 128x128, prefetch 2", dA, dB, dC, M, N, K, hA, hB);
  run<1, 64, 64, 2, 2, 2>("3 x bf16, 64x64, prefetch 2", dA, dB, dC, M, N, K, hA, hB);
  run<1, 128, 64, 2, 2, 2>("3 x bf16, 128x64, prefetch 2", dA, dB, dC, M, N, K, hA, hB);
  run<2, 128, 128, 2, 2>("3 x bf16, limbs in LDS, 128x128", dA, dB, dC, M, N, K, hA, hB);
  run<2, 64, 64, 2, 2>("3 x bf16, limbs in LDS, 64x64", dA, dB, dC, M, N, K, hA, hB);
  run<2, 128, 64, 2, 2>("3 x bf16, limbs in LDS, 128x64", dA, dB, dC, M, N, K, hA, hB);
  run<2, 64, 128, 2, 2>("3 x bf16, limbs in LDS, 64x128", dA, dB, dC, M, N, K, hA, hB);
  run<1, 64, 64, 1, 4>("3 x bf16, 64x64 as 1x4 waves", dA, dB, dC, M, N, K, hA, hB);
  run<1, 128, 128, 1, 4>("3 x bf16, 128x128 as 1x4 waves", dA, dB, dC, M, N, K, hA, hB);
  return 0;
}
