// fp32-accurate 16x16x16 products on the bf16 MFMA pipe (gfx950) -- optional mode of the fp32 GEMM kernels (SVS_MFMA_SPLIT=1 /
// svs_tuning_set("MFMA_SPLIT", 1)); default off.  See DESIGN.md section 5 and tools/split_bf16_probe.hip.
//
// An fp32 value splits EXACTLY into three bf16 limbs, x = x0 + x1 + x2 (8 significant bits each: round to bf16, subtract,
// twice).  Of the nine limb products the six above 2^-24 relative are kept: a0b0, a0b1, a1b0, a1b1, a0b2, a2b0.  A lane of the
// fp32 kernels holds 4 consecutive k of one row per 16-deep K-tile (an f32x4 fragment); v_mfma_f32_16x16x32_bf16 takes 8 bf16
// per lane, so ONE instruction sums two limb products over those 4 k: with the operand pairs
//     A = [a1|a0]  B = [b1|b0]   ->  a1 b1 + a0 b0
//     A = [a1|a0]  B = [b0|b1]   ->  a1 b0 + a0 b1
//     A = [a0|a2]  B = [b2|b0]   ->  a0 b2 + a2 b0
// three MFMAs of 16 cycles replace four fp32 MFMAs of 32, on the same LDS tiles, fragment reads and accumulator layout.
// Measured error against float64 (error / sum |a b|, operands over four decades): max 2.4e-7, mean 2.9e-8 -- the fp32 MFMA
// itself: 3.3e-7 / 3.4e-8.
#pragma once

typedef short svs_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 svs_bf16x2 __attribute__((ext_vector_type(2)));
typedef float svs_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned svs_u32x4 __attribute__((ext_vector_type(4)));

struct SvsLimbs { unsigned p[3][2]; };            // p[l][h]: limb l of elements (2h, 2h + 1), packed bf16 pair
__device__ __forceinline__ unsigned svs_cvt_pk_bf16(float a, float b) {          // v_cvt_pk_bf16_f32, round to nearest even
  const svs_f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, svs_bf16x2));
}
__device__ __forceinline__ SvsLimbs svs_split3(float x0, float x1, float x2, float x3) {
  SvsLimbs L;
  float r[4] = {x0, x1, x2, x3};
#pragma unroll
  for (int l = 0; l < 3; ++l)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const unsigned pk = svs_cvt_pk_bf16(r[2 * h], r[2 * h + 1]);
      L.p[l][h] = pk;
      if (l < 2) {                                                                // exact: the limb is the leading part of r
        r[2 * h] -= __builtin_bit_cast(float, pk << 16);
        r[2 * h + 1] -= __builtin_bit_cast(float, pk & 0xffff0000u);
      }
    }
  return L;
}
__device__ __forceinline__ svs_bf16x8 svs_limb_pair(const SvsLimbs& L, int lo, int hi) {
  const svs_u32x4 v = {L.p[lo][0], L.p[lo][1], L.p[hi][0], L.p[hi][1]};
  return __builtin_bit_cast(svs_bf16x8, v);
}
struct SvsSplitA { svs_bf16x8 a10, a02; };
struct SvsSplitB { svs_bf16x8 b10, b01, b20; };
__device__ __forceinline__ SvsSplitA svs_split_a(float x0, float x1, float x2, float x3) {
  const SvsLimbs L = svs_split3(x0, x1, x2, x3);
  return SvsSplitA{svs_limb_pair(L, 1, 0), svs_limb_pair(L, 0, 2)};
}
__device__ __forceinline__ SvsSplitB svs_split_b(float x0, float x1, float x2, float x3) {
  const SvsLimbs L = svs_split3(x0, x1, x2, x3);
  return SvsSplitB{svs_limb_pair(L, 1, 0), svs_limb_pair(L, 0, 1), svs_limb_pair(L, 2, 0)};
}
// acc[i][j] += A_i B_j^T over the 16 k of this K-tile; small terms first
template <int TM, int TN, class Acc>
__device__ __forceinline__ void svs_mma_split(Acc (&acc)[TM][TN], const SvsSplitA (&a)[TM], const SvsSplitB (&b)[TN]) {
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i].a02, b[j].b20, acc[i][j], 0, 0, 0);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i].a10, b[j].b01, acc[i][j], 0, 0, 0);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i].a10, b[j].b10, acc[i][j], 0, 0, 0);
}
