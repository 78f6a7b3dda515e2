// Probe: can VALU instructions of one wave issue while another wave on the same SIMD streams fp32 MFMAs
// (v_mfma_f32_16x16x4_f32)?  Each wave loops over { NM MFMAs ; NV v_fma } for ITER iterations; blocks of
// 256 threads (one wave per SIMD), 1 or 2 blocks per CU.  If VALU and fp32-MFMA co-issue across waves the
// 2-blocks-per-CU time is ~max(MFMA, VALU), otherwise ~sum.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_valu_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NM, int NV, bool SAME_WAVE_INTERLEAVE>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = a + i;
  for (int it = 0; it < iters; ++it) {
    if (SAME_WAVE_INTERLEAVE) {
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        acc[m & 15] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m & 15], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < NV / NM; ++k) v[(m + k) & 7] = __builtin_fmaf(v[(m + k) & 7], b, a);
      }
    } else {
#pragma unroll
      for (int m = 0; m < NM; ++m) acc[m & 15] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m & 15], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < NV; ++k) v[k & 7] = __builtin_fmaf(v[k & 7], b, a);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += v[i];
  if (s == 123.456f) out[0] = s;
}

template <int NM, int NV, bool IL>
static void run(const char* name, int blocks) {
  float* d;
  hipMalloc(&d, 4);
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  probe<NM, NV, IL><<<blocks, 256>>>(d, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<NM, NV, IL><<<blocks, 256>>>(d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double cyc_per_iter = ms * 1e-3 * 2.4e9 / iters;
  printf("%-34s blocks=%4d  NM=%3d NV=%3d  %8.3f ms  ~%7.0f cyc/iter @2.4GHz (MFMA alone = %d)\n", name, blocks, NM, NV, ms, cyc_per_iter, NM * 32);
  hipFree(d);
}

int main() {
  // one wave per SIMD (256 blocks = 1 per CU), then two and three (512 / 768 blocks)
  for (int blocks : {256, 512, 768}) {
    run<64, 0, false>("mfma only", blocks);
    run<0 + 1, 256, false>("valu only (256 fma)", blocks);
    run<64, 256, false>("64 mfma THEN 256 fma", blocks);
    run<64, 256, true>("64 mfma interleaved w/ 256 fma", blocks);
    run<64, 64, false>("64 mfma THEN 64 fma", blocks);
    run<64, 64, true>("64 mfma interleaved w/ 64 fma", blocks);
  }
  return 0;
}
