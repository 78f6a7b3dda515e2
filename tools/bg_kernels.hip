// Background-load kernels for tools/stress_mr_concurrent.py (which instruction mix of a co-running kernel disturbs the MR-STFT kernels?)
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/bg_kernels.hip -o tools/bin/libbg_kernels.so
#include <hip/hip_runtime.h>
#include "../svs_unet_pytorch_amd/csrc/fft_wave.h"
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

// kind 0: bf16 MFMA only (registers, no LDS); 1: fp32 MFMA only; 2: LDS traffic only (48 KB static, ds_write/ds_read);
// 3: v_cvt_pk_bf16_f32 + VALU only; 4: bf16 MFMA + LDS
template <int KIND>
__global__ __launch_bounds__(256) void bg_kernel(float* out, int iters) {
  __shared__ float lds[12288];
  const int t = threadIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  bf16x8 a = {1, 2, 3, 4, 5, 6, 7, (short)t}, b = {8, 7, 6, 5, 4, 3, 2, (short)(t * 3)};
  float x = (float)t * 1e-3f, y = 1.0f;
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0 || KIND == 4) {
#pragma unroll
      for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
    }
    if (KIND == 1) {
#pragma unroll
      for (int k = 0; k < 8; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc, 0, 0, 0);
    }
    if (KIND == 2 || KIND == 4) {
#pragma unroll
      for (int k = 0; k < 8; ++k) lds[(t * 17 + k * 256 + i) % 12288] = x + k;
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 8; ++k) x += lds[(t * 5 + k * 311 + i) % 12288] * 1e-6f;
      __syncthreads();
    }
    if (KIND == 3) {
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
        typedef float f2 __attribute__((ext_vector_type(2)));
        const f2 v = {x, y};
        const unsigned pk = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf2));
        x = x * 1.0001f + __builtin_bit_cast(float, pk << 16) * 1e-7f;
        y = y * 0.9999f + __builtin_bit_cast(float, pk & 0xffff0000u) * 1e-7f;
      }
    }
  }
  out[blockIdx.x * 256 + t] = acc[0] + acc[1] + acc[2] + acc[3] + x + y;
}
extern "C" int bg_launch(int kind, float* out, int blocks, int iters, hipStream_t s) {
  switch (kind) {
    case 0: hipLaunchKernelGGL(bg_kernel<0>, dim3(blocks), dim3(256), 0, s, out, iters); break;
    case 1: hipLaunchKernelGGL(bg_kernel<1>, dim3(blocks), dim3(256), 0, s, out, iters); break;
    case 2: hipLaunchKernelGGL(bg_kernel<2>, dim3(blocks), dim3(256), 0, s, out, iters); break;
    case 3: hipLaunchKernelGGL(bg_kernel<3>, dim3(blocks), dim3(256), 0, s, out, iters); break;
    default: hipLaunchKernelGGL(bg_kernel<4>, dim3(blocks), dim3(256), 0, s, out, iters); break;
  }
  return (int)hipGetLastError();
}

// ---- victims: one instruction class each; out[] must be bitwise identical run to run, whatever else the GPU is doing ----------
// 0 v_fma_f32   1 packed fp32 (v_pk_fma / v_pk_mul / v_pk_add)   2 v_sqrt / v_rcp / v_log / v_cos   3 wave-level LDS exchange without
// barriers (the wave FFT's pattern)   4 cross-lane shuffles   5 fp64 fma   6 sincospif   7 global loads + adds
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ __launch_bounds__(512) void victim_kernel(const float* in, float* out, int iters) {
  __shared__ float2 buf[8][1088];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  float x = in[(blockIdx.x * 512 + t) & 65535], y = 0.5f + 1e-3f * lane;
  f32x2 p = {x, y}, q = {y * 0.999f, x * 1.001f};
  double dx = x;
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) { x = x * 0.999f + y * 1e-3f; y = y * 1.0001f - x * 1e-4f; }
    if (KIND == 1) { p = p * q + (f32x2){1e-3f, -1e-3f}; q = q * (f32x2){0.9999f, 1.0001f} + p * (f32x2){1e-5f, 1e-5f}; }
    if (KIND == 2) {
      const float a = fabsf(x) + 0.25f;
      x = __builtin_amdgcn_sqrtf(a) * __builtin_amdgcn_rcpf(a + 1.f) + 1e-3f * __builtin_amdgcn_logf(a) + 1e-3f * __builtin_amdgcn_cosf(a * 0.1f);
    }
    if (KIND == 3) {
      buf[wave][lane + 64 * (i & 15) + ((lane + 64 * (i & 15)) >> 4)] = float2{x, y};
      const int j = (lane * 7 + 13) & 63;
      const float2 v = buf[wave][j + 64 * (i & 15) + ((j + 64 * (i & 15)) >> 4)];
      x = x * 0.5f + v.x * 0.5f + 1e-3f; y = y * 0.5f + v.y * 0.5f;
    }
    if (KIND == 4) { x = x * 0.5f + __shfl_xor(x, 1 + (i & 31), 64) * 0.5f + 1e-3f; }
    if (KIND == 5) { dx = dx * 0.999 + 1e-3 * (double)y; }
    if (KIND == 6) { float s, c; sincospif(x * 0.01f, &s, &c); x = x * 0.9f + s * 0.1f + c * 1e-2f; }
    if (KIND == 7) { x = x * 0.999f + in[(t * 17 + i * 4099 + blockIdx.x) & 65535] * 1e-3f; }
  }
  out[blockIdx.x * 512 + t] = x + y + p[0] + p[1] + q[0] + q[1] + (float)dx;
}
extern "C" int victim_launch(int kind, const float* in, float* out, int blocks, int iters, hipStream_t s) {
#define V(K) case K: hipLaunchKernelGGL(victim_kernel<K>, dim3(blocks), dim3(512), 0, s, in, out, iters); break;
  switch (kind) { V(0) V(1) V(2) V(3) V(4) V(5) V(6) default: hipLaunchKernelGGL(victim_kernel<7>, dim3(blocks), dim3(512), 0, s, in, out, iters); break; }
#undef V
  return (int)hipGetLastError();
}

// ---- barrier victims: 512-thread blocks, cross-wave exchange through LDS with __syncthreads(); static 32 KB or dynamic (any size)
template <bool DYN>
__global__ __launch_bounds__(512) void barrier_victim(const float* in, float* out, int iters, int words) {
  extern __shared__ float dyn[];
  __shared__ float stat[8192];
  float* lds = DYN ? dyn : stat;
  const int n = DYN ? words : 8192;
  const int t = threadIdx.x;
  float x = in[(blockIdx.x * 512 + t) & 65535];
  for (int i = 0; i < iters; ++i) {
    lds[(t * 16 + (i & 15)) % n] = x;                       // spread over the whole allocation
    if (DYN) lds[n - 1 - t] = x * 0.5f;                     // and its far end (beyond 64 KB when the allocation is larger)
    __syncthreads();
    const int u = (t + 64 + 7 * (i & 7)) & 511;             // another wave's slot
    x = x * 0.75f + 0.25f * lds[(u * 16 + (i & 15)) % n] + (DYN ? 0.01f * lds[n - 1 - u] : 0.f) + 1e-3f;
    __syncthreads();
  }
  out[blockIdx.x * 512 + t] = x;
}
extern "C" int barrier_victim_launch(int dyn_bytes, const float* in, float* out, int blocks, int iters, hipStream_t s) {
  if (dyn_bytes > 0) {
    hipFuncSetAttribute((const void*)barrier_victim<true>, hipFuncAttributeMaxDynamicSharedMemorySize, dyn_bytes);
    hipLaunchKernelGGL(barrier_victim<true>, dim3(blocks), dim3(512), dyn_bytes, s, in, out, iters, dyn_bytes / 4);
  } else hipLaunchKernelGGL(barrier_victim<false>, dim3(blocks), dim3(512), 0, s, in, out, iters, 0);
  return (int)hipGetLastError();
}

// ---- the library's wave FFT in isolation: MODE 0 dumps the twiddle table each block builds; 1 transforms a fixed pattern
template <int MODE>
__global__ __launch_bounds__(512) void fft_victim(float2* out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int N = 1024, BUF = FftSize<N>::BUF, TW = FftSize<N>::TW;
  float2* const fbuf = (float2*)smem;
  float2* const tw = fbuf + 8 * BUF;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  fft_build_twiddles<N>(tw, tid, 512);
  float2* const buf = fbuf + wave * BUF;
  for (int r = 0; r < N / 64; ++r) {
    const int m = lane + 64 * r;
    buf[fft_pad(m)] = float2{(float)((m * 37 + wave * 11 + blockIdx.x) & 255) * 0.01f - 1.f, (float)((m * 53 + wave) & 127) * 0.02f - 1.f};
  }
  __syncthreads();
  if (MODE == 0) {
    for (int e = tid; e < TW; e += 512) out[(long)blockIdx.x * TW + e] = tw[e];
    return;
  }
  fft_wave<N>(buf, tw, lane);
  for (int r = 0; r < N / 64; ++r) {
    const int m = lane + 64 * r;
    out[((long)blockIdx.x * 8 + wave) * N + m] = buf[fft_pad(m)];
  }
}
extern "C" int fft_victim_launch(int mode, float* out, int blocks, hipStream_t s) {
  const int lds = (8 * FftSize<1024>::BUF + FftSize<1024>::TW) * 8 + 128;
  if (mode == 0) {
    hipFuncSetAttribute((const void*)fft_victim<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(fft_victim<0>, dim3(blocks), dim3(512), lds, s, (float2*)out);
  } else {
    hipFuncSetAttribute((const void*)fft_victim<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(fft_victim<1>, dim3(blocks), dim3(512), lds, s, (float2*)out);
  }
  return (int)hipGetLastError();
}

// ---- LDS-overlap detector: every block signs all of its dynamic LDS, dwells, and checks the signature ----------------------------
__global__ __launch_bounds__(512) void lds_guard_kernel(unsigned* report, int words, int dwell) {
  extern __shared__ unsigned gl[];
  const int t = threadIdx.x;
  const unsigned sig = 0xA5000000u | (blockIdx.x << 4);
  for (int i = t; i < words; i += 512) gl[i] = sig ^ (unsigned)i * 2654435761u;
  __syncthreads();
  float x = (float)t;
  for (int i = 0; i < dwell; ++i) x = x * 1.0001f + 0.5f;            // dwell so that other blocks come and go
  __syncthreads();
  unsigned bad = 0, first = 0, firstpos = 0;
  for (int i = t; i < words; i += 512) {
    const unsigned v = gl[i];
    if (v != (sig ^ (unsigned)i * 2654435761u)) { if (!bad) { first = v; firstpos = i; } ++bad; }
  }
  if (bad) {
    const unsigned slot = atomicAdd(&report[0], 1u);
    atomicAdd(&report[1], bad);
    if (slot < 64) { report[4 + slot * 4] = blockIdx.x; report[5 + slot * 4] = firstpos; report[6 + slot * 4] = first; report[7 + slot * 4] = bad; }
  }
  if (x == 12345.678f) report[2] = 1;                                // keeps the dwell loop
}
extern "C" int lds_guard_launch(int bytes, unsigned* report, int blocks, int dwell, hipStream_t s) {
  hipFuncSetAttribute((const void*)lds_guard_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  hipLaunchKernelGGL(lds_guard_kernel, dim3(blocks), dim3(512), bytes, s, report, bytes / 4, dwell);
  return (int)hipGetLastError();
}

// ---- packed-fp32 forms in isolation (inline asm so that the compiler cannot pick another form) ------------------------------------
// 0 v_pk_add_f32   1 v_pk_add_f32 neg_lo/neg_hi on src1 (a - b)   2 v_pk_mul_f32   3 v_pk_mul_f32 op_sel (cross)   4 v_pk_fma_f32
// 5 v_pk_fma_f32 op_sel + neg   6 v_pk_mov_b32   7 scalar v_add/v_mul/v_fma of the same chain (control)
typedef float pkf2 __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ __launch_bounds__(512) void pk_victim(const float* in, float* out, int iters) {
  const int t = threadIdx.x;
  pkf2 a = {in[(blockIdx.x * 512 + t) & 65535], in[(blockIdx.x * 512 + t + 77) & 65535]};
  pkf2 b = {0.999f + 1e-6f * t, 1.001f - 1e-6f * t}, c = {1e-3f, -1e-3f};
  const pkf2 sb = {0.9995f, 1.0005f};
  for (int i = 0; i < iters; ++i) {
    if (KIND == 10) { a[0] *= 0.999f; a[1] *= 0.999f; asm volatile("" : "+v"(a)); }
    if (KIND == 0) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a) : "v"(c));
    if (KIND == 1) asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(a) : "v"(c));
    if (KIND == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a) : "v"(b));
    if (KIND == 3) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(a) : "v"(b));
    if (KIND == 4) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    if (KIND == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,1,0] op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,1,0]" : "+v"(a) : "v"(b), "v"(c));
    if (KIND == 6) { pkf2 d; asm volatile("v_pk_mov_b32 %0, %1, %1 op_sel:[1,0]" : "=v"(d) : "v"(a)); a = d * 0.5f + a * 0.5f; }
    if (KIND == 8) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(a) : "v"(b));                     // broadcast src1.lo
    if (KIND == 9) asm volatile("v_pk_mul_f32 %0, %1, %0 op_sel_hi:[0,1]" : "+v"(a) : "s"(sb));                    // broadcast an SGPR pair's lo
    if (KIND == 10) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(a) : "v"(c));
    if (KIND == 11) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[0,1]" : "+v"(a) : "v"(b));                       // op_sel alone (lo half takes src1.hi)
    if (KIND == 12) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "+v"(a) : "v"(b), "v"(c));   // neg only
    if (KIND == 7) { a[0] = a[0] * b[0] + c[0]; a[1] = a[1] * b[1] + c[1]; asm volatile("" : "+v"(a)); }
    if (KIND <= 1) { a[0] *= 0.999f; a[1] *= 0.999f; asm volatile("" : "+v"(a)); }      // keep the values bounded (scalar ops)
  }
  out[(blockIdx.x * 512 + t) * 2] = a[0];
  out[(blockIdx.x * 512 + t) * 2 + 1] = a[1];
}
extern "C" int pk_victim_launch(int kind, const float* in, float* out, int blocks, int iters, hipStream_t s) {
#define V(K) case K: hipLaunchKernelGGL(pk_victim<K>, dim3(blocks), dim3(512), 0, s, in, out, iters); break;
  switch (kind) { V(0) V(1) V(2) V(3) V(4) V(5) V(6) V(8) V(9) V(10) V(11) V(12) default: hipLaunchKernelGGL(pk_victim<7>, dim3(blocks), dim3(512), 0, s, in, out, iters); break; }
#undef V
  return (int)hipGetLastError();
}
