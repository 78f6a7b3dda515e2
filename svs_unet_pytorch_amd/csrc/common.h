// Shared host/device helpers for libsvs_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/svs_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- error reporting (thread-local, see svs_last_error_string) -------------------------------
void svs_set_error(const char* fmt, ...);

#define SVS_REQUIRE(cond, ...)                 \
  do {                                         \
    if (!(cond)) {                             \
      svs_set_error(__VA_ARGS__);              \
      return SVS_ERR_INVALID;                  \
    }                                          \
  } while (0)

#define SVS_CHECK_LAUNCH(name)                                                     \
  do {                                                                             \
    hipError_t e_ = hipGetLastError();                                             \
    if (e_ != hipSuccess) {                                                        \
      svs_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));         \
      return (int)e_;                                                              \
    }                                                                              \
  } while (0)

#define SVS_HIP(call)                                                              \
  do {                                                                             \
    hipError_t e_ = (call);                                                        \
    if (e_ != hipSuccess) {                                                        \
      svs_set_error("%s failed: %s", #call, hipGetErrorString(e_));                \
      return (int)e_;                                                              \
    }                                                                              \
  } while (0)

static inline bool svs_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }
static inline int svs_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline size_t svs_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- tuning switches (sweeps, A/B runs and tests only) ------------------------------------------
// Every planner override lives in ONE table that is filled from the environment (SVS_<NAME>) once, at the first
// use, and can afterwards only be changed through svs_tuning_set() (include/svs_hip.h) -- no getenv() in any call
// path.  -1 = not set: the planner decides.
enum SvsTune {
  SVS_TUNE_CONV_CFG, SVS_TUNE_CONV_KSPLIT, SVS_TUNE_CONV_WINDOW, SVS_TUNE_CONV_SKIP, SVS_TUNE_CONV_KORDER,
  SVS_TUNE_CONV_DIRECT, SVS_TUNE_SKIP_REDUCE, SVS_TUNE_WGRAD_CFG, SVS_TUNE_WGRAD_KSPLIT, SVS_TUNE_WGRAD_SKIP,
  SVS_TUNE_WGRAD_WINDOW, SVS_TUNE_WGRAD_C1_VALU, SVS_TUNE_SIDE_PRIORITY, SVS_TUNE_TRAIN_UNFUSED,
  SVS_TUNE_TRAIN_ONE_STREAM, SVS_TUNE_CONV_PLAN, SVS_TUNE_MFMA_SPLIT, SVS_TUNE_CONV_BALANCE, SVS_TUNE_CONV_C1_TILED, SVS_TUNE_BF16_KB, SVS_TUNE_BF16_CFG, SVS_TUNE_BF16_KSPLIT, SVS_TUNE_CONV_PF, SVS_TUNE_WGRAD_PF, SVS_TUNE_BN_INLINE, SVS_TUNE_BN_BLOCKS, SVS_TUNE_BF16_CONV3_WINDOW, SVS_TUNE_BF16_DECONV3_WINDOW, SVS_TUNE_CONV_GWINDOW, SVS_TUNE_COUNT
};
long svs_tune(int key);                       // -1 when unset
static inline bool svs_tune_on(int key) { return svs_tune(key) >= 0; }      // VALUED switches: "has been set" (0 is a value)
static inline bool svs_tune_flag(int key) { return svs_tune(key) > 0; }     // BOOLEAN switches: 0 and -1 both mean off

// geometry of the 5x5 / stride 2 / pad 2 layers (reference model.py:48,79: kernel (5,5), stride (2,2), padding 2)
static inline int svs_conv_out(int n) { return (n + 1) / 2; }  // floor((n+4-5)/2)+1

// ---- device helpers -------------------------------------------------------------------------
__device__ __forceinline__ float svs_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// the counter-based generator of svs_unet_pytorch_amd/synth.py
__host__ __device__ __forceinline__ uint32_t svs_mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  return x;
}
__host__ __device__ __forceinline__ uint32_t svs_u32(uint32_t seed, uint64_t idx) {
  uint32_t lo = (uint32_t)idx, hi = (uint32_t)(idx >> 32);
  uint32_t s = svs_mix32(seed + 1u);
  uint32_t h = svs_mix32(hi + 0x85EBCA6Bu * s);
  return svs_mix32(lo + 0x9E3779B9u * h);
}
