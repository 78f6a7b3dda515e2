// The twiddle tables of fft_wave.h (N = 512 / 1024 / 2048) built ONCE per device and kept in device memory, so that a transform
// block copies its table (4.6 / 10 / 18 KB, L2-resident) into LDS instead of evaluating 576 / 1280 / 2304 sincospif per block:
// with one or two transforms per wave that evaluation was 10-20 % of the VALU work of the STFT, iSTFT and MR-STFT kernels.
// The entries are the same sincospif values the blocks used to compute themselves: results are bitwise unchanged.
#include <mutex>

#include "internal.h"
#include "fft_wave.h"

#define FFT_TAB_TOTAL (FftSize<512>::TW + FftSize<1024>::TW + FftSize<2048>::TW)
__device__ __attribute__((aligned(16))) float2 g_fft_twiddles[FFT_TAB_TOTAL];

__global__ __launch_bounds__(256) void fft_tables_kernel() {
  const int tid = blockIdx.x * 256 + threadIdx.x, n = gridDim.x * 256;
  fft_build_twiddles<512>(g_fft_twiddles, tid, n);
  fft_build_twiddles<1024>(g_fft_twiddles + FftSize<512>::TW, tid, n);
  fft_build_twiddles<2048>(g_fft_twiddles + FftSize<512>::TW + FftSize<1024>::TW, tid, n);
}

// State per device: 0 = not built, 1 = build enqueued (other streams wait for `ready`), 2 = build seen complete.
namespace {
struct TabState { int state = 0; const float2* base = nullptr; hipEvent_t ready = nullptr; };
std::mutex g_mu;
TabState g_tab[64];
}

int svs_fft_twiddles(int n, hipStream_t stream, const float2** out) {
  SVS_REQUIRE(n == 512 || n == 1024 || n == 2048, "svs_fft_twiddles: no table for n_fft = %d", n);
  int dev = 0;
  SVS_HIP(hipGetDevice(&dev));
  SVS_REQUIRE(dev >= 0 && dev < 64, "svs_fft_twiddles: device index %d", dev);
  std::lock_guard<std::mutex> guard(g_mu);
  TabState& t = g_tab[dev];
  if (t.state != 2) {                 // (state 2 needs no stream operation at all and is safe inside a capture)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) {
      svs_set_error("svs_fft_twiddles: the first transform of a process must run outside a stream capture (it builds the twiddle tables)");
      return SVS_ERR_INVALID;
    }
    (void)hipGetLastError();
  }
  if (t.state == 0) {
    void* p = nullptr;
    SVS_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(g_fft_twiddles)));
    SVS_HIP(hipEventCreateWithFlags(&t.ready, hipEventDisableTiming));
    hipLaunchKernelGGL(fft_tables_kernel, dim3(8), dim3(256), 0, stream);
    SVS_CHECK_LAUNCH("fft_tables");
    SVS_HIP(hipEventRecord(t.ready, stream));
    t.base = (const float2*)p;
    t.state = 1;
  } else if (t.state == 1) {
    // built on some stream, possibly not this one: order this stream behind the build until the host has seen it complete
    if (hipEventQuery(t.ready) == hipSuccess) t.state = 2;
    else {
      (void)hipGetLastError();                     // (hipErrorNotReady is not an error of ours)
      SVS_HIP(hipStreamWaitEvent(stream, t.ready, 0));
    }
  }
  *out = t.base + (n == 512 ? 0 : n == 1024 ? FftSize<512>::TW : FftSize<512>::TW + FftSize<1024>::TW);
  return SVS_OK;
}
