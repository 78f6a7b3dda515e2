// Host emulation of csrc/fft_wave.h (one wave = 64 lanes run step by step) against a direct double-precision DFT.
//   g++ -O2 -std=c++17 tools/fft_host_check.cpp -o /tmp/fft_host_check && /tmp/fft_host_check
#include <cmath>
#include <complex>
#include <cstdio>
#include <vector>
struct float2 { float x, y; };
#define FFT_HD inline
#include "../svs_unet_pytorch_amd/csrc/fft_wave.h"

template <int N>
double check() {
  using P = FftPlan<N>;
  using S = FftSize<N>;
  std::vector<float2> buf(S::BUF), tw(S::TW);
  for (int e = 0; e < S::TW; ++e) { const double a = M_PI * (double)fft_twiddle_angle<N>(e); tw[e] = float2{(float)cos(a), (float)sin(a)}; }
  std::vector<std::complex<double>> x(N);
  unsigned s = 12345u;
  for (int i = 0; i < N; ++i) {
    s = s * 1664525u + 1013904223u; const float re = (float)(s >> 8) / 8388608.0f - 1.0f;
    s = s * 1664525u + 1013904223u; const float im = (float)(s >> 8) / 8388608.0f - 1.0f;
    x[i] = {re, im};
    buf[fft_pad(i)] = float2{re, im};
  }
  auto run = [&](auto pass_proto, const float2* t) {
    using Pass = decltype(pass_proto);
    std::vector<Pass> lanes(64);
    for (int l = 0; l < 64; ++l) lanes[l].load(buf.data(), l);
    for (int l = 0; l < 64; ++l) lanes[l].compute(t, l);
    for (int l = 0; l < 64; ++l) lanes[l].store(buf.data(), l);
  };
  run(FftPass<N, P::R0, 1>{}, nullptr);
  run(FftPass<N, P::R1, S::NS1>{}, tw.data());
  run(FftPass<N, P::R2, S::NS2>{}, tw.data() + S::TW1);
  double worst = 0, scale = 0;
  for (int k = 0; k < N; ++k) {
    std::complex<double> acc = 0;
    for (int n = 0; n < N; ++n) acc += x[n] * std::polar(1.0, -2.0 * M_PI * (double)((long)k * n % N) / N);
    const float2 g = buf[fft_pad(k)];
    worst = std::max(worst, std::abs(acc - std::complex<double>(g.x, g.y)));
    scale = std::max(scale, std::abs(acc));
  }
  printf("N=%d  max |err| = %.3e  (max |X| = %.1f, rel %.2e)\n", N, worst, scale, worst / scale);
  return worst / scale;
}
int main() {
  const double e = std::max(check<512>(), std::max(check<1024>(), check<2048>()));
  return e < 2e-6 ? 0 : 1;
}
