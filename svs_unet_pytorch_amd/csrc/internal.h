// Internal (non-exported) launchers shared between translation units of libsvs_hip.so.
#pragma once
#include "common.h"

enum { SVS_MODE_GATHER = 0, SVS_MODE_PARITY = 1 };

int svs_conv_gemm_run(int mode, const float* x, long ldx, int B, int H, int W, int C, const float* wp,
                      const float* bias, const float* scale, const float* shift, float slope, float* y, long ldy,
                      int Ho, int Wo, int N, int accumulate, void* ws, size_t ws_bytes, hipStream_t stream,
                      const char* who, float* stats = nullptr, int stats_cap = 0, int* stats_nblk = nullptr);
// stats / stats_cap / stats_nblk: the kernel that writes the output (split-K epilogue, GEMM epilogue or window kernel) also
// writes svs_bn_stats-style partials of it into stats[rows][2][N] (stats_cap = capacity in floats) and reports the number
// of rows (0: not produced, e.g. the LDS-free direct kernel).
int svs_bn_finalize_run(const void* partial, int nblk, long P, int C, float eps, float momentum, float* running_mean,
                        float* running_var, long long* nbt, float* save_mean, float* save_invstd, hipStream_t stream);
// svs_bn_finalize + svs_bn_act_apply of `rows` partial rows; one launch when the rows are few enough for a block-local finalise
int svs_bn_fin_act_apply_run(const void* partial, int rows, const float* raw, long ldr, long P, int C, long pixels_per_sample,
                             const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                             float* running_var, long long* nbt, float* save_mean, float* save_invstd, float slope,
                             const float* drop, float* y, long ldy, hipStream_t stream);
struct SvsSumJobs { int njobs; const float* partial[12]; int nblk[12]; int C[12]; float* out[12]; };
int svs_channel_sum_finalize_multi_run(const SvsSumJobs& jobs, hipStream_t stream);
size_t svs_bn_partial_floats(long P, int C);     // capacity of a partial buffer
int svs_bn_partial_rows(long P, int C);         // rows svs_bn_stats writes
size_t svs_conv_gemm_workspace(int mode, int B, int H, int W, int C, int Ho, int Wo, int N);

int svs_wgrad_gemm_run(const float* s, long lds, int B, int Hs, int Ws, int Cs, const float* l, long ldl, int Hl,
                       int Wl, int Cl, float* dw, void* ws, size_t ws_bytes, hipStream_t stream, const char* who);
size_t svs_wgrad_gemm_workspace(int B, int Hs, int Ws, int Cs, int Cl);

int svs_conv_c1_run(const float* x, int B, int H, int W, const float* w, const float* bias, const float* scale,
                    const float* shift, float slope, float* y, long ldy, int N, int accumulate, hipStream_t stream,
                    const char* who, long half = 0);
int svs_deconv_to1_run(const float* x, long ldx, int B, int H, int W, int C, const float* w, const float* bias,
                       float* y, int Ho, int Wo, int apply_sigmoid, hipStream_t stream, const char* who, long half = 0);
int svs_wgrad_c1_run(const float* s, long lds, int B, int Hs, int Ws, int Cs, const float* l, int Hl, int Wl,
                     float* dw, void* ws, size_t ws_bytes, hipStream_t stream, const char* who, long half = 0);
size_t svs_wgrad_c1_workspace(int B, int Hs, int Ws, int Cs);

int svs_channel_sum_run(const float* x, long ldx, long P, int C, float* out, void* ws, size_t ws_bytes, hipStream_t stream);
int svs_sum_run(const float* x, long n, float* out, void* ws, size_t ws_bytes, hipStream_t stream);
int svs_sigmoid_bwd_run(const float* mask, const float* dmask, long n, float* d_logit, hipStream_t stream);

// out[g][i] = sum over the g-th chunk of `per` consecutive slabs of slab[z][i]   (i < n, g < groups)
int svs_reduce_slabs_run(const float* slab, int nslab, int per, int groups, long n, float* out, hipStream_t stream);

// one launch for several weight packings: kind 0 = gather packing of w[N][C][25], kind 1 = parity packing of w[C][N][25]
struct SvsPackJob { const float* w; float* wp; int N, C, kind, first_block; };
struct SvsPackJobs { SvsPackJob j[20]; int n; };
int svs_pack_all_run(SvsPackJobs& jobs, hipStream_t stream);

// svs_bn_bwd plus (dbias != NULL) the per-channel sum of d_raw = gradient of the conv bias in front of the BatchNorm
int svs_bn_bwd_run(const float* dy, long lddy, const float* raw, long ldr, long P, int C, long pixels_per_sample,
                   const float* gamma, const float* beta, const float* save_mean, const float* save_invstd, float slope,
                   const float* drop, float* d_raw, float* dgamma, float* dbeta, float* dbias, void* ws, size_t ws_bytes,
                   hipStream_t stream, float* dbias_partial = nullptr, SvsSumJobs* defer = nullptr);
// dbias_partial + defer: the per-block sums of d_raw go to dbias_partial (svs_bn_partial_floats(P, C) floats, must
// stay untouched until the deferred pass) and the final reduction into dbias is appended to *defer instead of being
// launched -- the caller runs svs_channel_sum_finalize_multi_run once for all layers.

// device pointer to fft_wave.h's twiddle table for n_fft = 512 / 1024 / 2048 (FftSize<n>::TW float2), built once per device;
// work queued on `stream` after the call sees the table complete
int svs_fft_twiddles(int n, hipStream_t stream, const float2** out);

int svs_conv_gemm_describe(int mode, int B, int H, int W, int C, int Ho, int Wo, int N, long ldx, char* buf, size_t n);
int svs_wgrad_gemm_describe(int B, int Hs, int Ws, int Cs, int Cl, char* buf, size_t n);

// float offsets of the pieces of the fp32 eval blob (svs_unet_prepare_eval): packed weights, folded BatchNorm scale / shift
void svs_unet_prepared_offsets(long wp[12], long scale[11], long shift[11], long* bias6);
