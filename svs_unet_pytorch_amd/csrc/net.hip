// Host-side orchestration of the U-Net on one GPU: the per-block C-ABI wrappers and the whole-network
// eval forward / training forward+backward (one C call per step; every kernel goes to the caller's
// stream, nothing synchronises, so the calls are hipGraph-capturable).
//
// HBM layout (fp32, NHWC).  Level k (k = 0..6) has spatial size (h[k], w[k]) = repeated ceil-halving
// of the input and ch[k] = {1,16,32,64,128,256,512} channels (reference model.py:47-76).
//   cat[k], k=1..5 : (B, h[k], w[k], 2*ch[k])   first half  = output of decoder 6-k  (model.py:183-196)
//                                                second half = output of encoder k    (model.py:176-180)
//                    -> torch.cat([deconv_out, conv_out], 1) of model.py:186-198 is never materialised:
//                       producers write their half with ld = 2*ch[k], consumers read the full width.
//   c6             : (B, h[6], w[6], 512)
// Training additionally keeps the pre-BatchNorm ("raw") output of every block, the batch statistics,
// both weight packings, and gradient images dcat[k] / dc6 of the same shapes.
#include <math.h>
#include <atomic>
#include <mutex>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include "internal.h"

// ---------------------------------------------------------------------------------------------
// error string
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void svs_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* svs_last_error_string(void) { return g_err; }

// ---------------------------------------------------------------------------------------------
// tuning table (common.h: SvsTune)
// ---------------------------------------------------------------------------------------------
static const char* const TUNE_NAMES[SVS_TUNE_COUNT] = {
    "CONV_CFG", "CONV_KSPLIT", "CONV_WINDOW", "CONV_SKIP", "CONV_KORDER", "CONV_DIRECT", "SKIP_REDUCE", "WGRAD_CFG",
    "WGRAD_KSPLIT", "WGRAD_SKIP", "WGRAD_WINDOW", "WGRAD_C1_VALU", "SIDE_PRIORITY", "TRAIN_UNFUSED", "TRAIN_ONE_STREAM", "CONV_PLAN", "MFMA_SPLIT", "CONV_BALANCE", "CONV_C1_TILED", "BF16_KB", "BF16_CFG", "BF16_KSPLIT", "CONV_PF", "WGRAD_PF", "BN_INLINE", "BN_BLOCKS", "BF16_CONV3_WINDOW", "BF16_DECONV3_WINDOW", "CONV_GWINDOW"};
static std::atomic<long> g_tune[SVS_TUNE_COUNT];      // written by svs_tuning_set while compute threads read: relaxed atomics
static std::once_flag g_tune_once;
static void tune_load_env() {
  for (int k = 0; k < SVS_TUNE_COUNT; ++k) {
    char name[64];
    snprintf(name, sizeof(name), "SVS_%s", TUNE_NAMES[k]);
    const char* e = getenv(name);
    g_tune[k].store(e ? (*e ? atol(e) : 1) : -1, std::memory_order_relaxed);
  }
}
long svs_tune(int key) {
  std::call_once(g_tune_once, tune_load_env);
  return (key >= 0 && key < SVS_TUNE_COUNT) ? g_tune[key].load(std::memory_order_relaxed) : -1;
}
extern "C" int svs_tuning_set(const char* name, long value) {
  std::call_once(g_tune_once, tune_load_env);
  SVS_REQUIRE(name, "svs_tuning_set: null name");
  if (!strcmp(name, "*")) {                 // every switch: value -1 = back to the process defaults (the SVS_<NAME> environment)
    if (value == -1) tune_load_env();
    else for (int k = 0; k < SVS_TUNE_COUNT; ++k) g_tune[k].store(value, std::memory_order_relaxed);
    return SVS_OK;
  }
  for (int k = 0; k < SVS_TUNE_COUNT; ++k)
    if (!strcmp(name, TUNE_NAMES[k])) { g_tune[k].store(value, std::memory_order_relaxed); return SVS_OK; }
  svs_set_error("svs_tuning_set: unknown switch '%s'", name);
  return SVS_ERR_INVALID;
}
extern "C" int svs_version(void) { return SVS_ABI_VERSION; }

// ---------------------------------------------------------------------------------------------
// per-block wrappers
// ---------------------------------------------------------------------------------------------
extern "C" size_t svs_enc_block_workspace_bytes(int B, int H, int W, int C, int N) {
  return svs_conv_gemm_workspace(SVS_MODE_GATHER, B, H, W, C, svs_conv_out(H), svs_conv_out(W), N);
}
extern "C" int svs_enc_block_fwd(const float* x, int64_t ldx, int B, int H, int W, int C, const float* wp,
                                 const float* bias, const float* scale, const float* shift, float slope, float* y,
                                 int64_t ldy, int N, int accumulate, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (C == 1) {
    SVS_REQUIRE(ldx == 1, "svs_enc_block_fwd: single-channel input must be dense (ldx=1)");
    return svs_conv_c1_run(x, B, H, W, wp, bias, scale, shift, slope, y, ldy, N, accumulate, stream, "svs_enc_block_fwd");
  }
  return svs_conv_gemm_run(SVS_MODE_GATHER, x, ldx, B, H, W, C, wp, bias, scale, shift, slope, y, ldy, svs_conv_out(H),
                           svs_conv_out(W), N, accumulate, ws, ws_bytes, stream, "svs_enc_block_fwd");
}
extern "C" size_t svs_dec_block_workspace_bytes(int B, int H, int W, int C, int Ho, int Wo, int N) {
  return svs_conv_gemm_workspace(SVS_MODE_PARITY, B, H, W, C, Ho, Wo, N);
}
extern "C" int svs_dec_block_fwd(const float* x, int64_t ldx, int B, int H, int W, int C, const float* wp,
                                 const float* bias, const float* scale, const float* shift, float slope, float* y,
                                 int64_t ldy, int Ho, int Wo, int N, int accumulate, void* ws, size_t ws_bytes,
                                 hipStream_t stream) {
  return svs_conv_gemm_run(SVS_MODE_PARITY, x, ldx, B, H, W, C, wp, bias, scale, shift, slope, y, ldy, Ho, Wo, N,
                           accumulate, ws, ws_bytes, stream, "svs_dec_block_fwd");
}
extern "C" int svs_out_block_fwd(const float* x, int64_t ldx, int B, int H, int W, int C, const float* w,
                                 const float* bias, float* y, int Ho, int Wo, int apply_sigmoid, hipStream_t stream) {
  return svs_deconv_to1_run(x, ldx, B, H, W, C, w, bias, y, Ho, Wo, apply_sigmoid, stream, "svs_out_block_fwd");
}
extern "C" int svs_enc_block_bwd_data(const float* dy, int64_t lddy, int B, int Ho, int Wo, int N, const float* wpar,
                                      float* dx, int64_t lddx, int H, int W, int C, int accumulate, void* ws,
                                      size_t ws_bytes, hipStream_t stream) {
  // transposed conv of dy: "input" is (Ho,Wo,N), "output" is (H,W,C)
  return svs_conv_gemm_run(SVS_MODE_PARITY, dy, lddy, B, Ho, Wo, N, wpar, nullptr, nullptr, nullptr, 0.f, dx, lddx, H, W,
                           C, accumulate, ws, ws_bytes, stream, "svs_enc_block_bwd_data");
}
extern "C" int svs_dec_block_bwd_data(const float* dy, int64_t lddy, int B, int Ho, int Wo, int N, const float* wgat,
                                      float* dx, int64_t lddx, int H, int W, int C, int accumulate, void* ws,
                                      size_t ws_bytes, hipStream_t stream) {
  // strided conv of dy: "input" is (Ho,Wo,N), "output" is (H,W,C) with H = ceil(Ho/2)
  if (N == 1) {
    SVS_REQUIRE(lddy == 1, "svs_dec_block_bwd_data: single-channel gradient must be dense");
    SVS_REQUIRE(H == svs_conv_out(Ho) && W == svs_conv_out(Wo), "svs_dec_block_bwd_data: geometry mismatch");
    return svs_conv_c1_run(dy, B, Ho, Wo, wgat, nullptr, nullptr, nullptr, 0.f, dx, lddx, C, accumulate, stream,
                           "svs_dec_block_bwd_data");
  }
  return svs_conv_gemm_run(SVS_MODE_GATHER, dy, lddy, B, Ho, Wo, N, wgat, nullptr, nullptr, nullptr, 0.f, dx, lddx, H, W,
                           C, accumulate, ws, ws_bytes, stream, "svs_dec_block_bwd_data");
}
extern "C" size_t svs_block_bwd_weight_workspace_bytes(int B, int Hs, int Ws, int Cs, int Cl) {
  size_t a = (Cl == 1) ? svs_wgrad_c1_workspace(B, Hs, Ws, Cs) : svs_wgrad_gemm_workspace(B, Hs, Ws, Cs, Cl);
  size_t b = svs_bn_workspace_bytes((int64_t)B * Hs * Ws * 4, Cs > Cl ? Cs : Cl);   // bias-gradient partials
  return a > b ? a : b;
}
extern "C" int svs_enc_block_bwd_weight(const float* dy, int64_t lddy, int B, int Ho, int Wo, int N, const float* x,
                                        int64_t ldx, int H, int W, int C, float* dw, float* db, void* ws,
                                        size_t ws_bytes, hipStream_t stream) {
  int rc;
  if (C == 1) rc = svs_wgrad_c1_run(dy, lddy, B, Ho, Wo, N, x, H, W, dw, ws, ws_bytes, stream, "svs_enc_block_bwd_weight");
  else rc = svs_wgrad_gemm_run(dy, lddy, B, Ho, Wo, N, x, ldx, H, W, C, dw, ws, ws_bytes, stream, "svs_enc_block_bwd_weight");
  if (rc || !db) return rc;
  return svs_channel_sum_run(dy, lddy, (long)B * Ho * Wo, N, db, ws, ws_bytes, stream);
}
extern "C" int svs_dec_block_bwd_weight(const float* x, int64_t ldx, int B, int H, int W, int C, const float* dy,
                                        int64_t lddy, int Ho, int Wo, int N, float* dw, float* db, void* ws,
                                        size_t ws_bytes, hipStream_t stream) {
  int rc;
  if (N == 1) {
    rc = svs_wgrad_c1_run(x, ldx, B, H, W, C, dy, Ho, Wo, dw, ws, ws_bytes, stream, "svs_dec_block_bwd_weight");
    if (rc || !db) return rc;
    return svs_sum_run(dy, (long)B * Ho * Wo, db, ws, ws_bytes, stream);
  }
  rc = svs_wgrad_gemm_run(x, ldx, B, H, W, C, dy, lddy, Ho, Wo, N, dw, ws, ws_bytes, stream, "svs_dec_block_bwd_weight");
  if (rc || !db) return rc;
  return svs_channel_sum_run(dy, lddy, (long)B * Ho * Wo, N, db, ws, ws_bytes, stream);
}

// kind 0: gather GEMM (enc fwd / dec bwd_data), 1: parity GEMM (dec fwd / enc bwd_data), 2: weight-gradient GEMM
// (then H,W,C = the strided image S and N = channels of the windowed image).  Returns the K-split.
extern "C" int svs_describe_plan(int kind, int B, int H, int W, int C, int Ho, int Wo, int N, char* buf, size_t buflen) {
  if (!buf || !buflen) return SVS_ERR_INVALID;
  if (kind == 2) return svs_wgrad_gemm_describe(B, H, W, C, N, buf, buflen);
  return svs_conv_gemm_describe(kind == 1 ? SVS_MODE_PARITY : SVS_MODE_GATHER, B, H, W, C, Ho, Wo, N, C, buf, buflen);
}

// ---------------------------------------------------------------------------------------------
// network description
// ---------------------------------------------------------------------------------------------
static const int CH[7] = {1, 16, 32, 64, 128, 256, 512};                 // model.py:47-76
static const int DEC_C[6] = {512, 512, 256, 128, 64, 32};                // model.py:79-109 (in)
static const int DEC_N[6] = {256, 128, 64, 32, 16, 1};                   //                 (out)
#define BN_EPS 1e-5f
#define BN_MOMENTUM 0.1f
#define LEAKY 0.2f

// parameter tensor index: enc k (1..6): 4*(k-1) + {0 w, 1 b, 2 gamma, 3 beta}; dec j (1..6): 24 + 4*(j-1) + {...}
static long param_numel(int idx) {
  if (idx < 24) {
    const int k = idx / 4 + 1, f = idx % 4;
    return f == 0 ? (long)CH[k] * CH[k - 1] * 25 : CH[k];
  }
  const int j = (idx - 24) / 4, f = (idx - 24) % 4;
  return f == 0 ? (long)DEC_C[j] * DEC_N[j] * 25 : DEC_N[j];
}
extern "C" int64_t svs_unet_param_offset(int tensor_index) {
  if (tensor_index < 0 || tensor_index > SVS_UNET_NUM_PARAMS) return -1;
  long off = 0;
  for (int i = 0; i < tensor_index; ++i) off += param_numel(i);
  return off;
}
static int bn_channels(int bn) { return bn < 6 ? CH[bn + 1] : DEC_N[bn - 6]; }
extern "C" int64_t svs_unet_buffer_offset(int bn_index, int which) {
  if (bn_index < 0 || bn_index > SVS_UNET_NUM_BN) return -1;
  long off = 0;
  for (int i = 0; i < bn_index; ++i) off += 2 * bn_channels(i);
  if (bn_index < SVS_UNET_NUM_BN && which) off += bn_channels(bn_index);
  return off;
}

struct ParamView {
  const float* w[12]; const float* b[12]; const float* gamma[11]; const float* beta[11];   // 0..5 enc, 6..11 dec
};
static ParamView view_params(const float* params) {
  ParamView v{};
  for (int l = 0; l < 12; ++l) {
    const int base = 4 * l;
    v.w[l] = params + svs_unet_param_offset(base);
    v.b[l] = params + svs_unet_param_offset(base + 1);
    if (l < 11) {
      v.gamma[l] = params + svs_unet_param_offset(base + 2);
      v.beta[l] = params + svs_unet_param_offset(base + 3);
    }
  }
  return v;
}

struct Geo { int B; int h[7], w[7]; long P[7]; };
static int make_geo(int B, int H, int W, Geo& g) {
  SVS_REQUIRE(B > 0 && H > 0 && W > 0, "bad tile geometry B=%d H=%d W=%d", B, H, W);
  g.B = B; g.h[0] = H; g.w[0] = W;
  for (int k = 1; k <= 6; ++k) { g.h[k] = svs_conv_out(g.h[k - 1]); g.w[k] = svs_conv_out(g.w[k - 1]); }
  for (int k = 0; k <= 6; ++k) g.P[k] = (long)B * g.h[k] * g.w[k];
  return SVS_OK;
}

// One channel half of the level-k concat buffer (which: 0 = decoder output, 1 = encoder/skip output).
// Levels 2..5 interleave the halves inside a pixel (ld = 2*ch, a half is >= 128 B so accesses are whole lines).
// Level 1 has 16-channel halves (64 B): interleaved, every access of a half would touch half a 128-byte line and
// drag the other half through the caches, so level 1 is PLANAR -- two dense (P, 16) planes back to back; the three
// kernels that need all 32 channels of a pixel (deconv6 forward / weight gradient / data gradient) take the plane
// distance as `half` (special.hip: chan_off).
struct View { float* p; long ld; };
static View cat_half(float* const* cat, const Geo& g, int k, int which) {
  if (k == 1) return View{cat[1] + (long)which * g.P[1] * 16, 16};
  return View{cat[k] + (long)which * CH[k], 2L * CH[k]};
}

// bump allocator over the caller's workspace (every block 256-byte aligned)
struct Arena {
  char* base; size_t used;
  float* take(size_t nfloats) {
    float* p = base ? (float*)(base + used) : nullptr;
    used += svs_align_up(nfloats * sizeof(float), 256);
    return p;
  }
};

// ---------------------------------------------------------------------------------------------
// eval
// ---------------------------------------------------------------------------------------------
struct Prepared {      // offsets in floats into the prepared blob
  long wp[12], scale[11], shift[11], bias6, total;
};
static Prepared prepared_layout() {
  Prepared p{};
  long off = 0;
  auto take = [&](long n) { long o = off; off += (n + 63) / 64 * 64; return o; };
  for (int l = 0; l < 12; ++l) p.wp[l] = take(param_numel(4 * l));
  for (int l = 0; l < 11; ++l) { p.scale[l] = take(bn_channels(l)); p.shift[l] = take(bn_channels(l)); }
  p.bias6 = take(1);
  p.total = off;
  return p;
}
void svs_unet_prepared_offsets(long wp[12], long scale[11], long shift[11], long* bias6) {
  const Prepared L = prepared_layout();
  for (int l = 0; l < 12; ++l) wp[l] = L.wp[l];
  for (int l = 0; l < 11; ++l) { scale[l] = L.scale[l]; shift[l] = L.shift[l]; }
  *bias6 = L.bias6;
}
extern "C" size_t svs_unet_prepared_bytes(void) { return (size_t)prepared_layout().total * sizeof(float); }

extern "C" int svs_unet_prepare_eval(const float* params, const float* bn_buffers, void* prepared, hipStream_t stream) {
  SVS_REQUIRE(params && bn_buffers && prepared && svs_aligned16(params) && svs_aligned16(prepared), "svs_unet_prepare_eval: bad pointers");
  const Prepared L = prepared_layout();
  const ParamView v = view_params(params);
  float* blob = (float*)prepared;
  int rc;
  // conv1: C == 1, the gather packing is torch's layout; deconv6: N == 1, kernel reads torch's layout
  SVS_HIP(hipMemcpyAsync(blob + L.wp[0], v.w[0], param_numel(0) * sizeof(float), hipMemcpyDeviceToDevice, stream));
  {
    SvsPackJobs jobs{};
    for (int k = 2; k <= 6; ++k) jobs.j[jobs.n++] = SvsPackJob{v.w[k - 1], blob + L.wp[k - 1], CH[k], CH[k - 1], 0, 0};
    for (int j = 0; j < 5; ++j) jobs.j[jobs.n++] = SvsPackJob{v.w[6 + j], blob + L.wp[6 + j], DEC_N[j], DEC_C[j], 1, 0};
    if ((rc = svs_pack_all_run(jobs, stream))) return rc;
  }
  SVS_HIP(hipMemcpyAsync(blob + L.wp[11], v.w[11], param_numel(44) * sizeof(float), hipMemcpyDeviceToDevice, stream));
  SVS_HIP(hipMemcpyAsync(blob + L.bias6, v.b[11], sizeof(float), hipMemcpyDeviceToDevice, stream));
  for (int l = 0; l < 11; ++l) {
    const float* rm = bn_buffers + svs_unet_buffer_offset(l, 0);
    const float* rv = bn_buffers + svs_unet_buffer_offset(l, 1);
    if ((rc = svs_bn_fold(v.gamma[l], v.beta[l], rm, rv, v.b[l], BN_EPS, blob + L.scale[l], blob + L.shift[l], bn_channels(l), stream))) return rc;
  }
  return SVS_OK;
}

struct EvalWs { float* cat[6]; float* c6; float* scratch; size_t scratch_bytes; size_t total; };
static EvalWs eval_layout(const Geo& g, void* ws) {
  EvalWs e{};
  Arena a{(char*)ws, 0};
  for (int k = 1; k <= 5; ++k) e.cat[k] = a.take((size_t)g.P[k] * 2 * CH[k]);
  e.c6 = a.take((size_t)g.P[6] * 512);
  size_t sb = 0;
  for (int k = 2; k <= 6; ++k) {
    size_t s = svs_conv_gemm_workspace(SVS_MODE_GATHER, g.B, g.h[k - 1], g.w[k - 1], CH[k - 1], g.h[k], g.w[k], CH[k]);
    if (s > sb) sb = s;
  }
  for (int j = 0; j < 5; ++j) {
    size_t s = svs_conv_gemm_workspace(SVS_MODE_PARITY, g.B, g.h[6 - j], g.w[6 - j], DEC_C[j], g.h[5 - j], g.w[5 - j], DEC_N[j]);
    if (s > sb) sb = s;
  }
  e.scratch_bytes = sb;
  e.scratch = a.take(sb / sizeof(float) + 64);
  e.total = a.used;
  return e;
}
extern "C" size_t svs_unet_eval_workspace_bytes(int B, int H, int W) {
  Geo g;
  if (make_geo(B, H, W, g)) return 0;
  return eval_layout(g, nullptr).total;
}

extern "C" int svs_unet_forward_eval(const void* prepared, const float* mix, float* mask, int B, int H, int W, void* ws,
                                     size_t ws_bytes, hipStream_t stream) {
  Geo g;
  int rc = make_geo(B, H, W, g);
  if (rc) return rc;
  SVS_REQUIRE(prepared && mix && mask && svs_aligned16(mix) && svs_aligned16(mask), "svs_unet_forward_eval: bad pointers");
  const EvalWs e = eval_layout(g, ws);
  if (!ws || ws_bytes < e.total || !svs_aligned16(ws)) { svs_set_error("svs_unet_forward_eval: workspace too small (%zu < %zu)", ws_bytes, e.total); return SVS_ERR_WORKSPACE; }
  const Prepared L = prepared_layout();
  const float* blob = (const float*)prepared;
  // encoder (model.py:176-181): BN folded, LeakyReLU(0.2) in the epilogue
  for (int k = 1; k <= 6; ++k) {
    const View xi = (k == 1) ? View{const_cast<float*>(mix), 1} : cat_half(e.cat, g, k - 1, 1);
    const View yo = (k == 6) ? View{e.c6, 512} : cat_half(e.cat, g, k, 1);
    const float* x = xi.p; const long ldx = xi.ld;
    float* y = yo.p; const long ldy = yo.ld;
    rc = svs_enc_block_fwd(x, ldx, B, g.h[k - 1], g.w[k - 1], CH[k - 1], blob + L.wp[k - 1], nullptr,
                           blob + L.scale[k - 1], blob + L.shift[k - 1], LEAKY, y, ldy, CH[k], 0, e.scratch, e.scratch_bytes, stream);
    if (rc) return rc;
  }
  // decoder (model.py:183-196): BN folded, ReLU; Dropout2d is the identity in eval
  for (int j = 0; j < 5; ++j) {
    const int lin = 6 - j, lout = 5 - j;
    const float* x = (j == 0) ? e.c6 : e.cat[lin];
    const View yo = cat_half(e.cat, g, lout, 0);
    rc = svs_dec_block_fwd(x, DEC_C[j], B, g.h[lin], g.w[lin], DEC_C[j], blob + L.wp[6 + j], nullptr, blob + L.scale[6 + j],
                           blob + L.shift[6 + j], 0.f, yo.p, yo.ld, g.h[lout], g.w[lout], DEC_N[j], 0,
                           e.scratch, e.scratch_bytes, stream);
    if (rc) return rc;
  }
  // deconv6 + sigmoid (model.py:198-200)
  return svs_deconv_to1_run(e.cat[1], 16, B, g.h[1], g.w[1], 32, blob + L.wp[11], blob + L.bias6, mask, H, W, 1, stream,
                            "svs_unet_forward_eval", g.P[1] * 16);
}

// ---------------------------------------------------------------------------------------------
// training
// ---------------------------------------------------------------------------------------------
struct TrainWs {
  float* cat[6]; float* c6;
  float* raw_e[7]; float* raw_d[5];
  float* mean[11]; float* invstd[11];
  float* wfwd[12]; float* wbwd[12];       // packed weights (null where torch's layout is read directly)
  float* dcat[6]; float* dc6;
  float* d_raw; float* d_logit; float* mask;
  float* d_raw_l[11];                     // one d_raw per BatchNorm layer: the side stream's weight gradients trail freely
  float* bnws; size_t bnws_bytes;
  float* dbias_part[11];                  // per-layer partial sums of d_raw (bias gradients), reduced in one batched pass
  float* scratch; size_t scratch_bytes;
  float* scratch2; size_t scratch2_bytes;     // split-K slabs of the weight-gradient GEMMs (side stream)
  size_t total;
};
#define SVS_FUSED_STATS_ROWS 512          // rows of BatchNorm partials the split-K epilogue may write into bnws
static TrainWs train_layout(const Geo& g, void* ws) {
  TrainWs t{};
  Arena a{(char*)ws, 0};
  const int B = g.B;
  for (int k = 1; k <= 5; ++k) t.cat[k] = a.take((size_t)g.P[k] * 2 * CH[k]);
  t.c6 = a.take((size_t)g.P[6] * 512);
  for (int k = 1; k <= 6; ++k) t.raw_e[k] = a.take((size_t)g.P[k] * CH[k]);
  for (int j = 0; j < 5; ++j) t.raw_d[j] = a.take((size_t)g.P[5 - j] * DEC_N[j]);
  for (int l = 0; l < 11; ++l) { t.mean[l] = a.take(bn_channels(l)); t.invstd[l] = a.take(bn_channels(l)); }
  for (int k = 2; k <= 6; ++k) { t.wfwd[k - 1] = a.take(param_numel(4 * (k - 1))); t.wbwd[k - 1] = a.take(param_numel(4 * (k - 1))); }
  for (int j = 0; j < 5; ++j) { t.wfwd[6 + j] = a.take(param_numel(24 + 4 * j)); t.wbwd[6 + j] = a.take(param_numel(24 + 4 * j)); }
  for (int k = 1; k <= 5; ++k) t.dcat[k] = a.take((size_t)g.P[k] * 2 * CH[k]);
  t.dc6 = a.take((size_t)g.P[6] * 512);
  size_t dmax = 0;
  for (int k = 1; k <= 6; ++k) if ((size_t)g.P[k] * CH[k] > dmax) dmax = (size_t)g.P[k] * CH[k];
  t.d_raw = a.take(dmax);
  for (int l = 0; l < 11; ++l) {
    const int lvl = (l < 6) ? l + 1 : 5 - (l - 6);
    t.d_raw_l[l] = a.take((size_t)g.P[lvl] * bn_channels(l));
  }
  t.d_logit = a.take((size_t)g.P[0]);
  t.mask = a.take((size_t)g.P[0]);
  size_t bb = 0;
  for (int k = 1; k <= 6; ++k) { size_t s = svs_bn_workspace_bytes(g.P[k], CH[k]); if (s > bb) bb = s; }
  if (bb < (size_t)SVS_FUSED_STATS_ROWS * 2 * 512 * sizeof(float)) bb = (size_t)SVS_FUSED_STATS_ROWS * 2 * 512 * sizeof(float);
  t.bnws_bytes = bb + 4096;
  t.bnws = a.take(t.bnws_bytes / sizeof(float));
  for (int l = 0; l < 11; ++l) {
    const int lvl = (l < 6) ? l + 1 : 5 - (l - 6);
    t.dbias_part[l] = a.take(svs_bn_partial_floats(g.P[lvl], bn_channels(l)));
  }
  size_t sb = 4096;
  auto upd = [&](size_t s) { if (s > sb) sb = s; };
  for (int k = 2; k <= 6; ++k) {
    upd(svs_conv_gemm_workspace(SVS_MODE_GATHER, B, g.h[k - 1], g.w[k - 1], CH[k - 1], g.h[k], g.w[k], CH[k]));
    upd(svs_conv_gemm_workspace(SVS_MODE_PARITY, B, g.h[k], g.w[k], CH[k], g.h[k - 1], g.w[k - 1], CH[k - 1]));   // bwd data
    upd(svs_block_bwd_weight_workspace_bytes(B, g.h[k], g.w[k], CH[k], CH[k - 1]));
  }
  upd(svs_block_bwd_weight_workspace_bytes(B, g.h[1], g.w[1], 16, 1));
  for (int j = 0; j < 6; ++j) {
    const int lin = 6 - j, lout = 5 - j;
    if (j < 5) {
      upd(svs_conv_gemm_workspace(SVS_MODE_PARITY, B, g.h[lin], g.w[lin], DEC_C[j], g.h[lout], g.w[lout], DEC_N[j]));
      upd(svs_conv_gemm_workspace(SVS_MODE_GATHER, B, g.h[lout], g.w[lout], DEC_N[j], g.h[lin], g.w[lin], DEC_C[j]));  // bwd data
    }
    upd(svs_block_bwd_weight_workspace_bytes(B, g.h[lin], g.w[lin], DEC_C[j], DEC_N[j]));
  }
  t.scratch_bytes = sb;
  t.scratch = a.take(sb / sizeof(float) + 64);
  size_t sb2 = svs_block_bwd_weight_workspace_bytes(B, g.h[1], g.w[1], 16, 1);
  for (int k = 2; k <= 6; ++k) { const size_t s2 = svs_block_bwd_weight_workspace_bytes(B, g.h[k], g.w[k], CH[k], CH[k - 1]); if (s2 > sb2) sb2 = s2; }
  for (int j = 0; j < 6; ++j) { const size_t s2 = svs_block_bwd_weight_workspace_bytes(B, g.h[6 - j], g.w[6 - j], DEC_C[j], DEC_N[j]); if (s2 > sb2) sb2 = s2; }
  t.scratch2_bytes = sb2;
  t.scratch2 = a.take(sb2 / sizeof(float) + 64);
  t.total = a.used;
  return t;
}

// ---- side stream ---------------------------------------------------------------------------------
// In the backward pass the weight gradient of a layer (MFMA GEMM + slab reductions) depends only on that layer's
// d_raw, while the chain that the next layer waits for is d_raw -> backward-data GEMM -> next BatchNorm backward
// (bandwidth-bound passes).  The weight-gradient work therefore runs on a second HIP stream: its GEMMs fill the
// machine while the main stream is in the bandwidth-bound BatchNorm passes and launch gaps, and its small reduction
// kernels hide under the main stream's GEMMs.  Fork / join are events; d_raw is double-buffered so that the side
// stream may trail the main one by a layer.  Results do not depend on the interleaving (no atomics anywhere).
struct SideStream { hipStream_t s; hipEvent_t fork[4], done[4], sync; int nfork; std::mutex mu; };   // nfork: forks of the current backward pass; mu: held while a training entry point enqueues
static SideStream* g_side[64] = {};
static std::mutex g_side_mutex;
static SideStream* side_stream(hipStream_t of) {       // the side stream of the device `of` belongs to (legacy / null stream: the current device)
  int dev = 0;
  hipDevice_t sdev;
  if (of && hipStreamGetDevice(of, &sdev) == hipSuccess) dev = (int)sdev;
  else if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  if (dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lock(g_side_mutex);
  if (!g_side[dev]) {
    SideStream* sd = new SideStream();
    // HIGHEST priority (measured: 3.75 ms per step against 3.77 at normal and 3.83 at low priority): the weight-gradient
    // GEMMs then start as soon as their d_raw exists and are out of the way when the main stream reaches its next GEMM
    int least = 0, greatest = 0;
    bool ok = hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess;
    int prio = greatest;
    if (svs_tune_on(SVS_TUNE_SIDE_PRIORITY)) prio = (int)svs_tune(SVS_TUNE_SIDE_PRIORITY);   // sweeps only
    ok = ok && hipStreamCreateWithPriority(&sd->s, hipStreamNonBlocking, prio) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&sd->sync, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < 4 && ok; ++i)
      ok = hipEventCreateWithFlags(&sd->fork[i], hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&sd->done[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) { delete sd; return nullptr; }
    g_side[dev] = sd;
  }
  return g_side[dev];
}

extern "C" size_t svs_unet_train_workspace_bytes(int B, int H, int W) {
  Geo g;
  if (make_geo(B, H, W, g)) return 0;
  return train_layout(g, nullptr).total;
}

extern "C" int64_t svs_unet_ws_offset(const char* name, int B, int H, int W, int training) {
  Geo g;
  if (!name || make_geo(B, H, W, g)) return -1;
  char* const base = (char*)256;   // non-null dummy so the arena hands out addresses
  auto rel = [&](const float* p) { return p ? (int64_t)((const char*)p - base) : (int64_t)-1; };
  int idx = 0;
  if (training) {
    const TrainWs t = train_layout(g, base);
    if (sscanf(name, "cat%d", &idx) == 1 && idx >= 1 && idx <= 5 && name[0] == 'c') return rel(t.cat[idx]);
    if (!strcmp(name, "c6")) return rel(t.c6);
    if (sscanf(name, "raw_e%d", &idx) == 1 && idx >= 1 && idx <= 6) return rel(t.raw_e[idx]);
    if (sscanf(name, "raw_d%d", &idx) == 1 && idx >= 1 && idx <= 5) return rel(t.raw_d[idx - 1]);
    if (sscanf(name, "dcat%d", &idx) == 1 && idx >= 1 && idx <= 5) return rel(t.dcat[idx]);
    if (!strcmp(name, "dc6")) return rel(t.dc6);
    if (!strcmp(name, "d_logit")) return rel(t.d_logit);
    if (!strcmp(name, "mask")) return rel(t.mask);
    if (sscanf(name, "mean%d", &idx) == 1 && idx >= 0 && idx < 11) return rel(t.mean[idx]);
    if (sscanf(name, "invstd%d", &idx) == 1 && idx >= 0 && idx < 11) return rel(t.invstd[idx]);
    return -1;
  }
  const EvalWs e = eval_layout(g, base);
  if (sscanf(name, "cat%d", &idx) == 1 && idx >= 1 && idx <= 5) return rel(e.cat[idx]);
  if (!strcmp(name, "c6")) return rel(e.c6);
  return -1;
}

static int train_forward_impl(const ParamView& v, float* bn_buffers, int64_t* nbt, const float* mix, const float* drop,
                              const Geo& g, const TrainWs& t, float* mask, hipStream_t stream) {
  const int B = g.B;
  int rc;
  const int fused_rows = svs_tune_flag(SVS_TUNE_TRAIN_UNFUSED) ? 0 : (int)(t.bnws_bytes / sizeof(float));     // capacity (floats) for fused BatchNorm partials; A/B switch
  SideStream* sd = svs_tune_flag(SVS_TUNE_TRAIN_ONE_STREAM) ? nullptr : side_stream(stream);          // A/B switch
  std::unique_lock<std::mutex> guard;
  if (sd) guard = std::unique_lock<std::mutex>(sd->mu);
  // weight packings for this step (weights change every optimiser step)
  {
    SvsPackJobs jobs{};
    auto add = [&](const float* w, float* wp, int N, int C, int kind) { jobs.j[jobs.n++] = SvsPackJob{w, wp, N, C, kind, 0}; };
    for (int k = 2; k <= 6; ++k) {
      add(v.w[k - 1], t.wfwd[k - 1], CH[k], CH[k - 1], 0);          // conv forward: gather packing
      add(v.w[k - 1], t.wbwd[k - 1], CH[k - 1], CH[k], 1);          // conv bwd-data: (N,C,..) read as (in=N,out=C): one row per C
    }
    for (int j = 0; j < 5; ++j) {
      add(v.w[6 + j], t.wfwd[6 + j], DEC_N[j], DEC_C[j], 1);        // convT forward: parity packing, one row per output channel
      add(v.w[6 + j], t.wbwd[6 + j], DEC_C[j], DEC_N[j], 0);        // convT bwd-data: (C,N,..) read as (n=C,c=N)
    }
    // conv1 reads torch's layout directly, so the packing runs beside it on the side stream (joined before conv2)
    if (sd) {
      SVS_HIP(hipEventRecord(sd->fork[0], stream));
      SVS_HIP(hipStreamWaitEvent(sd->s, sd->fork[0], 0));
    }
    if ((rc = svs_pack_all_run(jobs, sd ? sd->s : stream))) return rc;
    if (sd) SVS_HIP(hipEventRecord(sd->done[0], sd->s));
  }
  // encoder: conv (+bias) -> raw; batch stats; BN + LeakyReLU -> second half of cat[k]
  for (int k = 1; k <= 6; ++k) {
    if (k == 2 && sd) SVS_HIP(hipStreamWaitEvent(stream, sd->done[0], 0));
    const View xi = (k == 1) ? View{const_cast<float*>(mix), 1} : cat_half(t.cat, g, k - 1, 1);
    const float* x = xi.p; const long ldx = xi.ld;
    const float* wp = (k == 1) ? v.w[0] : t.wfwd[k - 1];
    int stat_rows = 0;                       // > 0: the split-K epilogue already left the BatchNorm partials in bnws
    if (k == 1) rc = svs_enc_block_fwd(x, ldx, B, g.h[0], g.w[0], 1, wp, v.b[0], nullptr, nullptr, 0.f, t.raw_e[1], CH[1], CH[1], 0,
                                       t.scratch, t.scratch_bytes, stream);
    else rc = svs_conv_gemm_run(SVS_MODE_GATHER, x, ldx, B, g.h[k - 1], g.w[k - 1], CH[k - 1], wp, v.b[k - 1], nullptr, nullptr, 0.f,
                                t.raw_e[k], CH[k], g.h[k], g.w[k], CH[k], 0, t.scratch, t.scratch_bytes, stream, "conv forward",
                                t.bnws, fused_rows, &stat_rows);
    if (rc) return rc;
    const int l = k - 1;
    if (!stat_rows) {
      if ((rc = svs_bn_stats(t.raw_e[k], CH[k], g.P[k], CH[k], t.bnws, t.bnws_bytes, stream))) return rc;
      stat_rows = svs_bn_partial_rows(g.P[k], CH[k]);
    }
    const View yo = (k == 6) ? View{t.c6, 512} : cat_half(t.cat, g, k, 1);
    float* y = yo.p; const long ldy = yo.ld;
    if ((rc = svs_bn_fin_act_apply_run(t.bnws, stat_rows, t.raw_e[k], CH[k], g.P[k], CH[k], (long)g.h[k] * g.w[k], v.gamma[l], v.beta[l],
                                       BN_EPS, BN_MOMENTUM, bn_buffers ? bn_buffers + svs_unet_buffer_offset(l, 0) : nullptr,
                                       bn_buffers ? bn_buffers + svs_unet_buffer_offset(l, 1) : nullptr,
                                       nbt ? (long long*)(nbt + l) : nullptr, t.mean[l], t.invstd[l], LEAKY, nullptr, y, ldy, stream))) return rc;
  }
  // decoder: convT (+bias) -> raw; batch stats; BN + ReLU + Dropout2d -> first half of cat[lout]
  const float* dp = drop;
  for (int j = 0; j < 5; ++j) {
    const int lin = 6 - j, lout = 5 - j, l = 6 + j;
    const float* x = (j == 0) ? t.c6 : t.cat[lin];
    int stat_rows = 0;
    rc = svs_conv_gemm_run(SVS_MODE_PARITY, x, DEC_C[j], B, g.h[lin], g.w[lin], DEC_C[j], t.wfwd[l], v.b[l], nullptr, nullptr, 0.f,
                           t.raw_d[j], DEC_N[j], g.h[lout], g.w[lout], DEC_N[j], 0, t.scratch, t.scratch_bytes, stream,
                           "deconv forward", t.bnws, fused_rows, &stat_rows);
    if (rc) return rc;
    if (!stat_rows) {
      if ((rc = svs_bn_stats(t.raw_d[j], DEC_N[j], g.P[lout], DEC_N[j], t.bnws, t.bnws_bytes, stream))) return rc;
      stat_rows = svs_bn_partial_rows(g.P[lout], DEC_N[j]);
    }
    const View yo = cat_half(t.cat, g, lout, 0);
    if ((rc = svs_bn_fin_act_apply_run(t.bnws, stat_rows, t.raw_d[j], DEC_N[j], g.P[lout], DEC_N[j], (long)g.h[lout] * g.w[lout], v.gamma[l],
                                       v.beta[l], BN_EPS, BN_MOMENTUM, bn_buffers ? bn_buffers + svs_unet_buffer_offset(l, 0) : nullptr,
                                       bn_buffers ? bn_buffers + svs_unet_buffer_offset(l, 1) : nullptr,
                                       nbt ? (long long*)(nbt + l) : nullptr, t.mean[l], t.invstd[l], 0.f, dp, yo.p, yo.ld, stream))) return rc;
    if (dp) dp += (long)B * DEC_N[j];
  }
  return svs_deconv_to1_run(t.cat[1], 16, B, g.h[1], g.w[1], 32, v.w[11], v.b[11], mask, g.h[0], g.w[0], 1, stream,
                            "svs_unet_train_forward", g.P[1] * 16);
}

// parts: bit 0 = decoder half (deconv6..deconv1: gradients of parameter tensors 24..45, produced FIRST),
//        bit 1 = encoder half (conv6..conv1: tensors 0..23).  A data-parallel caller runs them as two calls and
//        all-reduces the decoder half of the flat gradient buffer while the encoder half is still being computed.
static int train_backward_impl(const ParamView& v, float* grads, const float* mix, const float* drop, const Geo& g,
                               const TrainWs& t, hipStream_t stream, int parts = 15) {
  const int B = g.B;
  int rc;
  auto G = [&](int idx) { return grads + svs_unet_param_offset(idx); };
  SvsSumJobs sums{};                         // bias-gradient reductions, run as one batched launch per half
  const bool unfused = svs_tune_flag(SVS_TUNE_TRAIN_UNFUSED);     // A/B switch: one launch per reduction, as before
  SideStream* sd = svs_tune_flag(SVS_TUNE_TRAIN_ONE_STREAM) ? nullptr : side_stream(stream);   // A/B switch: everything on `stream`
  std::unique_lock<std::mutex> guard;
  if (sd) guard = std::unique_lock<std::mutex>(sd->mu);
  const hipStream_t wstream = sd ? sd->s : stream;                  // where the weight gradients run
  float* const wscratch = sd ? t.scratch2 : t.scratch;
  const size_t wscratch_bytes = sd ? t.scratch2_bytes : t.scratch_bytes;
  // forks so far in this backward pass (kept in the SideStream across the calls of a split pass, which must come in
  // order on one host thread); fork n uses event slot n & 3.  Every layer has a d_raw buffer of its own, so the main
  // stream never waits for a weight gradient before the end of the pass (an event pair per layer measured dearer than
  // the 0.2 GB of workspace).
  int nfork_local = 0;
  int& nfork = sd ? sd->nfork : nfork_local;
  if (parts & 1) nfork = 0;
  auto layer_draw = [&](int l) -> float* { return sd ? t.d_raw_l[l] : t.d_raw; };
  auto fork = [&]() -> int {                 // the side stream may start once everything queued on `stream` so far is done
    if (!sd) return SVS_OK;
    SVS_HIP(hipEventRecord(sd->fork[nfork & 3], stream));
    SVS_HIP(hipStreamWaitEvent(sd->s, sd->fork[nfork & 3], 0));
    return SVS_OK;
  };
  auto forked = [&]() -> int { ++nfork; return SVS_OK; };
  auto join = [&]() -> int {                 // `stream` waits for all side work enqueued so far
    if (sd && nfork > 0) {
      SVS_HIP(hipEventRecord(sd->done[0], sd->s));
      SVS_HIP(hipStreamWaitEvent(stream, sd->done[0], 0));
    }
    return SVS_OK;
  };
  if (parts & 1) {
  // deconv6 (model.py:109,198): dw, db, dx -> dcat[1]
  const long half1 = g.P[1] * 16;     // level 1 is planar (cat_half)
  if ((rc = fork())) return rc;
  if ((rc = svs_wgrad_c1_run(t.cat[1], 16, B, g.h[1], g.w[1], 32, t.d_logit, g.h[0], g.w[0], G(44), wscratch, wscratch_bytes, wstream,
                             "deconv6 bwd_weight", half1))) return rc;
  if ((rc = svs_sum_run(t.d_logit, g.P[0], G(45), wscratch, wscratch_bytes, wstream))) return rc;      // deconv6 bias gradient
  if ((rc = forked())) return rc;
  if ((rc = svs_conv_c1_run(t.d_logit, B, g.h[0], g.w[0], v.w[11], nullptr, nullptr, nullptr, 0.f, t.dcat[1], 16, 32, 0, stream,
                            "deconv6 bwd_data", half1))) return rc;
  // decoders 5..1
  long drop_off[5];
  { long o = 0; for (int j = 0; j < 5; ++j) { drop_off[j] = o; o += (long)B * DEC_N[j]; } }
  for (int j = 4; j >= 0; --j) {
    const int lin = 6 - j, lout = 5 - j, l = 6 + j, N = DEC_N[j], C = DEC_C[j];
    const float* x = (j == 0) ? t.c6 : t.cat[lin];
    const View dyv = cat_half(t.dcat, g, lout, 0);
    float* const d_raw = layer_draw(l);
    rc = svs_bn_bwd_run(dyv.p, dyv.ld, t.raw_d[j], N, g.P[lout], N, (long)g.h[lout] * g.w[lout], v.gamma[l], v.beta[l],
                        t.mean[l], t.invstd[l], 0.f, drop ? drop + drop_off[j] : nullptr, d_raw, G(24 + 4 * j + 2), G(24 + 4 * j + 3),
                        G(24 + 4 * j + 1), t.bnws, t.bnws_bytes, stream, unfused ? nullptr : t.dbias_part[l], &sums);   // + bias gradient (sum of d_raw)
    if (rc) return rc;
    if ((rc = fork())) return rc;
    if ((rc = svs_dec_block_bwd_weight(x, C, B, g.h[lin], g.w[lin], C, d_raw, N, g.h[lout], g.w[lout], N, G(24 + 4 * j), nullptr,
                                       wscratch, wscratch_bytes, wstream))) return rc;
    if ((rc = forked())) return rc;
    float* dx = (j == 0) ? t.dc6 : t.dcat[lin];
    if ((rc = svs_dec_block_bwd_data(d_raw, N, B, g.h[lout], g.w[lout], N, t.wbwd[l], dx, C, g.h[lin], g.w[lin], C, 0,
                                     t.scratch, t.scratch_bytes, stream))) return rc;
  }
  if ((rc = svs_channel_sum_finalize_multi_run(sums, stream))) return rc;    // the five decoder bias gradients
  sums.njobs = 0;
  }
  if (!(parts & 14)) return SVS_OK;
  // encoders 6..1 (bit 2: block 6, whose 13 MB of gradients are most of the encoder's; bit 4: blocks 5 and 4 (4.1 MB);
  // bit 8: blocks 3..1 (0.26 MB: the only piece a data-parallel step exchanges after the backward has ended))
  for (int k = 6; k >= 1; --k) {
    if (!(parts & (k == 6 ? 2 : k >= 4 ? 4 : 8))) continue;
    const int l = k - 1, N = CH[k], C = CH[k - 1];
    const View dyv = (k == 6) ? View{t.dc6, 512} : cat_half(t.dcat, g, k, 1);
    const float* dy = dyv.p; const long lddy = dyv.ld;
    float* const d_raw = layer_draw(l);
    rc = svs_bn_bwd_run(dy, lddy, t.raw_e[k], N, g.P[k], N, (long)g.h[k] * g.w[k], v.gamma[l], v.beta[l], t.mean[l], t.invstd[l],
                        LEAKY, nullptr, d_raw, G(4 * l + 2), G(4 * l + 3), G(4 * l + 1), t.bnws, t.bnws_bytes, stream,
                        unfused ? nullptr : t.dbias_part[l], &sums);
    if (rc) return rc;
    const View xi = (k == 1) ? View{const_cast<float*>(mix), 1} : cat_half(t.cat, g, k - 1, 1);
    const float* x = xi.p; const long ldx = xi.ld;
    if (k == 1) {
      // conv1 is the end of the pass: nothing is left for the main stream to run beside this weight gradient, so it runs there
      // itself (a fork + join for it left the main stream idle for 44 us before Adam: the fork's start latency and the join)
      if ((rc = svs_enc_block_bwd_weight(d_raw, N, B, g.h[k], g.w[k], N, x, ldx, g.h[k - 1], g.w[k - 1], C, G(4 * l), nullptr,
                                         t.scratch, t.scratch_bytes, stream))) return rc;
      continue;
    }
    if ((rc = fork())) return rc;
    if ((rc = svs_enc_block_bwd_weight(d_raw, N, B, g.h[k], g.w[k], N, x, ldx, g.h[k - 1], g.w[k - 1], C, G(4 * l), nullptr,
                                       wscratch, wscratch_bytes, wstream))) return rc;
    if ((rc = forked())) return rc;
    if (k >= 2) {
      // gradient of the skip half of cat[k-1]: add to what decoder (7-k)'s bwd_data left there
      const View dxs = cat_half(t.dcat, g, k - 1, 1);
      if ((rc = svs_enc_block_bwd_data(d_raw, N, B, g.h[k], g.w[k], N, t.wbwd[l], dxs.p, dxs.ld, g.h[k - 1], g.w[k - 1], C, 1,
                                       t.scratch, t.scratch_bytes, stream))) return rc;
    }
  }
  if ((rc = svs_channel_sum_finalize_multi_run(sums, stream))) return rc;    // the encoder bias gradients of this call
  // `stream` is joined with the side stream only by the call that ends the pass (block 1 included); after an earlier
  // part of a split pass the caller uses svs_unet_train_bwd_sync() on the stream that consumes that part's gradients
  return (parts & 8) ? join() : SVS_OK;
}

extern "C" int svs_unet_train_bwd_sync(hipStream_t consumer) {
  SideStream* sd = svs_tune_flag(SVS_TUNE_TRAIN_ONE_STREAM) ? nullptr : side_stream(consumer);
  if (!sd) return SVS_OK;
  std::lock_guard<std::mutex> guard(sd->mu);
  SVS_HIP(hipEventRecord(sd->sync, sd->s));
  SVS_HIP(hipStreamWaitEvent(consumer, sd->sync, 0));
  return SVS_OK;
}

static int check_train_ws(const char* who, const Geo& g, void* ws, size_t ws_bytes, TrainWs& t) {
  t = train_layout(g, ws);
  if (!ws || ws_bytes < t.total || !svs_aligned16(ws)) {
    svs_set_error("%s: workspace too small (%zu < %zu)", who, ws_bytes, t.total);
    return SVS_ERR_WORKSPACE;
  }
  return SVS_OK;
}

extern "C" int svs_unet_train_forward(const float* params, float* bn_buffers, int64_t* num_batches_tracked, const float* mix,
                                      const float* drop, int B, int H, int W, float* mask, void* ws, size_t ws_bytes,
                                      hipStream_t stream) {
  Geo g; TrainWs t;
  int rc = make_geo(B, H, W, g);
  if (rc) return rc;
  SVS_REQUIRE(params && mix && mask && svs_aligned16(params) && svs_aligned16(mix) && svs_aligned16(mask), "svs_unet_train_forward: bad pointers");
  if ((rc = check_train_ws("svs_unet_train_forward", g, ws, ws_bytes, t))) return rc;
  return train_forward_impl(view_params(params), bn_buffers, num_batches_tracked, mix, drop, g, t, mask, stream);
}

extern "C" int svs_unet_train_backward(const float* params, float* grads, const float* mix, const float* mask,
                                       const float* d_mask, const float* drop, int B, int H, int W, void* ws,
                                       size_t ws_bytes, hipStream_t stream) {
  Geo g; TrainWs t;
  int rc = make_geo(B, H, W, g);
  if (rc) return rc;
  SVS_REQUIRE(params && grads && mix && mask && d_mask && svs_aligned16(grads), "svs_unet_train_backward: bad pointers");
  if ((rc = check_train_ws("svs_unet_train_backward", g, ws, ws_bytes, t))) return rc;
  if ((rc = svs_sigmoid_bwd_run(mask, d_mask, g.P[0], t.d_logit, stream))) return rc;
  return train_backward_impl(view_params(params), grads, mix, drop, g, t, stream);
}

extern "C" int svs_unet_train_fwd_bwd(const float* params, float* grads, float* bn_buffers, int64_t* num_batches_tracked,
                                      const float* mix, const float* voc, const float* drop, int B, int H, int W,
                                      float loss_scale, float* mask, float* loss, void* ws, size_t ws_bytes,
                                      hipStream_t stream) {
  Geo g; TrainWs t;
  int rc = make_geo(B, H, W, g);
  if (rc) return rc;
  SVS_REQUIRE(params && grads && mix && voc && loss && svs_aligned16(params) && svs_aligned16(grads) && svs_aligned16(mix),
              "svs_unet_train_fwd_bwd: bad pointers");
  if ((rc = check_train_ws("svs_unet_train_fwd_bwd", g, ws, ws_bytes, t))) return rc;
  float* m = mask ? mask : t.mask;
  const ParamView v = view_params(params);
  if ((rc = train_forward_impl(v, bn_buffers, num_batches_tracked, mix, drop, g, t, m, stream))) return rc;
  if ((rc = svs_l1_mask_loss_fwd_bwd(m, mix, voc, g.P[0], loss_scale, t.d_logit, loss, t.bnws, t.bnws_bytes, stream))) return rc;
  return train_backward_impl(v, grads, mix, drop, g, t, stream);
}

// Split form of svs_unet_train_fwd_bwd for gradient-exchange overlap: forward + loss, then the backward in
// two parts (see train_backward_impl).
extern "C" int svs_unet_train_fwd_loss(const float* params, float* bn_buffers, int64_t* num_batches_tracked, const float* mix,
                                       const float* voc, const float* drop, int B, int H, int W, float loss_scale, float* mask,
                                       float* loss, void* ws, size_t ws_bytes, hipStream_t stream) {
  Geo g; TrainWs t;
  int rc = make_geo(B, H, W, g);
  if (rc) return rc;
  SVS_REQUIRE(params && mix && voc && loss && svs_aligned16(params) && svs_aligned16(mix), "svs_unet_train_fwd_loss: bad pointers");
  if ((rc = check_train_ws("svs_unet_train_fwd_loss", g, ws, ws_bytes, t))) return rc;
  float* m = mask ? mask : t.mask;
  if ((rc = train_forward_impl(view_params(params), bn_buffers, num_batches_tracked, mix, drop, g, t, m, stream))) return rc;
  return svs_l1_mask_loss_fwd_bwd(m, mix, voc, g.P[0], loss_scale, t.d_logit, loss, t.bnws, t.bnws_bytes, stream);
}

// The reference's full objective (train.py:274-296): alpha_L1 * (L1 vocal + L1 accompaniment) + alpha_MR * MR-STFT(
// specific_istft(mask * mix, mix_phase), specific_istft(voc, voc_phase)).  Forward + both losses + d(total)/d(logit); the
// backward follows with svs_unet_train_bwd_part (part 4 = the whole pass, or the split forms).  Needs H = n_fft / 2 = 512.
//   losses[0] = L1 part (unscaled), losses[1] = MR part (unscaled); total = alpha_l1 * losses[0] + alpha_mr * losses[1]
//   mr_ws: svs_unet_train_mr_workspace_bytes(B, W, hop) bytes (waveforms, their gradient, the loss's frame buffers)
struct MrTrainWs { float* wav_pred; float* wav_tgt; float* d_wav; void* mr; size_t mr_bytes; size_t total; };
static MrTrainWs mr_train_layout(int B, int W, int hop, void* ws) {
  MrTrainWs m{};
  Arena a{(char*)ws, 0};
  const size_t L = (size_t)hop * (W - 1);
  m.wav_pred = a.take((size_t)B * L);
  m.wav_tgt = a.take((size_t)B * L);
  m.d_wav = a.take((size_t)B * L);
  m.mr_bytes = svs_mrstft_workspace_bytes(B, (int64_t)L);
  m.mr = a.take(m.mr_bytes / sizeof(float) + 64);
  m.total = a.used;
  return m;
}
extern "C" size_t svs_unet_train_mr_workspace_bytes(int B, int W, int hop) {
  if (B <= 0 || W < 2 || hop <= 0) return 0;
  return mr_train_layout(B, W, hop, nullptr).total;
}
extern "C" int svs_unet_train_fwd_loss_mr(const float* params, float* bn_buffers, int64_t* num_batches_tracked, const float* mix,
                                          const float* voc, const float* mix_phase, const float* voc_phase, const float* drop,
                                          int B, int H, int W, int hop, float alpha_l1, float alpha_mr, float* mask, float* losses,
                                          void* ws, size_t ws_bytes, void* mr_ws, size_t mr_ws_bytes, hipStream_t stream) {
  Geo g; TrainWs t;
  int rc = make_geo(B, H, W, g);
  if (rc) return rc;
  SVS_REQUIRE(params && mix && voc && mix_phase && voc_phase && losses && svs_aligned16(params) && svs_aligned16(mix),
              "svs_unet_train_fwd_loss_mr: bad pointers");
  SVS_REQUIRE(H == 512 && W >= 2 && hop >= 512 && hop <= 1024, "svs_unet_train_fwd_loss_mr: needs H = 512 (n_fft 1024) and 512 <= hop <= 1024");
  if ((rc = check_train_ws("svs_unet_train_fwd_loss_mr", g, ws, ws_bytes, t))) return rc;
  const MrTrainWs m = mr_train_layout(B, W, hop, mr_ws);
  if (!mr_ws || mr_ws_bytes < m.total || !svs_aligned16(mr_ws)) { svs_set_error("svs_unet_train_fwd_loss_mr: MR workspace too small (%zu < %zu)", mr_ws_bytes, m.total); return SVS_ERR_WORKSPACE; }
  float* mk = mask ? mask : t.mask;
  if ((rc = train_forward_impl(view_params(params), bn_buffers, num_batches_tracked, mix, drop, g, t, mk, stream))) return rc;
  // d_logit = alpha_l1 * d(L1)/d(logit)                                                   (train.py:281-283,296)
  if ((rc = svs_l1_mask_loss_fwd_bwd(mk, mix, voc, g.P[0], alpha_l1, t.d_logit, losses, t.bnws, t.bnws_bytes, stream))) return rc;
  // waveforms: predicted magnitude (mask * mix, fused into the inverse's load) with the MIXTURE phase, target with its own
  const int64_t cs = (int64_t)H * W;
  const long L = (long)hop * (W - 1);
  if ((rc = svs_istft_tiles(mix, cs, W, H, 1, mk, 0, mix_phase, 3, B, 2 * H, hop, W, m.wav_pred, nullptr, stream))) return rc;   // train.py:288
  if ((rc = svs_istft_tiles(voc, cs, W, H, 1, nullptr, 0, voc_phase, 3, B, 2 * H, hop, W, m.wav_tgt, nullptr, stream))) return rc; // train.py:291
  if ((rc = svs_mrstft_loss_fwd_bwd(m.wav_pred, m.wav_tgt, B, L, alpha_mr, losses + 1, m.d_wav, m.mr, m.mr_bytes, stream))) return rc;  // train.py:293
  // d_logit += d(alpha_mr * MR)/d(wav) through the inverse STFT and |S| = mask * mix
  return svs_istft_bwd_mask(m.d_wav, mix_phase, mix, mk, t.d_logit, 1.0f, B, 2 * H, hop, W, stream);
}

extern "C" int svs_unet_train_bwd_part(const float* params, float* grads, const float* mix, const float* drop, int B, int H, int W,
                                       int part, void* ws, size_t ws_bytes, hipStream_t stream) {
  Geo g; TrainWs t;
  int rc = make_geo(B, H, W, g);
  if (rc) return rc;
  SVS_REQUIRE(params && grads && mix && part >= 0 && part <= 6, "svs_unet_train_bwd_part: bad arguments");
  if ((rc = check_train_ws("svs_unet_train_bwd_part", g, ws, ws_bytes, t))) return rc;
  // decoder | whole encoder | conv6 block | conv5..conv1 blocks | everything | conv5 + conv4 blocks | conv3..conv1 blocks
  static const int bits[7] = {1, 2 | 4 | 8, 2, 4 | 8, 15, 4, 8};
  return train_backward_impl(view_params(params), grads, mix, drop, g, t, stream, bits[part]);
}
