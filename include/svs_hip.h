/* libsvs_hip.so -- C ABI of the MI355X (gfx950) singing-voice-separation hot path.
 *
 * The reference (zouyuoz/SVS-UNet-PyTorch, cited below as <file>:<line>) has no FFI: its hot path
 * is a chain of torch / librosa library calls made from Python.  Each entry point here replaces one
 * of those calls (or a run of them) and is what a maintainer would bind with ctypes -- see
 * INTEGRATION.md for the stub.  Conventions:
 *   - every pointer is a DEVICE pointer owned by the caller (torch); the library never allocates
 *     caller-visible memory and never synchronises the stream.  Library-owned state, all of it: (1) per device, ONE
 *     high-priority side stream and a handful of events, created on first use by the svs_unet_train_* entry points
 *     (the weight gradients of a backward pass run there, see svs_unet_train_bwd_sync); (2) the table of tuning
 *     switches behind svs_tuning_set (filled from the environment once).  Everything else is stateless;
 *   - scratch space is passed in (void* ws, size_t ws_bytes); the matching *_workspace_bytes query
 *     says how much is needed; 16-byte alignment is required of every tensor pointer;
 *   - activations are fp32 NHWC "views": pointer to channel 0 of pixel 0 plus `ld`, the distance in
 *     floats between consecutive pixels (so a channel slice of a wider buffer -- the decoder's
 *     skip-concat halves, model.py:186-198 -- is addressed without a copy).  With one channel NHWC
 *     and the reference's NCHW coincide, so the network input/output are the reference's tensors;
 *   - return value: 0 = OK, <0 = invalid argument / workspace too small, >0 = hipError_t;
 *     svs_last_error_string() returns a thread-local description of the last failure;
 *   - callable from any host thread and on any stream (torch's autograd engine calls from its own thread): the
 *     per-block and whole-network eval entry points are fully re-entrant; the svs_unet_train_* entry points of one
 *     DEVICE are serialised by a library mutex while they enqueue (they share that device's side stream), and a split
 *     pass (svs_unet_train_fwd_loss, then svs_unet_train_bwd_part 0, 2, 5, 6 / 0, 2, 3 / 0, 1 / 4) must be issued in order by one
 *     caller at a time;
 *   - the device is the one the `stream` argument belongs to; the caller makes it current (hipSetDevice /
 *     torch.cuda.device) around the call.
 */
#ifndef SVS_HIP_H
#define SVS_HIP_H

#include <stddef.h>
#include <stdint.h>

/* identical to the typedef in <hip/hip_runtime_api.h>; repeating it keeps this header free of HIP includes */
typedef struct ihipStream_t* hipStream_t;

#ifdef __cplusplus
extern "C" {
#endif

#define SVS_OK 0
#define SVS_ERR_INVALID (-1)
#define SVS_ERR_WORKSPACE (-2)
#define SVS_ABI_VERSION 1

int svs_version(void);
const char* svs_last_error_string(void);
/* Planner overrides for sweeps, A/B runs and tests -- never needed in production.  `name` is one of CONV_CFG,
 * CONV_KSPLIT, CONV_WINDOW, CONV_SKIP, CONV_KORDER, CONV_DIRECT, SKIP_REDUCE, WGRAD_CFG, WGRAD_KSPLIT, WGRAD_SKIP,
 * WGRAD_WINDOW, WGRAD_C1_VALU, SIDE_PRIORITY, TRAIN_UNFUSED, TRAIN_ONE_STREAM, CONV_PLAN (0: batch-64 tile table for the
 * inference calls too), MFMA_SPLIT (1: the fp32 GEMM kernels form their products on the bf16 MFMA from exact three-limb
 * splits of the fp32 operands -- fp32-accurate, see csrc/mfma_split.h; default off), CONV_BALANCE (0: one K-split count per
 * layer instead of per-position counts on the tap-skipping layers), CONV_C1_TILED (0: the thread-per-pixel form of the
 * single-channel convolution), BF16_KB (K-tiles per barrier of the bf16 GEMM: 1, 2 or 4), BF16_CFG / BF16_KSPLIT (its tile / K-split),
 * CONV_PF / WGRAD_PF (K-tiles the fp32 conv / weight-gradient GEMMs request ahead: 1 or 2; CONV_PF 3: on every tile shape),
 * CONV_GWINDOW (0: conv2 forward on the GEMM kernel instead of the LDS-window kernel; 2: the window kernel whenever the layer is
 * eligible; >= 16: that many blocks), BF16_CONV3_WINDOW / BF16_DECONV3_WINDOW (0: conv3 / deconv3 of the bf16 network in the GEMM form),
 * BN_INLINE (most partial rows a BatchNorm apply kernel folds itself instead of waiting for a finalise launch; 0: never;
 * default 128), BN_BLOCKS (blocks of the BatchNorm reduce / apply kernels, <= 1024; default 512),
 * or "*" for all; value -1 = planner
 * default ("*", -1: every switch back to what the environment gave at load).  The boolean switches (SKIP_REDUCE,
 * WGRAD_C1_VALU, TRAIN_UNFUSED, TRAIN_ONE_STREAM, MFMA_SPLIT) are ON for values > 0 only -- 0 and -1 both mean off; the
 * others are valued (0 is a value).  The table is initialised from the environment (SVS_<NAME>) at first use; no compute
 * path reads the environment; entries are atomics (a setter may run beside compute threads). */
int svs_tuning_set(const char* name, long value);

/* ---------------------------------------------------------------------------------------------
 * Synthetic data (bench / tests): the counter-based generator of svs_unet_pytorch_amd/synth.py.
 * out[i] = uniform(seed, offset + i) * scale + shift.                       (SURVEY.md 8d inputs) */
int svs_fill_uniform(float* out, int64_t n, uint32_t seed, uint64_t offset, float scale, float shift,
                     hipStream_t stream);
/* mix = u(seed 0), voc = mix * u(seed 1); counter = flat offset + 2^32 * (first_tile + b).
 * Stands in for SpectrogramDataset.__getitem__ (train.py:86-143) in every synthetic config. */
int svs_fill_tiles(float* mix, float* voc, int B, int H, int W, int64_t first_tile, hipStream_t stream);
/* Dropout2d(0.5) keep-masks (model.py:83,89,95,101,107): out[b*C+c] in {0, 2}. */
/* Training tiles from spectrograms kept in HBM -- SpectrogramDataset.__getitem__, train.py:86-143 (L1 path: magnitudes
 * only).  mix_songs / voc_songs: flat buffers, song s = rows 1..F of its (F+1, T_s) file (DC row dropped, train.py:109-112)
 * at float offset[s], (F, T_s) row-major.  Sample b = song[b], columns [start[b], start[b]+seg) of both tracks (shared
 * start, train.py:121), right zero-padded where the song ends (train.py:129-135).  mix / voc: (B, 1, F, seg). */
int svs_crop_tiles(const float* mix_songs, const float* voc_songs, const int64_t* offset, const int32_t* frames,
                   const int32_t* song, const int32_t* start, int B, int F, int seg, float* mix, float* voc,
                   hipStream_t stream);
/* The phase half of the same items (train.py:103-112): angle = np.angle(unit phasor) = atan2(im, re) as float32; the
 * resident angle buffers are then cropped with svs_crop_tiles exactly like the magnitudes (same shared start). */
int svs_phase_angle(const float* phasor /* n complex64 */, float* angle, int64_t n, hipStream_t stream);
int svs_dropout_mask(float* out, int B, int C, int layer, uint32_t seed, int step, int rank, hipStream_t stream);
/* The five decoder masks of one step in one launch: out = [B*256 | B*128 | B*64 | B*32 | B*16] floats, each block
 * bit-identical to svs_dropout_mask(layer = 0..4) -- the layout svs_unet_train_* take as `drop`. */
int svs_dropout_masks_all(float* out, int B, uint32_t seed, int step, int rank, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Weight layouts.  Checkpoints keep torch's layouts (Conv2d (N,C,5,5), model.py:48; ConvTranspose2d
 * (C,N,5,5), model.py:79); the MFMA kernels read K-contiguous packings made by these two.
 *   gather packing  wp[n][kh][kw][c]                       = w[n][c][kh][kw]
 *   parity packing  wp[p][n][th][tw][c], p = 2*ph+pw       = w[c][n][ph+2*th][pw+2*tw]
 *                   (the four output-parity sub-convolutions of a stride-2 transposed conv:
 *                    9/6/6/4 taps; block p starts at {0,9,15,21}*N*C floats) */
int svs_pack_weight_gather(const float* w, float* wp, int N, int C, hipStream_t stream);
int svs_pack_weight_parity(const float* w, float* wp, int C, int N, hipStream_t stream);

/* Eval-mode BatchNorm folded into the producing conv's epilogue (model.py:49,81 in .eval()):
 * scale = gamma / sqrt(running_var + eps), shift = beta + (conv_bias - running_mean) * scale. */
int svs_bn_fold(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                const float* conv_bias, float eps, float* scale, float* shift, int C, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Encoder block: Conv2d(k5,s2,p2) (+ folded BN + LeakyReLU) -- replaces self.convK(x), model.py:176-181.
 *   y[b,oh,ow,n] = epi(bias[n] + sum_{kh,kw,c} x[b,2oh-2+kh,2ow-2+kw,c] * wp[n][kh][kw][c])
 *   epi(v) = v                                  if scale == NULL   (training: raw output for BN stats)
 *          = leaky(v*scale[n]+shift[n], slope)  otherwise          (eval: BN folded)
 * Output is (B, (H+1)/2, (W+1)/2, N).  C == 1 (conv1) takes w as wp[25][N] (tap-major).
 * accumulate != 0 adds into y instead of overwriting it. */
size_t svs_enc_block_workspace_bytes(int B, int H, int W, int C, int N);
int svs_enc_block_fwd(const float* x, int64_t ldx, int B, int H, int W, int C,
                      const float* wp, const float* bias, const float* scale, const float* shift, float slope,
                      float* y, int64_t ldy, int N, int accumulate,
                      void* ws, size_t ws_bytes, hipStream_t stream);

/* Decoder block: ConvTranspose2d(k5,s2,p2, output_size=(Ho,Wo)) (+ folded BN + ReLU) -- replaces
 * self.deconvK(cat, output_size=...), model.py:183-196.  x is the (virtually concatenated) input with
 * C channels; wp is the parity packing.  Ho in {2H-1, 2H}, Wo in {2W-1, 2W}. */
size_t svs_dec_block_workspace_bytes(int B, int H, int W, int C, int Ho, int Wo, int N);
int svs_dec_block_fwd(const float* x, int64_t ldx, int B, int H, int W, int C,
                      const float* wp, const float* bias, const float* scale, const float* shift, float slope,
                      float* y, int64_t ldy, int Ho, int Wo, int N, int accumulate,
                      void* ws, size_t ws_bytes, hipStream_t stream);

/* Output block: deconv6 (C -> 1 channel) + sigmoid -- model.py:198-200.  w is torch's (C,1,5,5)
 * layout as-is; bias is a device scalar.  y is (B,1,Ho,Wo). */
int svs_out_block_fwd(const float* x, int64_t ldx, int B, int H, int W, int C,
                      const float* w, const float* bias, float* y, int Ho, int Wo, int apply_sigmoid,
                      hipStream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Backward (autograd of the above; the reference gets these from torch at train.py:299).
 * bwd_data of a conv is a transposed conv with the same weights and vice versa, so:
 *   svs_enc_block_bwd_data  dx (B,H,W,C) = ConvT(dy (B,Ho,Wo,N));  wpar = parity packing of the Conv2d weight
 *   svs_dec_block_bwd_data  dx (B,H,W,C) = Conv (dy (B,Ho,Wo,N));  wgat = gather packing of the ConvT weight
 * N == 1 in svs_dec_block_bwd_data (deconv6) takes w as wp[25][C]. */
int svs_enc_block_bwd_data(const float* dy, int64_t lddy, int B, int Ho, int Wo, int N, const float* wpar,
                           float* dx, int64_t lddx, int H, int W, int C, int accumulate,
                           void* ws, size_t ws_bytes, hipStream_t stream);
int svs_dec_block_bwd_data(const float* dy, int64_t lddy, int B, int Ho, int Wo, int N, const float* wgat,
                           float* dx, int64_t lddx, int H, int W, int C, int accumulate,
                           void* ws, size_t ws_bytes, hipStream_t stream);
/* Weight gradients, written in torch's layout:
 *   enc: dw[n][c][kh][kw] = sum_{b,oh,ow} dy[b,oh,ow,n] * x[b,2oh-2+kh,2ow-2+kw,c]      (N,C,5,5)
 *   dec: dw[c][n][kh][kw] = sum_{b,ih,iw} x[b,ih,iw,c] * dy[b,2ih-2+kh,2iw-2+kw,n]      (C,N,5,5)
 * and bias gradients db[n] = sum dy[...,n] when db != NULL. */
size_t svs_block_bwd_weight_workspace_bytes(int B, int Hs, int Ws, int Cs, int Cl);
int svs_enc_block_bwd_weight(const float* dy, int64_t lddy, int B, int Ho, int Wo, int N,
                             const float* x, int64_t ldx, int H, int W, int C,
                             float* dw, float* db, void* ws, size_t ws_bytes, hipStream_t stream);
int svs_dec_block_bwd_weight(const float* x, int64_t ldx, int B, int H, int W, int C,
                             const float* dy, int64_t lddy, int Ho, int Wo, int N,
                             float* dw, float* db, void* ws, size_t ws_bytes, hipStream_t stream);

/* Which kernel (name as rocprofv3 prints it) and K-split the planner uses for a block call of the given geometry:
 * kind 0 = gather GEMM (svs_enc_block_fwd / svs_dec_block_bwd_data), 1 = parity GEMM (svs_dec_block_fwd /
 * svs_enc_block_bwd_data), 2 = weight gradient (H,W,C: the strided image, N: channels of the windowed image).
 * Lets bench.py attribute its live HIP-event timings to the kernels of the rocprofv3 summary. */
int svs_describe_plan(int kind, int B, int H, int W, int C, int Ho, int Wo, int N, char* buf, size_t buflen);

/* ---------------------------------------------------------------------------------------------
 * Training-mode BatchNorm2d + activation (+ Dropout2d) -- model.py:49-50, 81-83 in .train().
 * raw is the (P = B*H*W, C) conv output.  svs_bn_stats writes per-workgroup partial sums,
 * svs_bn_finalize turns them into batch mean / 1/sqrt(biased var + eps), updates the running
 * statistics (momentum, UNBIASED variance) and num_batches_tracked, and svs_bn_act_apply writes
 * y = leaky((raw-mean)*invstd*gamma+beta, slope) * drop[b][c]   (drop == NULL: no dropout). */
size_t svs_bn_workspace_bytes(int64_t P, int C);
int svs_bn_stats(const float* raw, int64_t ldr, int64_t P, int C, void* ws, size_t ws_bytes, hipStream_t stream);
int svs_bn_finalize(const void* ws, int64_t P, int C, float eps, float momentum,
                    float* running_mean, float* running_var, int64_t* num_batches_tracked,
                    float* save_mean, float* save_invstd, hipStream_t stream);
int svs_bn_act_apply(const float* raw, int64_t ldr, int64_t P, int C, int64_t pixels_per_sample,
                     const float* gamma, const float* beta, const float* save_mean, const float* save_invstd,
                     float slope, const float* drop, float* y, int64_t ldy, hipStream_t stream);
/* Backward of (BN -> activation -> dropout): d_raw (P,C contiguous), dgamma, dbeta from dy (grad of y). */
int svs_bn_bwd(const float* dy, int64_t lddy, const float* raw, int64_t ldr, int64_t P, int C,
               int64_t pixels_per_sample, const float* gamma, const float* beta,
               const float* save_mean, const float* save_invstd, float slope, const float* drop,
               float* d_raw, float* dgamma, float* dbeta, void* ws, size_t ws_bytes, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Loss of the training step, train.py:274-283 with model.crit = nn.L1Loss() (config.py:33,44):
 *   loss = mean|mask*mix - voc| + mean|(1-mask)*mix - max(mix-voc,0)|
 * Writes the UNSCALED loss to *loss (device scalar) and d_logit = loss_scale * dloss/dmask * mask*(1-mask)
 * (the gradient w.r.t. deconv6's pre-sigmoid output; loss_scale = alpha_L1, train.py:24,296). */
size_t svs_l1_mask_loss_workspace_bytes(int64_t n);
int svs_l1_mask_loss_fwd_bwd(const float* mask, const float* mix, const float* voc, int64_t n, float loss_scale,
                             float* d_logit, float* loss, void* ws, size_t ws_bytes, hipStream_t stream);

/* torch.optim.Adam(lr, betas, eps), no weight decay / amsgrad (model.py:116), over flat buffers.
 * g is multiplied by grad_scale first (1/world for data-parallel mean). step is 1-based.  The hyper-parameters
 * are doubles because torch derives 1-beta and the bias corrections in double before rounding to fp32. */
int svs_adam_step(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2,
                  double eps, int step, float grad_scale, hipStream_t stream);

/* inference.py:100-107: out = mix * mask, or mix * (1 - mask) when invert != 0. */
int svs_apply_mask(const float* mix, const float* mask, float* out, int64_t n, int invert, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Whole-network entry points (host-side orchestration in C++; one call per forward / train step).
 * `params` is the flat fp32 buffer of the 46 trainable tensors in state_dict order (each tensor in
 * torch's layout; offsets from svs_unet_param_offset), `bn_buffers` the flat running_mean/var of the
 * 11 BatchNorms in state_dict order (mean then var per layer; offsets from svs_unet_buffer_offset). */
#define SVS_UNET_NUM_PARAMS 46
#define SVS_UNET_NUM_BN 11
int64_t svs_unet_param_offset(int tensor_index);      /* index == SVS_UNET_NUM_PARAMS -> total floats */
int64_t svs_unet_buffer_offset(int bn_index, int which /*0 mean, 1 var*/);  /* bn_index == 11 -> total */

/* Eval: pack weights + fold BN once per checkpoint, then run forwards. */
size_t svs_unet_prepared_bytes(void);
int svs_unet_prepare_eval(const float* params, const float* bn_buffers, void* prepared, hipStream_t stream);
size_t svs_unet_eval_workspace_bytes(int B, int H, int W);
/* mask = UNet.forward(mix) in eval mode -- model.py:169-201; mix, mask are (B,1,H,W). */
int svs_unet_forward_eval(const void* prepared, const float* mix, float* mask, int B, int H, int W,
                          void* ws, size_t ws_bytes, hipStream_t stream);

/* Train: forward (batch statistics, dropout) + L1 loss + backward, gradients into `grads` (flat,
 * same layout as params; overwritten).  Running stats in bn_buffers and num_batches_tracked (11
 * int64) are updated like torch does.  drop: 5 pointers' worth of masks laid out back to back
 * (B*256, B*128, B*64, B*32, B*16 floats) or NULL for no dropout.  The optimiser step is separate
 * (svs_adam_step) so that a data-parallel caller can all-reduce `grads` in between.
 * svs_unet_ws_offset reports where an intermediate lives inside ws (tests read them). */
size_t svs_unet_train_workspace_bytes(int B, int H, int W);
int svs_unet_train_fwd_bwd(const float* params, float* grads, float* bn_buffers, int64_t* num_batches_tracked,
                           const float* mix, const float* voc, const float* drop, int B, int H, int W,
                           float loss_scale, float* mask /*nullable*/, float* loss,
                           void* ws, size_t ws_bytes, hipStream_t stream);
/* Split form used by the autograd Function (train.py calls model(mix), builds the loss in torch and
 * calls .backward()): forward keeps its intermediates in ws; backward consumes d_mask. */
int svs_unet_train_forward(const float* params, float* bn_buffers, int64_t* num_batches_tracked,
                           const float* mix, const float* drop, int B, int H, int W, float* mask,
                           void* ws, size_t ws_bytes, hipStream_t stream);
int svs_unet_train_backward(const float* params, float* grads, const float* mix, const float* mask,
                            const float* d_mask, const float* drop, int B, int H, int W,
                            void* ws, size_t ws_bytes, hipStream_t stream);
/* Split form for overlapping the data-parallel gradient exchange with the backward pass: forward + loss, then
 * backward part 0 (decoder half: gradients of parameter tensors 24..45, i.e. grads[svs_unet_param_offset(24)..))
 * and part 1 (encoder half: tensors 0..23), or the encoder in two pieces: part 2 (the conv6 block, tensors 20..23 --
 * 13 of the encoder's 17.5 MB) then part 3 (tensors 0..19) -- or part 3 itself as part 5 (conv5 + conv4 blocks, tensors
 * 12..19, 4.1 MB) then part 6 (conv3..conv1, tensors 0..11, 0.26 MB: all that is left to exchange once the backward has
 * ended).  Part 4 is the whole backward in one call.  The caller starts the all-reduce of a piece on a second
 * stream as soon as the part is enqueued.  Same results as svs_unet_train_fwd_bwd.
 * Weight gradients are computed on a library-owned side stream.  The call that ends the pass (part 1, 3, 4 or 6) makes
 * `stream` wait for all of them; after an earlier part, svs_unet_train_bwd_sync(s) makes stream `s` (the one the
 * exchange is issued from) wait for the side-stream work enqueued so far WITHOUT stalling the backward's own stream
 * (`s` must also wait for `stream` itself, e.g. with an event).  The parts of one pass must be issued in order from one
 * host thread; one training pass per device at a time. */
int svs_unet_train_fwd_loss(const float* params, float* bn_buffers, int64_t* num_batches_tracked, const float* mix,
                            const float* voc, const float* drop, int B, int H, int W, float loss_scale,
                            float* mask /*nullable*/, float* loss, void* ws, size_t ws_bytes, hipStream_t stream);
int svs_unet_train_bwd_part(const float* params, float* grads, const float* mix, const float* drop, int B, int H, int W,
                            int part, void* ws, size_t ws_bytes, hipStream_t stream);
int svs_unet_train_bwd_sync(hipStream_t consumer);
/* The reference's FULL objective (train.py:274-296): alpha_l1 * L1 terms + alpha_mr * MultiResolutionSTFTLoss(
 * specific_istft(mask * mix, mix_phase), specific_istft(voc, voc_phase)) -- forward, both losses, d(total)/d(logit).
 * mix_phase / voc_phase: angles (B,1,512,W) (train.py:103-112).  Needs H = 512, W >= 2.  losses[0] = L1 part, losses[1] =
 * MR part, both unscaled.  Follow with svs_unet_train_bwd_part (part 4 = whole backward; or the split forms). */
size_t svs_unet_train_mr_workspace_bytes(int B, int W, int hop);
int svs_unet_train_fwd_loss_mr(const float* params, float* bn_buffers, int64_t* num_batches_tracked, const float* mix,
                               const float* voc, const float* mix_phase, const float* voc_phase, const float* drop, int B, int H,
                               int W, int hop, float alpha_l1, float alpha_mr, float* mask, float* losses, void* ws, size_t ws_bytes,
                               void* mr_ws, size_t mr_ws_bytes, hipStream_t stream);
int64_t svs_unet_ws_offset(const char* name, int B, int H, int W, int training);  /* bytes, <0 unknown */

/* bf16 eval forward (BASELINE configs[4]: "bf16 convs on MFMA"): the same network with bf16 NHWC activations and bf16
 * weights on v_mfma_f32_16x16x32_bf16, fp32 accumulation, BatchNorm folded, mix and mask still fp32.  Not bit-comparable with
 * the fp32 path (activations are rounded to 8 significant bits per layer); tests report its mask L1 against the fp32 forward.
 * prepared_f32 is the blob of svs_unet_prepare_eval. */
size_t svs_unet_prepared_bf16_bytes(void);
int svs_unet_prepare_eval_bf16(const void* prepared_f32, void* prepared_bf16, hipStream_t stream);
size_t svs_unet_eval_bf16_workspace_bytes(int B, int H, int W);
int svs_unet_forward_eval_bf16(const void* prepared_bf16, const float* mix, float* mask, int B, int H, int W, void* ws,
                               size_t ws_bytes, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Signal front/back end (replaces librosa.stft / magphase / istft at data.py:79-80,100-101,159 and
 * torch.istft at train.py:51-58).  n_fft-point periodic-Hann STFT, centred with zero padding,
 * frames = 1 + n_samples / hop.  mag is (n_fft/2+1, frames) float32 row-major; phase (optional)
 * is the unit phasor as interleaved (re, im) float32 pairs, 1+0j where the bin is exactly 0. */
int svs_stft_frames(int64_t n_samples, int hop);
int svs_stft_fwd(const float* y, int64_t n_samples, int n_fft, int hop, float* mag, float* phase,
                 hipStream_t stream);
/* y (hop*(frames-1) samples) = istft(mag * phase), window-sum-square normalised, both n_fft/2 edges
 * trimmed.  phase_is_angle != 0: phase holds angles in radians (train.py:45 torch.polar), else
 * interleaved unit phasors (data.py:159). */
size_t svs_istft_workspace_bytes(int n_fft, int hop, int frames);
int svs_istft(const float* mag, const float* phase, int phase_is_angle, int n_fft, int hop, int frames,
              float* y, void* ws, size_t ws_bytes, hipStream_t stream);
/* Batched / tiled forms (what the streaming path and the training loss use; the two calls above are the 1-channel,
 * (513, frames)-file special case).  A spectrogram is addressed as f-major tiles of `seg` frames and `rows` rows whose
 * first row is bin `first_bin` (0 or 1; rows == 513 - first_bin):
 *     element (channel c, bin k, frame t) = base[c * chan_stride + ((t / seg) * rows + (k - first_bin)) * seg + t % seg]
 * -> a (513, T) file is seg = T, rows = 513, first_bin = 0 (data.py:107); the network's input tiles (n, 1, 512, 128)
 * with the DC row dropped (inference.py:68,84; train.py:109-127) are seg = 128, rows = 512, first_bin = 1, so the
 * forward transform writes network tiles and the inverse reads them with no repacking pass.
 * svs_stft_tiles: y (channels, n_samples) -> mag; frames t >= 1 + n_samples / hop up to frames_alloc are written as
 *   zeros (tile padding, inference.py:90-92).  phase_mode 0: none; 1: frame-major unit phasors [c][t][513] (re, im) --
 *   the streaming form, read back by svs_istft_tiles; 2: f-major (513, T) phasors per channel (the .npy form).
 *   absmax_partial (optional): channels * svs_stft_groups(frames_alloc) per-block maxima of mag (reduce with svs_max). */
int svs_stft_tiles(const float* y, int64_t n_samples, int channels, int n_fft, int hop, float* mag, int64_t chan_stride,
                   int seg, int rows, int first_bin, int frames_alloc, float* phase, int phase_mode,
                   float* absmax_partial, hipStream_t stream);
int svs_stft_groups(int frames_alloc);
/* svs_istft_tiles: y (channels, hop * (frames - 1)) = istft(mag [* mask or * (1 - mask)] * phase).  mask (optional, same
 *   layout as mag) fuses inference.py:100-107.  phase_mode 1: frame-major phasors; 3: angles in the layout of mag
 *   (train.py:33-60, `specific_istft`: the DC row that train.py:41-42 pads back is the absent first_bin row).
 *   n_fft / 2 <= hop <= n_fft.  absmax_partial (optional): [channels][svs_istft_groups(hop, frames, channels)] maxima of |y|. */
int svs_istft_tiles(const float* mag, int64_t chan_stride, int seg, int rows, int first_bin, const float* mask, int invert,
                    const float* phase, int phase_mode, int channels, int n_fft, int hop, int frames, float* y,
                    float* absmax_partial, hipStream_t stream);
int svs_istft_groups(int hop, int frames, int channels);
/* (rows, cols) complex64 -> (cols, rows): f-major phasor files <-> the frame-major form */
int svs_transpose_c64(const float* in, float* out, int rows, int cols, hipStream_t stream);
/* Backward of `specific_istft` (train.py:33-60) fused with the chain rule of |S| = mask * mix (train.py:275,288):
 *   d_logit[b,f,t] += alpha * dL/d|S|[b,f,t] * mix * mask * (1 - mask);   d_wav (B, hop*(frames-1)); the rest (B,1,512,frames) */
int svs_istft_bwd_mask(const float* d_wav, const float* angle, const float* mix, const float* mask, float* d_logit,
                       float alpha, int B, int n_fft, int hop, int frames, hipStream_t stream);
/* Multi-resolution STFT loss of train.py:24-26,287-296 (auraloss.freq.MultiResolutionSTFTLoss with the reference's
 * arguments; restated from its published definition -- csrc/mrstft.hip).  x = predicted, y = target waveform, (B, L):
 *   loss[0] = MR-STFT(x, y);  d_x (optional) = grad_scale * d loss / d x. */
size_t svs_mrstft_workspace_bytes(int B, int64_t L);
int svs_mrstft_loss_fwd_bwd(const float* x, const float* y, int B, int64_t L, float grad_scale, float* loss, float* d_x,
                            void* ws, size_t ws_bytes, hipStream_t stream);
/* data.py:84-85,105 (divide by the mixture's maximum) and data.py:162-164 (peak-normalise to 0.9). */
int svs_absmax(const float* x, int64_t n, float* out /*device scalar*/, void* ws, size_t ws_bytes, hipStream_t stream);
/* out[0] = max(x[0..n)) for x >= 0 (the per-block partials of svs_stft_tiles / svs_istft_tiles) */
int svs_max(const float* x, int64_t n, float* out, hipStream_t stream);
int svs_scale_by_inv(float* x, int64_t n, const float* denom /*device scalar; 0 -> 1*/, float numer, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SVS_HIP_H */
