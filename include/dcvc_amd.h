/* dcvc_amd.h - C ABI of libdcvc_amd.so: the MI355X (gfx950) implementation of the DCVC-RT
 * per-frame encode / decode hot path.
 *
 * This is the drop-in boundary.  Plain pointers and sizes only; device
 * pointers are raw HIP device addresses (e.g. torch.Tensor.data_ptr()), `stream` is a
 * hipStream_t passed as void*.  Every function returns 0 on success or a negative code
 * (-1 bad argument, -2 HIP runtime error, -3 out of memory, -4 stream/decoder error) and never
 * throws; dcvc_last_error() returns a thread-local message.  No function allocates device
 * memory on the per-frame path: weights are packed at *_create, scratch is handed in by the
 * caller.  All kernels are deterministic (no atomics, fixed reduction order).
 *
 * Data layout inside the path: activations are "HWC" row-major [pixel][channel] with an explicit
 * row stride `ld` in ELEMENTS (so a producer can write into a slice of a channel-concat buffer);
 * element type is selected per object by `dtype` (DCVC_F16: _Float16 storage, fp32 MFMA
 * accumulate; DCVC_F32: float storage, fp32-input MFMA - bit-reproducible on a CPU).  Logical
 * channel counts must be multiples of 16 (outputs) / 32 (reduction dims); the Python host pads
 * 368 -> 384 and 514 -> 544 with zero weights.
 *
 * Each entry point cites the reference interface it replaces (paths relative to the reference
 * repository root).  INTEGRATION.md shows the reference-side binding for each seam.
 */
#ifndef DCVC_AMD_H
#define DCVC_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DCVC_ABI_VERSION 1

enum { DCVC_F16 = 0, DCVC_F32 = 1 };

enum {              /* epilogue of dcvc_conv_forward */
    DCVC_EPI_BIAS = 0,      /* out = conv + bias                                       */
    DCVC_EPI_BIAS_QUANT = 1,/* out = (conv + bias) * q[c]     bias_quant, cuda_inference.py:196 */
    DCVC_EPI_SHUFFLE2 = 2,  /* out = PixelShuffle(2)(conv + bias), SubpelConv2x layers.py:29-52 */
    DCVC_EPI_WSILU = 3      /* out = wsilu(conv + bias)                                 */
};

int dcvc_abi_version(void);
const char* dcvc_last_error(void);
/* number of HIP devices visible; <0 on error (used by the host code to fail loudly). */
int dcvc_device_count(void);

/* ------------------------------------------------------------------------------------------
 * Fused DepthConvBlock.
 * Replaces: DepthConvProxy (src/layers/extensions/inference/def.h:53-91, impl.cpp:7-121) and
 * DepthConvBlock.forward_torch (src/layers/layers.py:92-106), whose arithmetic order it follows
 * (adaptor not pre-multiplied into conv1; depthwise bias added before conv2).
 * Weights are HOST float32 pointers in the reference's torch layouts:
 *   adaptor_w [C][Cin] (NULL: no adaptor, then Cin == C), w1 [C][C], wd [C][3][3], w2 [C][C],
 *   w3 [4C][C], w4 [C][2C], biases [C] / [4C].
 * C and Cin are LOGICAL sizes; rows of the activation buffers hold round_up(.,32) channels. */
typedef struct dcvc_dcb dcvc_dcb;
typedef struct dcvc_conv dcvc_conv;   /* dense convolution layer, declared below */
int dcvc_dcb_create(int dtype, int cin, int c, int shortcut, const float* adaptor_w,
                    const float* adaptor_b, const float* w1, const float* b1, const float* wd,
                    const float* bd, const float* w2, const float* b2, const float* w3,
                    const float* b3, const float* w4, const float* b4, dcvc_dcb** out);
void dcvc_dcb_destroy(dcvc_dcb* h);
/* bytes of scratch dcvc_dcb_forward needs for an H x W input */
size_t dcvc_dcb_scratch_bytes(const dcvc_dcb* h, int H, int W);
/* x = concat(x0[:, :c0], x1[:, :c1]) (x1 may be NULL, c1 = 0); quant: device float[C] or NULL
 * (forward_with_quant_step, impl.cpp:91-97); out row stride ldo (forward_with_cat, impl.cpp:99-121:
 * the caller points `out` at the channel offset inside the concat buffer). */
int dcvc_dcb_forward(const dcvc_dcb* h, const void* x0, int64_t ld0, int c0, const void* x1,
                     int64_t ld1, int c1, int H, int W, const float* quant, void* out, int64_t ldo,
                     void* scratch, void* stream);
/* The same block inside a run of DepthConvBlocks that feed each other directly (the Sequential stacks of
 * video_model.py / image_model.py): the first 1x1 conv + WSiLU of a block is pointwise, so the PREVIOUS block
 * can compute it on its output tile before that tile leaves the CU.
 *   next != NULL : also produce next's pre-depthwise activation (needs: same width and type, next without
 *                  adaptor, this block without shortcut and without quant step); results are bit-identical
 *                  to calling the two blocks separately.
 *   head_done    : this block's own pre-depthwise activation was produced that way by the previous call.
 *   a_slot       : 0 / 1, alternating along the run (which half of the scratch holds this block's activation).
 * All calls of a run use the same scratch.  dcvc_dcb_forward(...) == (..., 0, 0, NULL). */
int dcvc_dcb_forward_chained(const dcvc_dcb* h, const void* x0, int64_t ld0, int c0, const void* x1,
                             int64_t ld1, int c1, int H, int W, const float* quant, void* out,
                             int64_t ldo, void* scratch, void* stream, int head_done, int a_slot,
                             const dcvc_dcb* next);

/* A run of DepthConvBlocks that ends in a 1x1 convolution of the same width (decoder.conv2 video_model.py:109, the
 * last layer of y_prior_fusion :199 and y_spatial_prior :212): the last block also computes that convolution
 * (bias, or bias * quant as dcvc_conv_forward would) on its output tile while the tile is still on the CU and writes
 * ONLY the convolution's output (conv_out, row stride ldco) - one launch and two passes over the feature map less;
 * the values are bit-identical to dcvc_dcb_forward_chained(..., next = NULL) followed by dcvc_conv_forward.
 * Needs: block without shortcut, conv 1x1 / stride 1 / cin = cout = the block's width. */
int dcvc_dcb_forward_then_conv(const dcvc_dcb* h, const void* x0, int64_t ld0, int c0, const void* x1,
                               int64_t ld1, int c1, int H, int W, void* scratch, void* stream,
                               int head_done, int a_slot, const dcvc_conv* conv, const float* conv_quant,
                               void* conv_out, int64_t ldco);

/* Measurement aid (bench.py roofline leg): runs the block `iters` times on `stream` with HIP events
 * recorded on that stream around each of its two kernels; returns the mean durations in ms. */
int dcvc_dcb_profile(const dcvc_dcb* h, const void* x0, int64_t ld0, int c0, int H, int W, void* out,
                     int64_t ldo, void* scratch, void* stream, int iters, float* head_ms, float* tail_ms);
/* The same for the second kernel (dcb_tail_kernel) alone: one head + tail launch, then `iters` back-to-back launches of
 * the tail between ONE pair of HIP events on `stream` (an event between two kernels costs 3-5 us of dispatch gap that a
 * per-launch bracket charges to the kernel); returns the mean time per launch in ms.  Blocks without adaptor only. */
int dcvc_dcb_profile_tail(const dcvc_dcb* h, const void* x0, int64_t ld0, int c0, int H, int W, void* out,
                          int64_t ldo, void* scratch, void* stream, int iters, float* tail_ms);

/* ------------------------------------------------------------------------------------------
 * Dense convolutions (implicit GEMM): 1x1, 3x3 (stride 1 or 2, pad 1), 2x2 stride 2.
 * Replaces: the at::conv2d calls of impl.cpp:61-80,133,148, video_model.py:63,127,162 and the
 * epilogue kernels bias_quant_cuda / bias_pixel_shuffle_2_cuda (kernel.cu:534,697).
 * w: HOST float32 [Cout][Cin][KH][KW]; b: HOST float32 [Cout].  For DCVC_EPI_SHUFFLE2 `cout` is
 * the conv's channel count (4 x the shuffled channel count). */
int dcvc_conv_create(int dtype, int cin, int cout, int kh, int kw, int stride, int pad, int epilogue,
                     const float* w, const float* b, dcvc_conv** out);
void dcvc_conv_destroy(dcvc_conv* h);
int dcvc_conv_forward(const dcvc_conv* h, const void* x0, int64_t ld0, int c0, const void* x1,
                      int64_t ld1, int c1, int H, int W, const float* quant, void* out, int64_t ldo,
                      void* stream);
/* The same with a per-input-channel factor applied while the input tile is staged: conv(x * in_scale[c]), the product
 * rounded to the element type exactly as dcvc_scale_channels would store it (so the values equal scale_channels followed
 * by dcvc_conv_forward; in_scale: DEVICE float32 [cin], NULL = plain forward).  Replaces the `ctx_t = conv1(x) * q_feature`
 * tensor of FeatureExtractor.forward (video_model.py:43-47), whose only reader is temporal_prior_encoder (:238, :311). */
int dcvc_conv_forward_scaled(const dcvc_conv* h, const void* x0, int64_t ld0, int c0, const void* x1,
                             int64_t ld1, int c1, int H, int W, const float* in_scale, const float* quant,
                             void* out, int64_t ldo, void* stream);

/* ------------------------------------------------------------------------------------------
 * Frame <-> feature layout kernels (dtype = element type of both sides).
 * pixel_unshuffle(8) of an NCHW frame into HWC features: video_model.py:68,276, image_model.py:33 */
int dcvc_unshuffle8(int dtype, const void* x_nchw, int C, int H, int W, void* out_hwc, int64_t ldo,
                    void* stream);
/* bias + PixelShuffle(8) + clamp[0,1] from HWC [H*W][C*64] to NCHW [C][8H][8W]:
 * bias_pixel_shuffle_8 (cuda_inference.py:182-193, kernel.cu:763); bias may be NULL. */
int dcvc_shuffle8_clamp(int dtype, const void* x_hwc, int64_t ld, const float* bias, int C, int H,
                        int W, int do_clamp, void* out_nchw, void* stream);
/* Frame I/O fused into one pass each.
 * Planar 8-bit YUV 4:2:0 -> the padded YCbCr 4:4:4 model input [3][H+pad_b][W+pad_r]: nearest chroma
 * upsampling (src/utils/transforms.py:13-24), /255 and cast (test_video.py:60-63,90), replicate pad
 * (test_video.py:179). */
int dcvc_yuv420_to_frame(int dtype, const uint8_t* y, const uint8_t* u, const uint8_t* v, int H, int W,
                         int pad_b, int pad_r, void* out_nchw, void* stream);
/* Reconstruction [3][Hp][Wp] -> planar 8-bit YUV 4:2:0 of the HxW picture (test_video.py:307-311):
 * crop, chroma 2x2 mean (transforms.py:56-63), clamp(x*255, 0, 255); Y rounded, U/V truncated like
 * the reference's `.to(uint8)` unless round_uv. */
int dcvc_frame_to_yuv420(int dtype, const void* x_nchw, int Hp, int Wp, int H, int W, int round_uv,
                         uint8_t* y, uint8_t* u, uint8_t* v, void* stream);
/* PNG (RGB) sources of the reference harness: uint8 planar RGB [3][H][W] -> the padded YCbCr model input (/255, BT.709
 * rgb2ycbcr in fp32 + clamp: src/utils/transforms.py:27-38, test_video.py:84-90, replicate pad :179) ... */
int dcvc_rgb_to_frame(int dtype, const uint8_t* rgb, int H, int W, int pad_b, int pad_r, void* out_nchw,
                      void* stream);
/* ... and a reconstruction [3][Hp][Wp] -> clamp(ycbcr2rgb(x) * 255, 0, 255) of the HxW picture, [3][H][W] in the
 * storage type (transforms.py:41-53, test_video.py:118-119; every operation rounded to the storage type like the
 * reference's fp16 tensors): the values its RGB PSNR, MS-SSIM and PNG writer read. */
int dcvc_frame_to_rgb(int dtype, const void* x_nchw, int Hp, int Wp, int H, int W, void* out_chw, void* stream);
/* right/bottom edge replication on HWC: replicate_pad (cuda_inference.py:174-179), pad_for_y */
int dcvc_replicate_pad_hwc(int dtype, const void* x, int64_t ldx, int H, int W, int C, int pad_b,
                           int pad_r, void* out, int64_t ldo, void* stream);
/* out[p][c] = x[p][c] * q[c]      (ctx_t = x1 * quant, video_model.py:46) */
int dcvc_scale_channels(int dtype, const void* x, int64_t ldx, const float* q, int64_t P, int C,
                        void* out, int64_t ldo, void* stream);
/* strided 2-D copy of a channel slice (torch.cat halves, crops) */
int dcvc_copy_channels(int dtype, const void* x, int64_t ldx, int64_t P, int C, void* out,
                       int64_t ldo, void* stream);
/* crop HWC [H][W] -> [H2][W2] (hierarchical_params[:, :, :H, :W], video_model.py:283) */
int dcvc_crop_hwc(int dtype, const void* x, int64_t ldx, int W, int H2, int W2, int C, void* out,
                  int64_t ldo, void* stream);

/* ------------------------------------------------------------------------------------------
 * Entropy-model glue, HWC pipeline form (masks are computed from (h, w, c), never stored).
 *
 * z quantiser: z_hat = clamp(round(z)) in place + int8 copy in the reference's CHW order.
 * round_and_to_int8 (cuda_inference.py:26-33, kernel.cu:828). */
int dcvc_round_z(int dtype, void* z, int64_t ld, int H, int W, int C, int8_t* z_chw, void* stream);
/* int8 CHW -> HWC (get_z, entropy_models.py:221-224) */
int dcvc_z_from_int8(int dtype, const int8_t* z_chw, int H, int W, int C, void* out, int64_t ldo,
                     void* stream);

/* One checkerboard step of the ENCODER prior loop.  n_groups = 2 (video, compress_prior_2x,
 * common_model.py:143-161) or 4 (intra, compress_prior_4x, :206-256); `step` selects mask_step.
 *   q_mode 0: yq = y * (1 / max(qdec, 0.5)) per element      (separate_prior_for_video_encoding)
 *   q_mode 1: yq = y * q_enc[p] with q_enc = sigmoid(qraw[p][0]) * 1.5 + 0.5  (separate_prior :70-71)
 * Computes process_with_mask (cuda_inference.py:58-74), collapses the n_groups channel groups
 * (single_part_for_writing_*), builds the packed int16 symbol (build_index_enc :146-171) and
 * writes it in CHW order of the collapsed tensor; skipped symbols (scale <= thres) get the
 * sentinel low byte 0xFF, which dcvc_rans_encode_y drops (= the reference's boolean-mask
 * compaction `out[skip_cond]`).  y_hat accumulates:  yhat_out = (step == 0 ? 0 : yhat_in) + y_hat.
 * thres < 0 disables skipping. */
int dcvc_prior_enc_step(int dtype, int n_groups, int step, int q_mode, const void* y, int64_t ldy,
                        const void* qsrc, int64_t ldq, const void* scales, int64_t lds_,
                        const void* means, int64_t ldm, int H, int W, int C, float thres,
                        const void* yhat_in, int64_t ldhi, void* yhat_out, int64_t ldho,
                        int16_t* packed_chw, void* stream);
/* DECODER side: indexes for one step (combine_for_reading_* + build_index_dec, :77-143);
 * uint8 in CHW order of the collapsed tensor, 0xFF = skipped. */
int dcvc_prior_dec_index(int dtype, int n_groups, int step, const void* scales, int64_t lds_, int H,
                         int W, int C, float thres, uint8_t* idx_chw, void* stream);
/* DECODER side: restore_y_2x / restore_y_4x (+ running sum):
 *   yhat_out = (step == 0 ? 0 : yhat_in) + (sym + means) * mask_step   (common_model.py:196-203,270-292) */
int dcvc_prior_dec_restore(int dtype, int n_groups, int step, const int8_t* sym_chw,
                           const void* means, int64_t ldm, int H, int W, int C, const void* yhat_in,
                           int64_t ldhi, void* yhat_out, int64_t ldho, void* stream);
/* DECODER hand-off with the kept entries compacted on the device (no reference counterpart: the reference gathers the kept
 * indexes with a boolean mask, copies them with .cpu() and scatters the decoded symbols back, entropy_models.py:330-341 -
 * same stream order; the round-3 path copied the whole index / symbol arrays, ~96 % sentinels).
 *   dcvc_prior_dec_index_compact: as dcvc_prior_dec_index (idx_chw stays on the DEVICE), then the kept indexes in CHW order
 *     are written by a kernel into idx_host[0 .. *count_host) and their number into *count_host - both dcvc_host_alloc
 *     buffers (idx_host: one byte per position rounded up to 16), valid for the host once the stream has passed this point.
 *   host: dcvc_rans_dec_decode_compact decodes exactly *count_host symbols into a pinned int8 buffer (16-byte aligned,
 *     capacity as idx_host).
 *   dcvc_prior_dec_restore_compact: fetches those symbols (coalesced 16-byte reads over the host link) and restores y_hat as
 *     dcvc_prior_dec_restore does; idx_chw and workspace must be the ones of the step's index call.
 * workspace: device, 16-byte aligned, dcvc_prior_dec_compact_ws_bytes(H, W, C, n_groups) bytes. */
int64_t dcvc_prior_dec_compact_ws_bytes(int H, int W, int C, int n_groups);
int dcvc_prior_dec_index_compact(int dtype, int n_groups, int step, const void* scales, int64_t lds_, int H, int W, int C,
                                 float thres, uint8_t* idx_chw, void* workspace, uint8_t* idx_host,
                                 int32_t* count_host, void* stream);
int dcvc_prior_dec_restore_compact(int dtype, int n_groups, int step, const int8_t* sym_host, const uint8_t* idx_chw,
                                   void* workspace, const void* means, int64_t ldm, int H, int W, int C,
                                   const void* yhat_in, int64_t ldhi, void* yhat_out, int64_t ldho, void* stream);
/* y_hat = y_hat * q   q_mode 0: max(qdec,0.5) per element; q_mode 1: sigmoid(qraw[p][1])*1.5+0.5
 * add_and_multiply (cuda_inference.py:48-55), common_model.py:246,294 */
int dcvc_prior_finish(int dtype, int q_mode, void* yhat, int64_t ldh, const void* qsrc, int64_t ldq,
                      int H, int W, int C, void* stream);

/* ------------------------------------------------------------------------------------------
 * Operator-module seam (inference_extensions_cuda, bind.cpp:7-36): flat NCHW-contiguous
 * elementwise kernels with the reference's own signatures (def.h:6-51).  n = element count. */
int dcvc_op_process_with_mask(int dtype, const void* y, const void* scales, const void* means,
                              const void* mask, float thres, void* y_res, void* y_q, void* y_hat,
                              void* s_hat, int64_t n, void* stream);
int dcvc_op_combine_for_reading_2x(int dtype, void* out, const void* x, const void* mask,
                                   int64_t half_n, void* stream);
int dcvc_op_restore_y_2x(int dtype, void* out, const void* y, const void* means, const void* mask,
                         int64_t half_n, void* stream);
int dcvc_op_restore_y_4x(int dtype, void* out, const void* y, const void* means, const void* mask,
                         int64_t quarter_n, void* stream);
int dcvc_op_build_index_dec(int dtype, uint8_t* out, uint8_t* cond_out, const void* scales,
                            float scale_min, float scale_max, float log_scale_min,
                            float log_step_recip, float skip_thres, int64_t n, void* stream);
int dcvc_op_build_index_enc(int dtype, int16_t* out, uint8_t* cond_out, const void* symbols,
                            const void* scales, float scale_min, float scale_max,
                            float log_scale_min, float log_step_recip, float skip_thres, int64_t n,
                            void* stream);
int dcvc_op_round_and_to_int8(int dtype, void* z, int8_t* z_int8, int64_t n, void* stream);
int dcvc_op_clamp_reciprocal_with_quant(int dtype, const void* q_dec, void* y, float min_val,
                                        void* q_out, int64_t n, void* stream);
int dcvc_op_add_and_multiply(int dtype, void* x0, const void* x1, const void* q, int64_t n,
                             void* stream);
int dcvc_op_bias_quant(int dtype, void* x, const void* bias, const void* quant, int C, int64_t HW,
                       void* stream);
int dcvc_op_bias_pixel_shuffle_8(int dtype, void* out, const void* x, const void* bias, int C, int H,
                                 int W, int do_clamp, void* stream);
int dcvc_op_replicate_pad(int dtype, const void* x, int C, int H, int W, int pad_b, int pad_r,
                          void* out, void* stream);
int dcvc_op_bias_wsilu_depthwise_conv2d(int dtype, const void* x, const void* weight,
                                        const void* bias, int C, int H, int W, void* out,
                                        void* stream);
/* NCHW <-> HWC transposes used by the operator-seam proxies */
int dcvc_nchw_to_hwc(int dtype, const void* x, int C, int64_t HW, void* out, int64_t ldo, void* stream);
int dcvc_hwc_to_nchw(int dtype, const void* x, int64_t ldx, int C, int64_t HW, void* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Host entropy coder (rANS).  Replaces MLCodec_extensions_cpp (src/cpp/py_rans/py_rans.cpp:14-393,
 * rans.cpp:60-534); the byte stream is identical to the reference's.  One worker thread per
 * coder half, like RansEncoderLibMultiThread / RansDecoderLibMultiThread; calls enqueue and
 * return, dcvc_rans_get_* block.  Input buffers are copied on entry unless the entry point says
 * otherwise (*_borrowed, decode_and_get_y). */
typedef struct dcvc_rans_enc dcvc_rans_enc;
typedef struct dcvc_rans_dec dcvc_rans_dec;
dcvc_rans_enc* dcvc_rans_enc_create(void);
void dcvc_rans_enc_destroy(dcvc_rans_enc*);
/* cdf [n][stride] int32, sizes [n], offsets [n] -> group index (add_cdf, py_rans.cpp:69-95) */
int dcvc_rans_enc_add_cdf(dcvc_rans_enc*, const int32_t* cdf, int n, int stride,
                          const int32_t* sizes, const int32_t* offsets);
/* forgets every table group; the next add_cdf returns 0 again (empty_cdf_buffer, py_rans.cpp:97-101) */
int dcvc_rans_enc_empty_cdf(dcvc_rans_enc*);
void dcvc_rans_enc_set_use_two(dcvc_rans_enc*, int two);
int dcvc_rans_enc_reset(dcvc_rans_enc*);
/* symbols: (int8 symbol << 8) + uint8 cdf index (encode_y, py_rans.cpp:20-41).
 * Entries whose low byte is 0xFF are dropped first (see dcvc_prior_enc_step). */
int dcvc_rans_enc_encode_y(dcvc_rans_enc*, const int16_t* symbols, int64_t n, int group);
/* same, without the copy: `symbols` must stay valid and unchanged until dcvc_rans_enc_get_stream
 * has returned (the pinned staging buffer of the D2H hand-off qualifies) */
int dcvc_rans_enc_encode_y_borrowed(dcvc_rans_enc*, const int16_t* symbols, int64_t n, int group);
int dcvc_rans_enc_encode_z(dcvc_rans_enc*, const int8_t* symbols, int64_t n, int group,
                           int start_offset, int per_channel_size);
int dcvc_rans_enc_flush(dcvc_rans_enc*);
/* blocks until the flush completed; returns the stream length, *data valid until next reset */
int64_t dcvc_rans_enc_get_stream(dcvc_rans_enc*, const uint8_t** data);

dcvc_rans_dec* dcvc_rans_dec_create(void);
void dcvc_rans_dec_destroy(dcvc_rans_dec*);
int dcvc_rans_dec_add_cdf(dcvc_rans_dec*, const int32_t* cdf, int n, int stride,
                          const int32_t* sizes, const int32_t* offsets);
int dcvc_rans_dec_empty_cdf(dcvc_rans_dec*);   /* py_rans.cpp:291-295 */
void dcvc_rans_dec_set_use_two(dcvc_rans_dec*, int two);
int dcvc_rans_dec_set_stream(dcvc_rans_dec*, const uint8_t* data, int64_t n);
/* indexes: uint8 cdf index per symbol, 0xFF = skipped (decodes nothing, yields 0).
 * The result has one int8 per index (decode_y + the scatter-back of get_y,
 * entropy_models.py:330-341). */
int dcvc_rans_dec_decode_y(dcvc_rans_dec*, const uint8_t* indexes, int64_t n, int group);
int dcvc_rans_dec_decode_z(dcvc_rans_dec*, int64_t total, int group, int start_offset,
                           int per_channel_size);
/* blocks; copies the decoded int8 symbols of the last decode_* call into out[0..n) */
int64_t dcvc_rans_dec_get(dcvc_rans_dec*, int8_t* out, int64_t capacity);
/* decode_and_get_y (py_rans.cpp:175-262 decode_y + get_decoded_tensor in one call): synchronous,
 * no staging copies - reads `indexes` and writes out[0..n) directly; the first coder runs on the
 * calling thread, the second on its worker. */
int dcvc_rans_dec_decode_and_get_y(dcvc_rans_dec*, const uint8_t* indexes, int64_t n, int group,
                                   int8_t* out);
/* The compacted form of decode_and_get_y: `indexes` holds ONLY the kept entries (count of them, stream order, as
 * dcvc_prior_dec_index_compact writes them); out[0..count) receives one int8 each.  Same coder split as above (the first
 * coder takes floor(count / 2) symbols): the two forms consume a stream identically. */
int dcvc_rans_dec_decode_compact(dcvc_rans_dec*, const uint8_t* indexes, int64_t count, int group, int8_t* out);
/* Integrity check after the LAST symbol of a frame has been decoded (no reference counterpart: the reference decodes a
 * corrupt or truncated payload into garbage, rans.cpp:356-429): a rANS decoder that has undone every encoder step is back
 * in the encoder's initial state and has consumed every byte; returns 0, or -4 with dcvc_last_error() set. */
int dcvc_rans_dec_check_end(dcvc_rans_dec*);
/* pmf_to_quantized_cdf (py_rans.cpp:307-364); out holds n+1 entries */
int dcvc_pmf_to_quantized_cdf(const float* pmf, int n, int precision, uint32_t* out);

/* pinned host staging buffers for the device <-> coder hand-off (hipHostMalloc / hipHostFree) */
void* dcvc_host_alloc(size_t bytes);
void dcvc_host_free(void* p);
/* the device address of a dcvc_host_alloc buffer (kernels may read / write pinned host memory in place) */
void* dcvc_host_device_ptr(void* host);
/* Encoder symbol hand-off without a copy command (replaces the reference's boolean-mask compaction + .cpu(),
 * src/layers/cuda_inference.py:159, src/models/entropy_models.py:46-52, and its device synchronisation for the size):
 * `packed` = n_parts arrays of n_per_part int16 (sym << 8 | index, low byte 0xFF = skipped, as dcvc_prior_enc_step writes
 * them).  The kept entries of part p are written IN ORDER to out_host[p * n_per_part ...] and their number to
 * counts_host[p]; both are dcvc_host_alloc buffers that the kernels write in place over the host link.  workspace: device,
 * DCVC_COMPACT_BLOCKS * n_parts int32.  The host may read the buffers once the stream has passed this point. */
#define DCVC_COMPACT_BLOCKS 256
int dcvc_compact_symbols(const int16_t* packed, int n_per_part, int n_parts, int16_t* out_host, int32_t* counts_host,
                         int32_t* workspace, void* stream);
/* dst[0..n) = src[0..n) on the device by a kernel (the per-frame row of the quantisation tables: src/models/video_model.py:303-305
 * slices them per call; a runtime copy command costs an order of magnitude more than the kernel) */
int dcvc_copy_f32(float* dst, const float* src, int n, void* stream);
/* stream-ordered copies (hipMemcpyAsync) and event-free host wait for a stream */
int dcvc_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes, void* stream);
int dcvc_memcpy_h2d(void* dst_dev, const void* src_host, size_t bytes, void* stream);
int dcvc_stream_sync(void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DCVC_AMD_H */
