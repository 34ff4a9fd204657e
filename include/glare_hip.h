/*
 * libglare_hip.so -- C ABI of the MI355X-native GLARE hot path.
 *
 * This is the drop-in boundary.  In the reference the only native
 * boundary is the pybind11 module `deform_conv_ext`
 * (code/models/modules/ops/dcn/src/deform_conv_ext.cpp:150-164) called from
 * code/models/modules/ops/dcn/deform_conv.py:66,89,97,149,166; everything else on the path
 * is a torch op called from the nn.Module surface named in BASELINE.json
 * (VectorQuantizer2, FlowUpsamplerNet, the VQGAN Encoder/Decoder, MultiScaleDecoder2).
 * Each entry point below cites the reference interface it replaces.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only; no torch / C++ types.
 *   - every pointer is a DEVICE pointer unless the name ends in _host.
 *   - the caller owns every buffer, including scratch ("workspace"); the library keeps no
 *     state between calls, is re-entrant and never synchronises the device.
 *   - work is enqueued on `stream` (a hipStream_t passed as void*; NULL = default stream).
 *   - return value: GLARE_OK (0) or a negative GLARE_ERR_* code; nothing throws, nothing is
 *     only printed (the reference merely printf()s launch failures,
 *     deform_conv_cuda_kernel.cu:794-798).
 *   - "NHWC" tensors are [B][H][W][pitch] with the used channels at [off, off+C) of each
 *     pixel's `pitch`-element record; bf16 elements are raw uint16_t.
 */
#ifndef GLARE_HIP_H
#define GLARE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* glare_stream_t; /* hipStream_t */

#define GLARE_OK 0
#define GLARE_ERR_INVALID (-1)     /* bad pointer / shape / inconsistent sizes */
#define GLARE_ERR_LAUNCH (-2)      /* HIP reported a launch or attribute failure */
#define GLARE_ERR_WORKSPACE (-3)   /* workspace too small; see the *_workspace_bytes query */
#define GLARE_ERR_UNSUPPORTED (-4) /* valid request outside what the kernels implement */

/* Library ABI version (major*100 + minor). */
int glare_version(void);
/* Static description of a status code. */
const char* glare_status_string(int status);

/* ---- a5: codebook retrieval -------------------------------------------------------------
 * Replaces the distance/argmin/gather of VectorQuantizer2.forward
 * (code/models/modules/quantize.py:276-285).
 * z_nhwc   [n_tokens][dim] fp32 (the 'b c h w -> b h w c' flattening of quantize.py:276-277)
 * codebook [n_codes][dim]  fp32 (embedding.weight)
 * idx_i64  [n_tokens] int64 (min_encoding_indices, quantize.py:284)
 * zq_nhwc  [n_tokens][dim] fp32 or NULL (embedding(idx), quantize.py:285)
 * Indices are bit-exact against the reference's fp32 CPU arithmetic; ties -> lowest index.
 * dim must be 3 (VQModel_arch.py:44-45, confs/LOL.yml:89). */
int glare_vq_nearest_f32(const float* z_nhwc, const float* codebook, long long n_tokens,
                         int n_codes, int dim, long long* idx_i64, float* zq_nhwc,
                         glare_stream_t stream);

/* ---- a1/a6/a8: im2col-free direct convolution on bf16 MFMA ----------------------------------
 * Replaces torch.nn.Conv2d / F.conv2d for the 3x3 and 1x1 convolutions of the VQGAN
 * encoder/decoder, AttnBlock projections, WarpBlock.offset, DCNv2Pack.conv_offset and the flow
 * coupling nets (encoder_decoder.py:43-52,62-75,88-115,146-165; deformableDecoder_arch.py:282;
 * deform_conv.py:357-364; flow.py:13-70).  fp32 accumulation, bf16 NHWC activations.
 * Fused in the loader: zero padding (pad 1 for 3x3 stride 1; (0,1,0,1) for stride 2 ==
 * Downsample, encoder_decoder.py:71-73), nearest x2 upsampling of the input (Upsample,
 * encoder_decoder.py:50), channel concatenation of two sources (torch.cat, deformableDecoder_arch.py:286).
 * Fused in the epilogue: bias, residual add (encoder_decoder.py:137,192), activation, layout/dtype. */
#define GLARE_ACT_NONE 0
#define GLARE_ACT_RELU 1
#define GLARE_ACT_SIGMOID 2
#define GLARE_ACT_SWISH 3

#define GLARE_OUT_NHWC_BF16 0   /* out[pixel][out_pitch] bf16                              */
#define GLARE_OUT_NHWC_F32 1    /* out[pixel][out_pitch] fp32                              */
#define GLARE_OUT_PLANAR_F32 2  /* out[b][out_off+co][plane_pitch] fp32 (NCHW)             */
#define GLARE_OUT_PLANAR_BF16 3 /* out[b][out_off+co][plane_pitch] bf16 (e.g. V^T of attention) */

typedef struct glare_conv_desc {
  const void* in;            /* bf16 NHWC [B][H][W][in_pitch], channels [in_off, in_off+Cin)       */
  const void* in2;           /* optional second source concatenated after `in` along C, or NULL    */
  const void* weight_packed; /* from glare_conv2d_pack_weight (Cin = Cin + Cin2)                   */
  const float* bias;         /* [Cout] fp32 or NULL                                                */
  const void* residual;      /* bf16 NHWC [B][OH][OW][res_pitch] added before `act`, or NULL       */
  void* out;                 /* see out_mode                                                       */
  int B, H, W;               /* source size; conv input is 2H x 2W when upsample != 0              */
  int Cin, in_pitch, in_off;
  int Cin2, in2_pitch, in2_off;
  int Cout, out_pitch, out_off; /* planar modes: out_pitch = number of planes per image            */
  int res_pitch, res_off;
  int ksize;                 /* 1 or 3                                                             */
  int stride;                /* 1 (pad ksize/2) or 2 (3x3 only, pad (0,1,0,1))                     */
  int upsample;              /* 0: none; 1: nearest x2 on the input first (Upsample, encoder_decoder.py:43-52);   */
                             /* 2: the same operator in SUB-PIXEL form (3x3, stride 1, bf16 NHWC out): four 2x2   */
                             /* convs of the source, 16 instead of 36 tap-MACs per source pixel; weight_packed    */
                             /* from glare_conv2d_pack_weight_upsample, gn_partial sized / reduced by the         */
                             /* glare_conv2d_upsample_gn_* pair                                                   */
  int act;                   /* GLARE_ACT_*.  SIGMOID / SWISH without a residual run on the accumulators in the general  */
                             /* (element-wise) epilogue (round 5: one epilogue per kernel instantiation): such a launch   */
                             /* cannot ALSO ask for gn_partial or upsample == 2 (both live in the 16-B slab epilogue) --  */
                             /* GLARE_ERR_UNSUPPORTED; with a residual, or with NONE / RELU, every combination holds.     */
                             /* The path fuses sigmoid / swish only into the small-Cin convs (glare_conv2d_smallcin_*)    */
  int out_mode;              /* GLARE_OUT_*                                                        */
  long long plane_pitch;     /* planar modes: elements per plane (>= OH*OW); 0 = OH*OW             */
  float* gn_partial;         /* optional: fused GroupNorm statistics of the OUTPUT (bf16 NHWC, Cout % 128 == 0): */
                             /* glare_conv2d_gn_partial_elems() floats, reduced by glare_conv2d_gn_reduce()       */
  int cout_tile;             /* 0: the default output-channel tile of the workgroup for this Cout (128 / 64 / 32); else  */
                             /* the tile the weights were packed for (glare_conv2d_pack_weight_batched's cout_tile):     */
                             /* small launches (training crops, batch 1) fill the chip better with 64- / 32-wide tiles,  */
                             /* glare_conv2d_cout_tile() picks                                                            */
  /* ---- hi / lo output (round 3; zero-initialise the struct if unused) -----------------------------------------------------
   * The residual stream of the conditional encoder kept to 22 mantissa bits in two 16-bit tensors: value = hi + lo, `hi` the
   * ordinary activation tensor every consumer reads (convs, attention), `lo` the rounding remainder that only residual adds and
   * GroupNorm read back.  out_lo != NULL (GLARE_OUT_NHWC_BF16, 3x3, Cout tile 128, stride 1 or 2): `out` receives
   * hi = round16(v), out_lo (same pitch / offset) lo = round16(v - hi) of v = acc + bias (+ residual + residual_lo) after `act`,
   * without the intermediate 16-bit rounding of acc + bias the plain epilogue has; the fused GroupNorm statistics are those of v. */
  const void* residual_lo;   /* optional lo half of `residual` (same pitch / offset), or NULL      */
  void* out_lo;              /* optional: see above                                                */
  /* ---- grouped launch (round 3; zero = one filter): `groups` independent filters of ONE shape applied to channel slices of one
   * tensor in a single launch -- group g reads input channels [in_off + g * group_in_step, + Cin), writes output channels
   * [out_off + g * group_out_step, + Cout), with the g-th of `groups` consecutive packed filters in weight_packed
   * (glare_conv2d_pack_weight_batched) and the g-th Cout-vector of bias.  No residual or fused statistics; a second source only as the
   * lo half of a k_wrap launch (shifted like the first); out_lo follows out. */
  int groups, group_in_step, group_out_step;
  /* ---- fp32-class contraction on the 16-bit MFMA (round 4; zero = plain): with the activation and the filter each a hi / lo pair
   * (22 mantissa bits), x . w = x_hi . w_hi + x_lo . w_hi + x_hi . w_lo up to 2^-22, i.e. ONE accumulation over three K segments.
   * k_wrap != 0: after the concatenated sources (`in`, then `in2`) the FIRST source is read again, so that with in = x_hi,
   * in2 = x_lo (same geometry, Cin2 = Cin) and weight_packed = the pack of [w_hi | w_hi | w_lo] along the input channels (3 Cin) the
   * launch computes exactly that sum; with in2 = NULL it is x_hi . (w_hi + w_lo) on a filter packed as [w_hi | w_lo] (2 Cin).
   * k_wrap == 2 (round 5; needs in2): the same three products with every halo tile of `in` staged ONCE -- stages 2c and 2c + 1 contract
   * chunk c of `in` (one stage = 16 channels for 3x3) with the 2c-th and (2c + 1)-th chunk of the packed filter, the `in2` segment
   * follows: weight_packed = the pack of [w_hi(c0) | w_lo(c0) | w_hi(c1) | w_lo(c1) | ... | w_hi] (3 Cin).  3x3 (either stride), Cin % 16 == 0,
   * Cin2 == Cin, no upsample / gn_coef (GLARE_ERR_UNSUPPORTED / GLARE_ERR_INVALID otherwise); any other k_wrap value is GLARE_ERR_INVALID.
   * The packed filter's input-channel count must be Cin + Cin2 + Cin; Cin (and Cin2) multiples of the kernel's 16-channel (3x3) /
   * 32-channel (1x1) stage.  Grouped launches shift both sources by group_in_step.  Where the reference contracts in fp32 and the
   * codebook search downstream needs it (the conditional encoder and the flow's nets under the fp16 inference precision). */
  int k_wrap;
  /* ---- GroupNorm (+ swish) of the INPUT as the loader's prologue (round 4; NULL = none): gn_coef = fp32 [B][Cin][2] from
   * glare_groupnorm_coeffs_f32, (a, d) per (image, channel); the kernel applies y = swish(a x + d) (gn_swish != 0) or y = a x + d,
   * rounded to 16 bits exactly as glare_groupnorm_apply_bf16 would have stored it, to its halo tile in LDS -- zero padding stays zero --
   * so `in` is the RAW tensor and the normalised one is never written (encoder_decoder.py:119-120,126-127: Normalize + swish in front
   * of conv1 / conv2).  3x3 stride 1, one source, 128-wide tile, single pass only. */
  const float* gn_coef;
  int gn_swish;
} glare_conv_desc;

/* The output-channel tile (128, 64 or 32) that gives a B x OH x OW x cout conv enough workgroups (8 x 32 output pixels each). */
int glare_conv2d_cout_tile(int B, int OH, int OW, int cout);
/* glare_conv2d_packed_weight_elems for an explicit tile (0 = default). */
long long glare_conv2d_packed_weight_elems_tile(int cout, int cin_total, int ksize, int cout_tile);

/* Number of bf16 elements of the packed weight image for an OIHW [cout][cin_total][k][k] filter. */
long long glare_conv2d_packed_weight_elems(int cout, int cin_total, int ksize);
/* Packs fp32 OIHW weights (device) into the kernel's stage-ordered bf16 image (device). */
/* Filter rounding with error feedback (round 6): out = the filter rounded to the library's 16-bit format (as fp32 values) with the rounding
 * error of each weight carried into the next along (cin, ky, kx) of its output channel -- every weight within one 16-bit ulp of the
 * channel's largest weights, the sum of a channel's errors below half such an ulp.  A filter's rounding is the same perturbation at every pixel; this keeps each output channel's
 * response to the input's mean exact.  Inference packs its single-pass filters from this (glare_amd.modules._base.packed_conv). */
int glare_filter_feedback_round_bf16(const float* w_oihw, float* out_oihw, int cout, long long elems_per_cout, glare_stream_t stream);
int glare_conv2d_pack_weight(const float* w_oihw, int cout, int cin_total, int ksize, void* packed_bf16,
                             glare_stream_t stream);
/* Packs the filter of the DATA-GRADIENT convolution straight from the forward OIHW filter: w'[ci][co][tap] =
 * w[co][ci][k*k-1-tap], co padded with zeros to cout_padded (% 8 == 0).  The result is a packed filter for a conv with
 * cout_padded input and cin output channels (glare_conv2d_packed_weight_elems(cin, cout_padded, ksize) elements). */
int glare_conv2d_pack_weight_dgrad(const float* w_oihw, int cout, int cin, int ksize, int cout_padded, void* packed_bf16,
                                   glare_stream_t stream);
/* Filters of DIFFERENT shapes packed in ONE launch: the trainable convs of a training step, whose packed images live across steps and
 * are refreshed after the optimizer update (the reference re-reads its fp32 weights through cuDNN every step; here the MFMA kernels
 * read stage-ordered bf16 images).  A job = one packed image: glare_conv2d_pack_job_init fills its geometry (kind FORWARD as
 * glare_conv2d_pack_weight, DGRAD as glare_conv2d_pack_weight_dgrad with dgrad_cout_padded, PLAIN_BF16 as
 * glare_conv1x1_ws_pack_weight); the caller sets block_begin = sum over the earlier jobs of ceil(total / GLARE_PACK_BLOCK_ELEMS), copies the table to
 * the device and launches it with the total block count. */
enum { GLARE_PACK_FORWARD = 0, GLARE_PACK_DGRAD = 1, GLARE_PACK_PLAIN_BF16 = 2 };
#define GLARE_PACK_BLOCK_ELEMS 2048   /* packed elements per block of glare_conv2d_pack_multi (8 per thread: one 16-B chunk) */
typedef struct glare_pack_job {
  const float* w;          /* fp32 OIHW filter */
  void* out;               /* packed bf16 image */
  long long total;         /* elements of the packed image */
  long long block_begin;   /* first block (of GLARE_PACK_BLOCK_ELEMS packed elements) of this job in the launch */
  int cout, cin, ksize;    /* of the PACKED conv (DGRAD: outputs = the forward conv's inputs) */
  int tn, ksteps, n_stages, cin_real, kind;
} glare_pack_job;
int glare_conv2d_pack_job_init(glare_pack_job* job, int kind, const float* w_oihw, int cout, int cin, int ksize, int dgrad_cout_padded,
                               int cout_tile, void* packed_bf16);
int glare_conv2d_pack_multi(const glare_pack_job* jobs_device, int n_jobs, long long total_blocks, glare_stream_t stream);
/* `batch` filters of one shape ([batch][cout][cin][k][k] fp32, consecutive) packed in one launch into consecutive packed images:
 * dgrad_cout_padded = 0 as glare_conv2d_pack_weight, > 0 as glare_conv2d_pack_weight_dgrad with that padding (the per-step convs
 * of the flow's coupling nets, FlowAffineCouplingsAblation.py:117-160: 24 steps x 4 convs of two shapes); cout_tile = 0 or the
 * workgroup tile to pack for (glare_conv_desc.cout_tile of the launches that use the image). */
int glare_conv2d_pack_weight_batched(const float* w_boihw, int batch, int cout, int cin, int ksize, int dgrad_cout_padded,
                                     int cout_tile, void* packed_bf16, glare_stream_t stream);
/* Sub-pixel filters of "nearest x2 upsample, then 3x3 conv": per output phase (row parity a, column parity b) the 3x3 taps
 * that read the same source pixel are summed in fp32 and rounded to bf16 once (rows a=0: {0},{1,2}; a=1: {0,1},{2}). */
long long glare_conv2d_upsample_packed_weight_elems(int cout, int cin_total);
int glare_conv2d_pack_weight_upsample(const float* w_oihw, int cout, int cin_total, void* packed_bf16, glare_stream_t stream);
/* ... from the four phase filters themselves: w_phases fp32 [4][cout][cin][2][2] (phase = 2 a + b the output sub-pixel, tap (r, c) the source
 * pixel: the sums the entry above forms) -- inference rounds them with error feedback first (glare_filter_feedback_round_bf16). */
int glare_conv2d_pack_weight_upsample_phases(const float* w_phases, int cout, int cin_total, void* packed_bf16, glare_stream_t stream);
int glare_conv2d_bf16(const glare_conv_desc* desc_host, glare_stream_t stream);
/* Fused GroupNorm statistics: the conv epilogue leaves per-tile partial sums of its output; the reduce turns
 * them into the [B][1][32][2] (sum, sum of squares per group) block glare_groupnorm_apply_bf16 consumes, so the
 * consumer's statistics pass (one full read of the tensor) disappears. */
long long glare_conv2d_gn_partial_elems(int B, int OH, int OW, int Cout);
int glare_conv2d_gn_reduce(const float* gn_partial, float* stats_out, int B, int OH, int OW, int Cout,
                           glare_stream_t stream);
/* The same pair for desc.upsample == 2 (partials laid out on the SOURCE grid H x W, four phases per tile). */
long long glare_conv2d_upsample_gn_partial_elems(int B, int H, int W, int Cout);
int glare_conv2d_upsample_gn_reduce(const float* gn_partial, float* stats_out, int B, int H, int W, int Cout,
                                    glare_stream_t stream);

/* Thin convolutions whose INPUT has <= 4 channels (conv_in 3->128 / 3->512, cond_conv 3->64 +
 * sigmoid, color_conv 3->3, quant_conv / post_quant_conv 1x1 3->3: encoder_decoder.py:355,467;
 * ConditionEncoder.py:41-43; VQModel_arch.py:46-47).  fp32 input addressed by explicit element
 * strides (NCHW image or NHWC latent read in place), fp32 arithmetic, w OIHW fp32 (device).
 * out: NHWC [pixel][out_pitch], bf16 or fp32 (out_is_f32). */
int glare_conv2d_smallcin_f32(const float* x, long long stride_b, long long stride_c, long long stride_y,
                              long long stride_x, const float* w_oihw, const float* bias, void* out, int B, int H,
                              int W, int Cin, int Cout, int ksize, int out_pitch, int out_off, int act,
                              int out_is_f32, glare_stream_t stream);
/* ... with the output as a hi / lo pair (16-bit NHWC; Cout, out_pitch, out_off multiples of 8; see glare_conv_desc.out_lo). */
int glare_conv2d_smallcin_hilo_f32(const float* x, long long stride_b, long long stride_c, long long stride_y, long long stride_x,
                                   const float* w_oihw, const float* bias, void* out_hi, void* out_lo, int B, int H, int W, int Cin,
                                   int Cout, int ksize, int out_pitch, int out_off, int act, glare_stream_t stream);

/* ---- GroupNorm(32, C, eps) [+ swish] --------------------------------------------------------
 * Replaces Normalize() + nonlinearity() (encoder_decoder.py:29-35,119-120,126-127,170,436-437).
 * x: bf16 NHWC [B][HW][in_pitch] channels [in_off, in_off+C); y: bf16 NHWC [B][HW][C] dense.
 * C % 32 == 0, C <= 2048.  workspace: glare_groupnorm_workspace_bytes(B, HW) bytes of scratch. */
size_t glare_groupnorm_workspace_bytes(int B, long long HW);
/* apply only, with statistics already available as `splits` partial blocks [B][splits][32][2] */
/* out = a + b (bf16, dense [B][HW][C]) together with the GroupNorm statistics of `out`: stats receives
 * glare_groupnorm_workspace_bytes(B, HW) bytes = [B][splits][32][2] partial sums for glare_groupnorm_apply_bf16(..., stats,
 * splits) with splits = that size / (B * 256).  The residual add of AttnBlock (encoder_decoder.py:188: x + proj_out(h)) once
 * proj_out has been folded into v, fused with the statistics pass of the norm that consumes the block's output. */
int glare_add_groupnorm_stats_bf16(const void* a, const void* b, void* out, int B, long long HW, int C, void* stats,
                                   size_t stats_bytes, glare_stream_t stream);
/* The (a, d) pairs of y = act(a x + d) per (image, channel), fp32 [B][C][2], from the statistics block: for glare_conv_desc.gn_coef. */
int glare_groupnorm_coeffs_f32(const float* stats, int splits, int B, long long HW, int C, const float* gamma, const float* beta,
                               float eps, float* coef_out, glare_stream_t stream);
int glare_groupnorm_apply_bf16(const void* x, int in_pitch, int in_off, const float* gamma, const float* beta, void* y,
                               int B, long long HW, int C, float eps, int swish, const float* stats, int splits,
                               glare_stream_t stream);
/* GroupNorm of a hi / lo pair (value = x_hi + x_lo, same pitch / offset; see glare_conv_desc.out_lo): statistics and normalisation
 * from the 22-bit value, one rounding on the way out.  stats != NULL ([B][splits][32][2], from the producer's epilogue): apply only;
 * stats == NULL: the statistics pass runs first (workspace as glare_groupnorm_swish_bf16).
 * glare_split_hilo_f32: fp32 [n] (n % 8 == 0) -> hi = round16(v), lo = round16(v - hi): how conv_in's output opens the stream. */
int glare_groupnorm_hilo_bf16(const void* x_hi, const void* x_lo, int in_pitch, int in_off, const float* gamma, const float* beta,
                              void* y, int B, long long HW, int C, float eps, int swish, const float* stats, int splits,
                              void* workspace, size_t workspace_bytes, glare_stream_t stream);
/* ... with the output a hi / lo pair too (y_lo dense like y, or NULL = glare_groupnorm_hilo_bf16): the activation operand pair of an
 * fp32-class conv (glare_conv_desc.k_wrap). */
int glare_groupnorm_hilo_pair_bf16(const void* x_hi, const void* x_lo, int in_pitch, int in_off, const float* gamma, const float* beta,
                                   void* y, void* y_lo, int B, long long HW, int C, float eps, int swish, const float* stats, int splits,
                                   void* workspace, size_t workspace_bytes, glare_stream_t stream);
int glare_split_hilo_f32(const float* src, long long n, void* hi_bf16, void* lo_bf16, glare_stream_t stream);
int glare_groupnorm_swish_bf16(const void* x, int in_pitch, int in_off, const float* gamma, const float* beta,
                               void* y, int B, long long HW, int C, float eps, int swish, void* workspace,
                               size_t workspace_bytes, glare_stream_t stream);

/* ---- a8 glue: Mix and the mean rescale of MultiScaleDecoder2 --------------------------------
 * glare_mix_bf16: out = sigmoid(w)*a + (1-sigmoid(w))*b   (Mix.forward, deformableDecoder_arch.py:587-590)
 * glare_mean_rescale_bf16: out = h + xw * (mean(h)/mean(xw))  (deformableDecoder_arch.py:567); means per
 * sample (whole_batch_mean = 0, the build's inference contract) or over the whole batch
 * tensor as the reference does (whole_batch_mean = 1).  h, out: bf16 [B][n_per_sample]; xw: fp32. */
int glare_mix_bf16(const void* a, int a_pitch, int a_off, const void* b, int b_pitch, int b_off, void* out,
                   int out_pitch, int out_off, long long n_pixels, int C, float mix_w, glare_stream_t stream);
size_t glare_mean_rescale_workspace_bytes(int B, long long n_per_sample);
int glare_mean_rescale_bf16(const void* h, const float* xw, void* out, int B, long long n_per_sample,
                            int whole_batch_mean, void* workspace, size_t workspace_bytes, glare_stream_t stream);
/* The same rescale WITHOUT its own statistics pass (round 6): sum(h) from the kernel that produces h -- glare_mix_sum_bf16 = Mix.forward on
 * dense [B][n_per_sample] tensors + sum_partial[B * glare_mix_sum_blocks(n)] (per-block sums of the ROUNDED outputs; mix_w_dev != NULL: the
 * logit read on the device) --, sum(xw) from the DCN's epilogue (glare_mdcn_forward_nhwc_fused: per-tile sums of its fp32 outputs).
 * glare_mean_rescale_fused_bf16: ratio per image (or whole batch) in fp64 from the two partial sets, then out = h + xw * ratio with xw
 * fp32 or 16-bit (xw_is_16bit); ratio_scratch: fp32 [B]; tile_pixels = glare_mdcn_tile_pixels() of the DCN launch that wrote xw_tile_sums. */
int glare_mix_sum_blocks(long long n_per_sample);
int glare_mix_sum_bf16(const void* a, const void* b, void* out, int B, long long n_per_sample, float mix_w, const float* mix_w_dev_or_null,
                       float* sum_partial, glare_stream_t stream);
int glare_mean_rescale_fused_bf16(const void* h, const void* xw, int xw_is_16bit, void* out, int B, long long n_per_sample,
                                  long long pixels_per_sample, const float* h_sum_partial, const float* xw_tile_sums, int tile_pixels,
                                  int whole_batch_mean, float* ratio_scratch, glare_stream_t stream);

/* Layout conversion at the module boundary: the reference's tensors are NCHW fp32. */
int glare_nchw_to_nhwc(const float* src_nchw, void* dst_nhwc, int B, int C, long long HW, int dst_pitch, int dst_off,
                       int dst_is_bf16, glare_stream_t stream);
int glare_nhwc_to_nchw(const void* src_nhwc, float* dst_nchw, int B, int C, long long HW, int src_pitch, int src_off,
                       int src_is_bf16, glare_stream_t stream);

/* ---- a3: conditional flow, reverse direction, per coupling step ------------------------------
 * Replaces FlowStep.reverse_flow (FlowStep.py:100-119) = CondAffineSeparatedAndCond reverse
 * (FlowAffineCouplingsAblation.py:83-110) + InvertibleConv1x1 reverse (Permutations.py:45-59) +
 * ActNorm2d reverse (FlowActNorms.py:81-100).  z: fp32 token-major [B*H*W][3], updated in place.
 * glare_flow_h1_f32:  h1 = relu(ftA[:, off:off+64] + conv3x3(z[:,0] -> 64; wz[64][9]))  -> bf16 [pixel][64]
 * glare_flow_tail_f32: self-conditional affine from h4 [pixel][4], feature affine from
 *                      hF[pixel][pitch] (6 used at hF_off), then z = M z + t (host 3x3 / 3). */
int glare_flow_h1_f32(const float* z_nhwc3, const float* ftA, int ftA_pitch, int ftA_off, const float* wz_64x9,
                      void* h1_bf16, int B, int H, int W, glare_stream_t stream);
/* ... h1 as a hi / lo pair (two 16-bit [pixel][64] tensors, value = hi + lo): the operand of an fp32-class 1x1 conv (k_wrap). */
int glare_flow_h1_pair_f32(const float* z_nhwc3, const float* ftA, int ftA_pitch, int ftA_off, const float* wz_64x9,
                           void* h1_hi, void* h1_lo, int B, int H, int W, glare_stream_t stream);
int glare_flow_tail_f32(float* z_nhwc3, const float* h4, const float* hF, int hF_pitch, int hF_off,
                        long long n_pixels, const float* M_3x3_host, const float* t_3_host, float eps,
                        glare_stream_t stream);
/* The whole coupling step as ONE launch (round 6; csrc/flow_fused.hip): the three launches above and the two convs between them
 * (FlowAffineCouplingsAblation.py:143-151: 3x3 65 -> 64, ReLU, 1x1 64 -> 64, ReLU, Conv2dZeros 3x3 64 -> 4) fused per 8 x 32
 * pixel tile with the halo recomputed, every contraction in the fp32-class form (hi / lo operand pairs, three MFMA products),
 * h1 / h2 chained through registers.  z_in -> z_out, both fp32 [B*H*W][3], must be DIFFERENT buffers (a tile reads the
 * neighbouring tiles' z_in channel 0).  `image`: the step's filters as the kernel's fragment image, 16-B aligned,
 * glare_flow_step_fused_image_bytes() bytes, laid out as glare_amd.ops.flow_fused_image documents:
 *   [wz hi | wz lo] 2 x 2 KB, [W2 hi | W2 lo] 2 x 8 KB, [W4 hi | W4 lo] 2 x 8 KB of A fragments ([half][row][8] 16-bit each),
 *   then fp32 b2 in accumulator order [tile][half][16] and fp32 b4[4].
 * ftA / hF / M / t / eps as in glare_flow_h1_f32 / glare_flow_tail_f32. */
long long glare_flow_step_fused_image_bytes(void);
int glare_flow_step_fused_bf16(const float* z_in, float* z_out, const float* ftA, int ftA_pitch, int ftA_off, const void* image,
                               const float* hF, int hF_pitch, int hF_off, int B, int H, int W, const float* M_3x3_host,
                               const float* t_3_host, float eps, glare_stream_t stream);

/* ---- ActNorm data-dependent initialisation (a12: the first training forward of a fresh flow) ----------------------------
 * Replaces _ActNorm.initialize_parameters (FlowActNorms.py:32-46), reached from _ActNorm.forward (:82-83) for the 28 step
 * ActNorms and, through flow.Conv2d.forward (flow.py:48-52), the 96 coupling-net ActNorms:
 *   bias[c] = -mean_p x[p][c];  logs[c] = log(scale / (sqrt(mean_p (x[p][c] + bias[c])^2) + 1e-6))     over all B*H*W pixels.
 * x: fp32 [n_pixels][pitch], channels [off, off + C), C <= 64.  Two-pass, fp64 partials, fixed reduction order (deterministic).
 * glare_flow_h1_raw_f32: the PRE-activation of fAffine's first conv, fp32 [pixel][64] (glare_flow_h1_f32 without the relu / the
 *   16-bit rounding) -- what that conv's ActNorm is initialised from.
 * glare_flow_affine3_f32: z = M z + t in place, one coupling-free step (FlowStep.py:83-88). */
size_t glare_actnorm_init_workspace_bytes(long long n_pixels);
int glare_actnorm_init_f32(const float* x, int pitch, int off, int C, long long n_pixels, float scale, float* bias_out,
                           float* logs_out, void* workspace, size_t workspace_bytes, glare_stream_t stream);
int glare_flow_h1_raw_f32(const float* z_nhwc3, const float* ftA, int ftA_pitch, int ftA_off, const float* wz_64x9,
                          float* raw_f32, int B, int H, int W, glare_stream_t stream);
int glare_flow_affine3_f32(float* z_nhwc3, long long n_pixels, const float* M_3x3_host, const float* t_3_host,
                           glare_stream_t stream);

/* ---- 1x1 convolution, weight-stationary form (csrc/conv1x1.hip) ----------------------------------------------------------
 * The 1x1 nn.Conv2d's with Cin in {128, 256, 512} and Cout % 128 == 0 (Cout / 128 a divisor of 32): AttnBlock's query / output
 * projections (encoder_decoder.py:146-165) and ResnetBlock.nin_shortcut (:104-115).  Same arithmetic as glare_conv2d_bf16 with
 * ksize 1 (bf16 operands, fp32 accumulation, bias, residual add, activation, bf16 NHWC output, optional GroupNorm partial sums
 * of the rounded output), different schedule: one persistent workgroup per CU keeps its 128-cout slice of the filter in LDS and
 * streams 32-pixel row blocks through it.  x: bf16 [B][pixels][x_pitch] (channels x_off .. x_off + Cin); w_bf16: [Cout][Cin]
 * from glare_conv1x1_ws_pack_weight; gn_partial: glare_conv1x1_ws_gn_partial_elems floats (NULL = none), turned into the
 * [B][1][32][2] statistics block of glare_groupnorm_apply_bf16 by glare_conv1x1_ws_gn_reduce. */
int glare_conv1x1_ws_supported(int Cin, int Cout);
int glare_conv1x1_ws_pack_weight(const float* w_oihw, int cout, int cin, void* w_bf16, glare_stream_t stream);
long long glare_conv1x1_ws_gn_partial_elems(int B, long long pixels_per_image, int Cout);
int glare_conv1x1_ws_gn_reduce(const float* gn_partial, float* stats_out, int B, long long pixels_per_image, int Cout,
                               glare_stream_t stream);
int glare_conv1x1_ws_bf16(const void* x, int x_pitch, int x_off, const void* w_bf16, const float* bias, const void* residual,
                          int res_pitch, int res_off, void* out, int out_pitch, int out_off, int B, long long pixels_per_image, int Cin,
                          int Cout, int act, float* gn_partial, glare_stream_t stream);
/* The same with ONE FILTER PER IMAGE (w_bf16 [B][Cout][Cin], bias [B][bias_image_stride]; strides in elements): a GroupNorm without
 * activation in front of a 1x1 conv is a per-(image, channel) affine map of the conv's input, i.e. a per-image filter, and the
 * normalised tensor need not exist.  (8 * 32 / (Cout / 128)) % B == 0.  glare_attn_fold_groupnorm_f32 builds the two per-image
 * filters of AttnBlock (norm -> q / k / v 1x1 -> bmm softmax bmm -> proj_out, encoder_decoder.py:146-188) from the tensor's GroupNorm
 * statistics block [B][splits][32][2] and the folded projections wq = s Wk^T Wq, bq = s Wk^T bq_ref, wo = Wp Wv, bo = Wp bv + bp
 * (fp32 [C][C] / [C]): with a = rstd gamma, d = beta - mean a:  wq_out = diag(a) wq diag(a), bq_out = a o (wq d + bq),
 * wo_out = wo diag(a), bo_out = wo d + bo;  attention then runs with keys = values = the RAW tensor. */
int glare_conv1x1_ws_image_bf16(const void* x, int x_pitch, int x_off, const void* w_bf16, long long w_image_stride, const float* bias,
                                int bias_image_stride, const void* residual, int res_pitch, int res_off, void* out, int out_pitch,
                                int out_off, int B, long long pixels_per_image, int Cin, int Cout, int act, float* gn_partial,
                                glare_stream_t stream);
/* hi / lo form of the two entry points above (w_image_stride = 0: one filter for all images; see glare_conv_desc.out_lo): `out`
 * receives hi = round16(v), `out_lo` lo = round16(v - hi) of v = acc + bias + residual + residual_lo (residual_lo optional),
 * nothing rounded before the residual add; the fused GroupNorm statistics are those of v. */
int glare_conv1x1_ws_hilo_bf16(const void* x, int x_pitch, int x_off, const void* w_bf16, long long w_image_stride,
                               const float* bias, int bias_image_stride, const void* residual, const void* residual_lo,
                               int res_pitch, int res_off, void* out, void* out_lo, int out_pitch, int out_off, int B,
                               long long pixels_per_image, int Cin, int Cout, int act, float* gn_partial, glare_stream_t stream);

/* The fp32-class form on the weight-stationary kernel (round 4; glare_conv_desc.k_wrap for the path's 1x1 convs with Cin in {128, 256,
 * 512} and Cout a multiple of 64 dividing 2048: the conditional encoder's nin_shortcuts and its AttnBlocks' folded query / output
 * projections, encoder_decoder.py:104-115,146-165 -- nn.Conv2d in fp32 there).  Activation and filter are hi / lo pairs of 16-bit
 * tensors (x_lo: same pitch / offset as x_hi; w_hi / w_lo: [Cout][Cin] from glare_conv1x1_ws_pack_weight on the two halves of the
 * fp32 filter); out = x_hi.w_hi + x_lo.w_hi + x_hi.w_lo + bias + residual (+ residual_lo) in fp32, then act; written as the pair
 * (out, out_lo) or, with out_lo == NULL, as its 16-bit rounding alone.  gn_partial as in glare_conv1x1_ws_bf16 (Cout % 128 == 0). */
int glare_conv1x1_ws_split_supported(int Cin, int Cout);
int glare_conv1x1_ws_split_bf16(const void* x_hi, const void* x_lo, int x_pitch, int x_off, const void* w_hi, const void* w_lo,
                                const float* bias, const void* residual, const void* residual_lo, int res_pitch, int res_off, void* out,
                                void* out_lo, int out_pitch, int out_off, int B, long long pixels_per_image, int Cin, int Cout, int act,
                                float* gn_partial, glare_stream_t stream);
int glare_attn_fold_groupnorm_f32(const float* stats, int splits, int B, long long HW, int C, const float* gamma, const float* beta,
                                  float eps, const float* wq, const float* bq, const float* wo, const float* bo, void* wq_out,
                                  float* bq_out, void* wo_out, float* bo_out, int feedback, glare_stream_t stream);
/* feedback != 0 (round 6): the two 16-bit filters rounded with error feedback along the input channels of every output row, as
 * glare_filter_feedback_round_bf16 rounds the static filters (0: round-to-nearest per weight). */

/* ---- a2: blockwise spatial self-attention, one head, d = 512 ---------------------------------
 * Replaces the bmm / softmax / bmm of AttnBlock.forward (encoder_decoder.py:176-188) without the
 * [B, N, N] score tensor.   out[b, i, :] = sum_j softmax_j(q_i . k_j) v_j
 * q, k : bf16 [B][N][ldq|ldk] row-major (d contiguous); the caller folds 512^-0.5 * log2(e) into q
 *        (the kernel exponentiates with exp2).
 * v_t  : bf16 [B][512][v_pitch], V TRANSPOSED (token index contiguous), v_pitch >= roundup(N, 32),
 *        multiple of 8, pad columns zero (written so by glare_conv2d_bf16 with GLARE_OUT_PLANAR_BF16).
 * out  : bf16 [B][N][ldo]. */
int glare_attention_d512_bf16(const void* q, int ldq, const void* k, int ldk, const void* v_t, long long v_pitch,
                              void* out, int ldo, int B, int N, glare_stream_t stream);
/* The same product with the KEYS split over `key_splits` workgroups per query block (flash-decoding style), for batches too
 * small to fill 256 CUs with 128-row query blocks (one 400x600 image = 128 blocks): each split leaves an un-normalised fp32
 * partial + (max, sum) in `workspace`, a second kernel merges them.  key_splits == 1 is glare_attention_d512_bf16. */
size_t glare_attention_d512_splitk_workspace_bytes(int B, int N, int key_splits);
int glare_attention_d512_splitk_bf16(const void* q, int ldq, const void* k, int ldk, const void* v_t, long long v_pitch, void* out,
                                     int ldo, int B, int N, int key_splits, void* workspace, size_t workspace_bytes,
                                     glare_stream_t stream);

/* The forward above (key_splits >= 1) that also leaves lse[b][i] = log2 sum_j 2^(q_i . k_j), fp32 [B][N], and the fused backward of
 * AttnBlock's two torch.bmm + softmax that consumes it (reference: autograd of encoder_decoder.py:176-188 behind loss.backward(),
 * LLFlow_model.py:231-236): no N x N tensor; q (pre-scaled: q.k are base-2 logits), k, v, o, d_o and the three gradients are dense
 * bf16 [B][N][512]; ds = ln2_scale * p * (dp - do.o); workspace of glare_attention_d512_backward_workspace_bytes (B*N floats). */
int glare_attention_d512_lse_bf16(const void* q, int ldq, const void* k, int ldk, const void* v_t, long long v_pitch, void* out, int ldo,
                                  float* lse, int B, int N, int key_splits, void* workspace, size_t workspace_bytes,
                                  glare_stream_t stream);
size_t glare_attention_d512_backward_workspace_bytes(int B, int N);
int glare_attention_d512_backward_bf16(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse,
                                       void* dq, void* dk, void* dv, int B, int N, float ln2_scale, void* workspace,
                                       size_t workspace_bytes, glare_stream_t stream);
/* The attention of AttnBlock with keys and values SHARED: out[b, i, :] = sum_j softmax_j(q_i . x_j) x_j.
 * AttnBlock is single-head self-attention on one tensor h = GroupNorm(x) (encoder_decoder.py:168-188): every term of
 * (Wq h_i + bq).(Wk h_j + bk) that does not depend on j cancels in softmax_j, so the key projection folds into the query
 * (q'_i = Wk^T Wq h_i + Wk^T bq, times 512^-0.5 * log2(e)), and since softmax rows sum to 1 the value projection commutes with
 * the weighted average and folds into proj_out: the N^2 part runs on x = h for BOTH operands -- one tile stream instead of two.
 * q : bf16 [B][N][ldq];  kv : bf16 [B][N][ldkv] (d = 512 contiguous);  out : bf16 [B][N][ldo].
 * key_splits > 1: keys split over workgroups as above (workspace of glare_attention_d512_splitk_workspace_bytes); 1: none needed. */
int glare_attention_kv512_bf16(const void* q, int ldq, const void* kv, int ldkv, void* out, int ldo, int B, int N, int key_splits,
                               void* workspace, size_t workspace_bytes, glare_stream_t stream);
/* ... with the output as a hi / lo pair (out_lo: same shape / ldo, value = out + out_lo; NULL = the call above): the fp32
 * accumulators leave with 22 bits, as the operand pair of an fp32-class output projection (glare_conv_desc.k_wrap). */
int glare_attention_kv512_pair_bf16(const void* q, int ldq, const void* kv, int ldkv, void* out, void* out_lo, int ldo, int B, int N,
                                    int key_splits, void* workspace, size_t workspace_bytes, glare_stream_t stream);

/* ---- a9: modulated deformable convolution (DCNv2), forward -----------------------------------
 * glare_mdcn_forward_f32 is the drop-in for the pybind function
 *   deform_conv_ext.modulated_deform_conv_forward(input, weight, bias, ones, offset, mask, output,
 *       columns, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w,
 *       group, deformable_group, with_bias)            (deform_conv_ext.cpp:107-124,158-160)
 * called from ModulatedDeformConvFunction.forward (deform_conv.py:149-152): same tensors in the
 * reference's layouts (NCHW fp32; offset [B][dg*2*kh*kw][Ho][Wo], channel g*2K+2k = dh, +1 = dw;
 * mask [B][dg*kh*kw][Ho][Wo]; weight [Co][C/groups][kh][kw]), caller-allocated `out`
 * (deform_conv.py:147).  The reference's callee-owned `columns`/`ones` scratch is replaced by a
 * caller-owned workspace of glare_mdcn_workspace_bytes(); bias_or_null == NULL is with_bias=False.
 * Any shape the reference accepts (deform_conv_cuda.cpp:497-516: C % group == 0, Co % group == 0, C % deformable_group == 0):
 * groups == 1, C/dg in {32, 64}, Co % 64 == 0, Co <= 256 (the GLARE warps: C = Co = 256 and 128, dg = 4) run on the MFMA kernels
 * (they need the workspace); every other configuration runs on general fp32 kernels (csrc/dcn_generic.hip: any groups /
 * deformable groups / channels / kernel / stride / padding / dilation; the workspace may then be NULL).
 *
 * glare_mdcn_forward_nhwc is the same operator on the pipeline's native layouts: x NHWC (fp32 or
 * bf16) with pitch/offset, offset/mask planar with explicit plane pitches and per-sample strides in
 * elements (0 = dense; as written by glare_conv2d_bf16 in GLARE_OUT_PLANAR_F32 mode into ONE buffer), mask optionally still a logit (sigmoid fused,
 * DCNv2Pack.forward deformableDecoder_arch.py:148), weights pre-packed once by
 * glare_mdcn_pack_weight_f32 (fp32 [Co][C][kh][kw] -> split-bf16 fragment image of the SAME byte size,
 * `packed` holds Co*C*kh*kw floats' worth of bytes); out NHWC fp32 or planar.
 * Arithmetic: operands carried as bf16 hi + lo pairs, 3 bf16 MFMAs per product term, fp32 accumulation
 * (>= 16 mantissa bits per operand; <= ~1e-5 relative vs fp32). */
size_t glare_mdcn_workspace_bytes(int B, int C, int H, int W, int Co, int kh, int kw);
int glare_mdcn_forward_f32(const float* x, const float* offset, const float* mask, const float* weight,
                           const float* bias_or_null, float* out, int B, int C, int H, int W, int Co, int kh, int kw,
                           int sh, int sw, int ph, int pw, int dh, int dw, int groups, int dg, void* workspace,
                           size_t workspace_bytes, glare_stream_t stream);
int glare_mdcn_pack_weight_f32(const float* weight_oihw, float* packed, int Co, int C, int kh, int kw, int dg,
                               glare_stream_t stream);
int glare_mdcn_forward_nhwc(const void* x, int x_is_bf16, int x_pitch, int x_off, const float* offset,
                            long long offset_plane, long long offset_batch_stride, const float* mask,
                            long long mask_plane, long long mask_batch_stride, int mask_is_logit,
                            const float* weight_packed, const float* bias, float* out, int out_planar, int out_pitch,
                            int out_off, long long out_plane, int B, int C, int H, int W, int Co, int kh, int kw, int sh,
                            int sw, int ph, int pw, int dh, int dw, int groups, int dg, int flags, glare_stream_t stream);
/* The pipeline's form (round 6; the lean kernel only: 16-bit x, 3 | kh kw, every tensor < 2 GB -- GLARE_ERR_UNSUPPORTED
 * otherwise, the caller then uses the call above): the output as 16-bit NHWC (out16 != NULL: rounded once from the fp32 accumulator +
 * bias; else fp32 out32) and / or tile_sums [B][ceil(Ho Wo / glare_mdcn_tile_pixels(C, Co, dg, flags))] = per pixel tile the sum of
 * its fp32 outputs -- the tiles are then cut PER IMAGE (the last one ragged), so an image's sums are the same in any batch: mean(x_w)
 * of the rescale that follows the warp (deformableDecoder_arch.py:567) without another pass.  flags: 0 or GLARE_MDCN_SINGLE_PASS. */
int glare_mdcn_tile_pixels(int C, int Co, int dg, int flags);
int glare_mdcn_forward_nhwc_fused(const void* x, int x_pitch, int x_off, const float* offset, long long offset_plane,
                                  long long offset_batch_stride, const float* mask, long long mask_plane, long long mask_batch_stride,
                                  int mask_is_logit, const float* weight_packed, const float* bias, float* out32_or_null, void* out16_or_null,
                                  int out_pitch, int out_off, float* tile_sums_or_null, int B, int C, int H, int W, int Co, int kh, int kw,
                                  int sh, int sw, int ph, int pw, int dh, int dw, int groups, int dg, int flags, glare_stream_t stream);
/* glare_mdcn_forward_nhwc picks between two MFMA kernels with the same arithmetic: the general-extent one (fp32 or bf16 x) and a
 * leaner one for bf16 x when every tensor is < 2 GB and kh*kw % 3 == 0.  `flags` & GLARE_MDCN_GENERAL_KERNEL pins the former for
 * this call (a per-call argument: the library keeps no state; tests compare the two kernels on one input with it). */
#define GLARE_MDCN_GENERAL_KERNEL 1
/* `flags` & GLARE_MDCN_SINGLE_PASS (round 3): the blended, modulated sample and the filter are each rounded ONCE to the library's
 * 16-bit activation format (bf16 in libglare_hip.so, IEEE half in libglare_hip_f16.so) and contracted by one MFMA per product with
 * fp32 accumulation -- the arithmetic of every other convolution on the path -- instead of the split form (two 16-bit halves per
 * operand, three MFMAs per product, fp32-class).  `weight_packed` must then come from glare_mdcn_pack_weight_single_f32 (half the
 * bytes: Co*C*kh*kw 16-bit values).  Only where the leaner kernel applies (16-bit x, 3 | kh*kw, tensors < 2 GB), else UNSUPPORTED. */
#define GLARE_MDCN_SINGLE_PASS 2
int glare_mdcn_pack_weight_single_f32(const float* weight_oihw, void* packed, int Co, int C, int kh, int kw, int dg,
                                      glare_stream_t stream);

/* ---- a10: modulated deformable convolution (DCNv2), backward ----------------------------------
 * Drop-in for the pybind function
 *   deform_conv_ext.modulated_deform_conv_backward(input, weight, bias, ones, offset, mask, columns,
 *       grad_input, grad_weight, grad_bias, grad_offset, grad_mask, grad_output, kernel_h, kernel_w,
 *       stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, group, deformable_group, with_bias)
 *   (deform_conv_ext.cpp:127-147,161-163), called from ModulatedDeformConvFunction.backward
 * (deform_conv.py:166-171): reference layouts (NCHW fp32), all gradient buffers caller-allocated and
 * caller-ZEROED (deform_conv.py:161-165); grad_weight / grad_bias are accumulated into, grad_input /
 * grad_offset / grad_mask are overwritten.  grad_input may be NULL (skipped: the GLARE warp input needs
 * no gradient, VQLLFLOWDeformable_arch.py:240-248); grad_bias_or_null NULL = with_bias False.
 * grad_input is summed with fp32 atomics (order-nondeterministic), like the reference (kernel.cu:688).
 * Shapes as the forward: the MFMA kernels additionally need Co % 128 == 0; everything else runs on the general fp32 kernels
 * (grad_offset / grad_mask / grad_weight deterministic there, grad_input by atomics). */
size_t glare_mdcn_backward_workspace_bytes(int B, int C, int H, int W, int Co, int kh, int kw);
int glare_mdcn_backward_f32(const float* x, const float* offset, const float* mask, const float* weight,
                            const float* grad_out, float* grad_input, float* grad_offset, float* grad_mask,
                            float* grad_weight, float* grad_bias_or_null, int B, int C, int H, int W, int Co, int kh,
                            int kw, int sh, int sw, int ph, int pw, int dh, int dw, int groups, int dg, void* workspace,
                            size_t workspace_bytes, glare_stream_t stream);

/* ---- a4: conditional flow, normal (training) direction + negative log-likelihood -------------------
 * Replaces FlowStep.normal_flow (FlowStep.py:75-98), CondAffineSeparatedAndCond forward
 * (FlowAffineCouplingsAblation.py:51-81), get_logdet (:121-122) and GaussianDiag.logp (flow.py:76-95).
 * Per coupling step: glare_flow_fwd_pre_f32 (z = M z + t, then the feature affine), glare_flow_h1_f32 and the
 * two convs as in the reverse direction, glare_flow_fwd_post_f32 (self-conditional affine on z[1:]).
 * Each call writes B * glare_flow_blocks_per_sample() fp32 partial sums of log(scale) into one row of
 * logdet_partial; glare_flow_nll_reduce_f32 sums all rows per sample and evaluates the Gaussian term:
 *   out[2b] = sum of the data-dependent log-determinant, out[2b+1] = sum_p -0.5((z-mean)^2 + log 2pi).
 * The data-independent part (actnorm logs, slogdet of the 1x1 kernels, times pixels) is the caller's (fp64). */
int glare_flow_blocks_per_sample(long long pixels_per_sample);
int glare_flow_fwd_pre_f32(float* z_nhwc3, const float* hF, int hF_pitch, int hF_off, int B, long long pixels_per_sample,
                           const float* M_3x3_host, const float* t_3_host, float eps, float* logdet_partial,
                           glare_stream_t stream);
/* same, with the 3x3 matrix (9 floats, row-major) followed by the offset (3 floats) read from DEVICE memory: no host round trip
 * inside the training step (the launch sequence stays capturable in a hipGraph) */
int glare_flow_fwd_pre_dev_f32(float* z_nhwc3, const float* hF, int hF_pitch, int hF_off, int B, long long pixels_per_sample,
                               const float* Mt_12_device, float eps, float* logdet_partial, glare_stream_t stream);
int glare_flow_fwd_post_f32(float* z_nhwc3, const float* h4, int B, long long pixels_per_sample, float eps,
                            float* logdet_partial, glare_stream_t stream);
/* out-of-place forms (z_in -> z_out; z_in == z_out is the in-place call): the training forward keeps every step's input and
 * mid-step latent for the backward, so each kernel writes straight into the step-major buffer instead of being followed by a copy */
int glare_flow_fwd_pre_dev_io_f32(const float* z_in, float* z_out, const float* hF, int hF_pitch, int hF_off, int B,
                                  long long pixels_per_sample, const float* Mt_12_device, float eps, float* logdet_partial,
                                  glare_stream_t stream);
int glare_flow_fwd_post_io_f32(const float* z_in, float* z_out, const float* h4, int B, long long pixels_per_sample, float eps,
                               float* logdet_partial, glare_stream_t stream);
int glare_flow_nll_reduce_f32(const float* z_nhwc3, const float* mean_nhwc3, const float* logdet_partial,
                              int n_partial_rows, int B, long long pixels_per_sample, double* out_2_per_sample,
                              glare_stream_t stream);

/* ---- training step: contractions over the pixel / token axis (rows a12, a13) --------------------------------
 * Batched NT GEMM on bf16 MFMA, fp32 accumulate:  C[b][m][n] = alpha * sum_k A[b][m][k] * B[b][n][k]  (+ C[b][m][n]
 * when accumulate != 0).  A, B bf16 with K contiguous (lda, ldb, strides in ELEMENTS, multiples of 8; K % 32 == 0;
 * rows beyond M / N are not read); C fp32 or bf16 (out_bf16).  Split-K = a batch whose strideA/strideB is the K slice
 * and whose C is a partial per slice, summed by glare_reduce_parts_f32.  Replaces the cuDNN weight-gradient and the
 * autograd of the two torch.bmm in AttnBlock (encoder_decoder.py:176-188) behind `loss.backward()`
 * (LLFlow_model.py:231-236, VQLLFLOWD_model.py:226-229). */
int glare_gemm_nt_bf16(const void* A, const void* B, void* C, int M, int N, int K, long long lda, long long ldb,
                       long long ldc, int batch, long long strideA, long long strideB, long long strideC, float alpha,
                       int out_bf16, int accumulate, glare_stream_t stream);
/* out[i] = scale * sum_s parts[s][i] (+ out[i]) -- deterministic second level of split reductions */
int glare_reduce_parts_f32(const float* parts, int n_parts, long long n, float scale, float* out, int accumulate,
                           glare_stream_t stream);
/* the same for n_groups independent sums: out[grp][i] = scale * sum_s parts[grp][s][i] */
int glare_reduce_parts_grouped_f32(const float* parts, int n_groups, int n_parts, long long n, float scale, float* out, int accumulate,
                                   glare_stream_t stream);

/* K-contiguous operand builders for glare_gemm_nt_bf16.
 * glare_im2col_t_bf16: colT[row_base + c*k*k + tap][p] = x[b, oy*stride+ty-pad, ox*stride+tx-pad, c] (0 outside), p the
 *   flattened output pixel (b, oy, ox); x bf16 NHWC [B][H][W][pitch], channels [off, off+Ci), optionally seen through
 *   the nearest x2 upsample; stride 2 uses Downsample's (0,1,0,1) padding (encoder_decoder.py:71-74, pass pad = 0).
 *   Columns [P, ldp) are zero-filled (ldp % 64 == 0); ones_row >= 0 additionally writes a row of ones (p < P), which
 *   turns the weight-gradient GEMM's extra column into the bias gradient.  The weight gradient of a k x k conv is then
 *   dW[co][ci*k*k + tap] = gemm_nt(gO^T [Co][P], colT) -- the layout of an OIHW filter (cuDNN wgrad in the reference).
 * glare_im2col_t_f32: the same matrix from an fp32 tensor addressed by element strides (stride 1, symmetric pad).
 * glare_transpose_bf16: out[b][c][r] = in[b][r][c], columns [rows, ld_out) zero-filled (ld_out % 64 == 0). */
int glare_im2col_t_bf16(const void* x_nhwc, int B, int H, int W, int pitch, int off, int Ci, int ksize, int stride, int pad,
                        int upsample, void* colT, long long ldp, int row_base, int ones_row, glare_stream_t stream);
int glare_im2col_t_f32(const float* x, long long stride_b, long long stride_c, long long stride_y, long long stride_x, int B,
                       int H, int W, int Ci, int ksize, int pad, void* colT, long long ldp, int row_base, int ones_row,
                       glare_stream_t stream);
/* Weight + bias gradients of `groups` independent ksize x ksize (3: pad 1; or 1), stride-1 convolutions of one shape, straight from
 * the NHWC operands (csrc/wgrad.hip; cuDNN wgrad in the reference's loss.backward(), LLFlow_model.py:231-236):
 * dWt[grp][(ty*ksize+tx)*Ci + ci][co] = sum_{b,y,x} g[b,y,x,co] x[b,y+ty-pad,x+tx-pad,ci], row ksize^2*Ci = the bias gradient; fp32
 * [groups][ksize^2*Ci + 1][Co].  Group grp reads x + grp*x_gstride (bf16 NHWC [B][H][W][xpitch], the Ci channels at xoff) and
 * g + grp*g_gstride (bf16 NHWC [B][H][W][gpitch], the Co channels at 0), strides in elements: B*H*W*pitch walks step-major tensors,
 * the channel count walks channel blocks of one tensor.  Ci, Co, xoff, pitches and strides multiples of 8, one image below 2 GB.
 * Split over pixel ranges with fp32 partials in `workspace` (glare_conv_wgrad_workspace_bytes), summed in a fixed order. */
size_t glare_conv_wgrad_workspace_bytes(int ksize, int groups, int B, int H, int W, int Ci, int Co);
int glare_conv_wgrad_bf16(int ksize, const void* x, int xpitch, int xoff, long long x_gstride, const void* g, int gpitch,
                          long long g_gstride, float* dWt, int groups, int B, int H, int W, int Ci, int Co, void* workspace,
                          size_t workspace_bytes, glare_stream_t stream);
/* ... of ONE conv with the result laid out like the filter itself (round 6): dW_oihw fp32 [Co][ci_total][ksize][ksize], this launch
 * filling the input channels [ci_off, ci_off + Ci) (torch.cat((x, x2), 1) as the conv's input: one launch per source), db fp32 [Co] or
 * NULL.  A gradient laid out like its parameter is taken over by autograd as it is (no per-parameter copy). */
size_t glare_conv_wgrad_oihw_workspace_bytes(int ksize, int B, int H, int W, int Ci, int Co);
int glare_conv_wgrad_oihw_bf16(int ksize, const void* x, int xpitch, int xoff, const void* g, int gpitch, float* dW_oihw, int ci_total,
                               int ci_off, float* db_or_null, int B, int H, int W, int Ci, int Co, void* workspace, size_t workspace_bytes,
                               glare_stream_t stream);
/* out[c] = sum_p g[p][c] (bias gradient); g bf16 [P][pitch]; workspace >= 256 * C floats */
int glare_colsum_bf16(const void* g, int pitch, long long P, int C, float* out, void* workspace, size_t workspace_bytes,
                      glare_stream_t stream);
int glare_transpose_bf16(const void* in, long long ld_in, long long batch_stride_in, void* out, long long ld_out,
                         long long batch_stride_out, long long rows, int cols, int batch, glare_stream_t stream);

/* Data gradients are stride-1 convolutions (glare_conv2d_bf16 with the flipped, transposed filter) after:
 * glare_dilate2_bf16: out[B][2OH][2OW][C], out[b][2oy+1][2ox+1] = g[b][oy][ox], 0 elsewhere (Downsample backward);
 * glare_pool2_sum_bf16: out[B][H][W][C] = 2x2 block sums of g[B][2H][2W][C] (Upsample backward, encoder_decoder.py:50).
 * glare_act_backward: g *= act'(y) in place for the activation the forward conv fused (GLARE_ACT_RELU / _SIGMOID). */
int glare_dilate2_bf16(const void* g, void* out, int B, int OH, int OW, int C, glare_stream_t stream);
int glare_pool2_sum_bf16(const void* g, void* out, int B, int H, int W, int C, glare_stream_t stream);
int glare_act_backward(void* g, int g_is_f32, int g_pitch, int g_off, const void* y, int y_is_f32, int y_pitch, int y_off,
                       long long pixels, int C, int act, glare_stream_t stream);
/* out = a + b (+ c), bf16, n % 8 == 0: gradient accumulation where an activation feeds several consumers (the residual
 * branch of ResnetBlock / AttnBlock, the q / k / v projections) -- done here rather than by the autograd engine's add */
int glare_add_bf16(const void* a, const void* b, const void* c_or_null, void* out, long long n, glare_stream_t stream);
int glare_cast_f32_bf16(const float* in, int in_pitch, int in_off, void* out, int out_pitch, int out_off, long long pixels,
                        int C, glare_stream_t stream);
int glare_cast_bf16_f32(const void* in, int in_pitch, int in_off, float* out, int out_pitch, int out_off, long long pixels,
                        int C, glare_stream_t stream);

/* Backward of glare_groupnorm_swish_bf16 (autograd of Normalize()+nonlinearity(), encoder_decoder.py:29-35).
 * stats: the forward's statistics block [B][stat_splits][32][2]; dy, dx dense bf16 NHWC [B][HW][C];
 * dgamma_dbeta_per_image: fp32 [B][2][C] = per-image (dbeta, dgamma) contributions, summed over B by the caller
 * (glare_reduce_parts_f32) so the result is deterministic. */
size_t glare_groupnorm_backward_workspace_bytes(int B, long long HW, int C);
int glare_groupnorm_swish_backward_bf16(const void* x, int in_pitch, int in_off, const void* dy, const float* stats,
                                        int stat_splits, const float* gamma, const float* beta, void* dx,
                                        float* dgamma_dbeta_per_image, int B, long long HW, int C, float eps, int swish,
                                        void* workspace, size_t workspace_bytes, glare_stream_t stream);

/* Attention backward in materialised form (training crops: N = 6400 / 4096 tokens), all contractions through
 * glare_gemm_nt_bf16: P = softmax2(q' k^T) recomputed by glare_softmax2_rows_f32 (base-2 logits, as the forward
 * kernel), dP = dO v^T, dS = scale * P o (dP - rowsum(dO o O)) by glare_attention_ds_bf16, then dq' = dS k,
 * dk = dS^T q', dv = P^T dO. */
int glare_softmax2_rows_f32(const float* S, long long lds, void* P_bf16, long long ldp, long long rows, int n,
                            glare_stream_t stream);
int glare_attention_ds_bf16(const void* P, long long ldp, const float* dP, long long lddp, const void* dO, int ld_do,
                            const void* O, int ld_o, int d, void* dS, long long ldds, long long rows, int n, float scale,
                            glare_stream_t stream);

/* torch.optim.Adam step (LLFlow_model.py:110-118, VQLLFLOWD_model.py:112-119) on flat fp32 buffers; `step` is the
 * 1-based step count, grad_scale multiplies the gradient first (1/world for the data-parallel mean). */
int glare_adam_step_f32(float* w, const float* grad, float* exp_avg, float* exp_avg_sq, long long n, float lr, float beta1,
                        float beta2, float eps, float weight_decay, int step, float grad_scale, glare_stream_t stream);
/* GradScaler.step / .update (LLFlow_model.py:236-241, VQLLFLOWD_model.py:226-228): a step whose gradients hold an inf or a
 * NaN is SKIPPED -- no moment update, no weight decay, no step count -- and the loss scale backs off.
 * glare_grad_nonfinite_f32 ORs "some grad[i] is not finite" into *found_device (int32; the caller zeroes it before the first
 *   buffer of a step; OR is order-independent, so the flag is deterministic).
 * glare_adam_prepare_guarded / glare_adam_step_dev_guarded_f32 are the device-state Adam entry points below made no-ops when
 *   *skip_if_nonzero_device != 0 (the whole step stays one capturable launch sequence: no host read of the flag).
 * glare_gradscaler_update: GradScaler.update on device state -- found: scale *= backoff, tracker = 0; else tracker += 1 and,
 *   at growth_interval, scale *= growth and tracker = 0.  (bf16 activations keep fp32's range: the scale is bookkeeping for
 *   `.state` file parity, it multiplies nothing.) */
int glare_grad_nonfinite_f32(const float* grad, long long n, int* found_device, glare_stream_t stream);
int glare_adam_prepare_guarded(int* step_device, float* state3_device, float beta1, float beta2, const int* skip_if_nonzero_device,
                               glare_stream_t stream);
int glare_adam_step_dev_guarded_f32(float* w, const float* grad, float* exp_avg, float* exp_avg_sq, long long n, float lr, float beta1,
                                    float beta2, float eps, float weight_decay, const float* state3_device, float grad_scale,
                                    const int* skip_if_nonzero_device, glare_stream_t stream);
/* ... under the fp16 precision (round 4: the training kernels exist in libglare_hip_f16.so too) the scale is REAL, as in the
 * reference: the caller multiplies the loss by *loss_scale_device before backward (`scaler.scale(loss).backward()`), 16-bit
 * activation gradients carry it, and this entry point divides it out of the fp32 gradient (`scaler.unscale_`) on the fly. */
int glare_adam_step_dev_scaled_f32(float* w, const float* grad, float* exp_avg, float* exp_avg_sq, long long n, float lr, float beta1,
                                   float beta2, float eps, float weight_decay, const float* state3_device, float grad_scale,
                                   const float* loss_scale_device, const int* skip_if_nonzero_device, glare_stream_t stream);
int glare_gradscaler_update(float* scale_device, int* growth_tracker_device, const int* found_device, float growth_factor,
                            float backoff_factor, int growth_interval, glare_stream_t stream);
/* The same step with ALL optimizer state on the device, so that a whole training step replays from a hipGraph:
 * glare_adam_prepare increments *step_device and writes state3 = {1 - beta1^t, sqrt(1 - beta2^t), (unchanged) lr multiplier};
 * glare_adam_step_dev_f32 reads the bias corrections and the lr multiplier from state3 (lr_effective = lr * state3[2]). */
int glare_adam_prepare(int* step_device, float* state3_device, float beta1, float beta2, glare_stream_t stream);
int glare_adam_step_dev_f32(float* w, const float* grad, float* exp_avg, float* exp_avg_sq, long long n, float lr, float beta1,
                            float beta2, float eps, float weight_decay, const float* state3_device, float grad_scale,
                            glare_stream_t stream);

/* Backward of the flow's normal direction (adjoint of glare_flow_fwd_pre / _h1 / _fwd_post / _nll_reduce): autograd
 * through FlowStep.normal_flow (FlowStep.py:75-98) and GaussianDiag.logp (flow.py:76-95).  gz is the latent's gradient,
 * fp32 [pixel][3], updated in place from the last coupling step to the first; g_logdet / g_logp are the per-sample
 * gradients of the two sums glare_flow_nll_reduce_f32 returns.  Partials are [glare_flow_bwd_blocks(n_pixels)][...]
 * rows summed by glare_reduce_parts_f32: gwz [64][9]; gMt = 9 entries of dL/dM then 3 of dL/dt. */
int glare_flow_bwd_blocks(long long n_pixels);
int glare_flow_nll_backward_f32(const float* z, const float* mean, const float* g_logp_per_sample, int B,
                                long long pixels_per_sample, float* gz, float* gmean, glare_stream_t stream);
int glare_flow_fwd_post_backward_f32(float* gz, const float* z_pre, const float* h4, const float* g_logdet_per_sample, int B,
                                     long long pixels_per_sample, float eps, void* gh4_bf16x8, glare_stream_t stream);
int glare_flow_h1_backward_f32(float* gz, const void* gh1_bf16, int g_pitch, int g_off, const float* z_pre,
                               const float* wz_64x9, int B, int H, int W, float* gwz_partial, glare_stream_t stream);
int glare_flow_fwd_pre_backward_f32(float* gz, const float* z_in, const float* hF, int hF_pitch, int hF_off,
                                    const float* g_logdet_per_sample, int B, long long pixels_per_sample,
                                    const float* M_3x3_host, const float* t_3_host, float eps, void* ghF_bf16, int ghF_pitch,
                                    int ghF_off, float* gMt_partial, glare_stream_t stream);
int glare_flow_fwd_pre_backward_dev_f32(float* gz, const float* z_in, const float* hF, int hF_pitch, int hF_off,
                                        const float* g_logdet_per_sample, int B, long long pixels_per_sample,
                                        const float* Mt_12_device, float eps, void* ghF_bf16, int ghF_pitch, int ghF_off,
                                        float* gMt_partial, glare_stream_t stream);

/* Backward of the a8 glue (stage-3 step, row a13).
 * glare_mix_backward_bf16: out = s a + (1-s) b, s = sigmoid(w): gb = (1-s) g, ga = s g (or NULL), *dw_out = s(1-s) sum g(a-b);
 *   n elements (n % 8 == 0); workspace >= 512 floats.
 * glare_mean_rescale_backward_bf16: out = h + xw * (sum h / sum xw): gh (bf16), gxw (fp32) from g (bf16); means per
 *   sample or over the whole per-rank batch (training), matching glare_mean_rescale_bf16.
 * glare_sigmoid_f32: y = sigmoid(x) (the DCN mask the backward kernels take explicitly). */
int glare_mix_backward_bf16(const void* g, const void* a, const void* b, void* ga_or_null, void* gb, long long n, float w,
                            float* dw_out, void* workspace, size_t workspace_bytes, glare_stream_t stream);
/* the same two with the mixing logit read from DEVICE memory (training: no host synchronisation inside the step) */
int glare_mix_dev_bf16(const void* a, int a_pitch, int a_off, const void* b, int b_pitch, int b_off, void* out, int out_pitch,
                       int out_off, long long n_pixels, int C, const float* mix_w_device, glare_stream_t stream);
int glare_mix_backward_dev_bf16(const void* g, const void* a, const void* b, void* ga_or_null, void* gb, long long n,
                                const float* w_device, float* dw_out, void* workspace, size_t workspace_bytes,
                                glare_stream_t stream);
size_t glare_mean_rescale_backward_workspace_bytes(int B, long long n_per_sample);
int glare_mean_rescale_backward_bf16(const void* g, const void* h, const float* xw, void* gh, float* gxw, int B,
                                     long long n_per_sample, int whole_batch_mean, void* workspace, size_t workspace_bytes,
                                     glare_stream_t stream);
int glare_sigmoid_f32(const float* x, float* y, long long n, glare_stream_t stream);

/* Stage-3 pixel loss and its gradient in one pass (VQLLFLOWD_model.py:209-217): sr = clamp(rec, 0, 1) with NaN -> 0 and
 * masked, *loss_out = mean |sr - gt|, grad = d loss / d rec.  rec, grad: NHWC fp32 [B][HW][C]; gt: NCHW fp32.
 * workspace >= 512 floats.  (The VGG-perceptual and MS-SSIM terms of :219-220 are SURVEY.md row f1, not built.) */
int glare_l1_clamp_loss_f32(const float* rec_nhwc, const float* gt_nchw, int B, long long HW, int C, float* loss_out,
                            float* grad_nhwc, void* workspace, size_t workspace_bytes, glare_stream_t stream);

/* ---- a14 / f2: the inference harness' pre- and post-processing on the device (code/infer_dataset_lol.py:113-153) ------
 * pre : uint8 HWC [B][H][W][3] -> fp32 NCHW [B][3][H+pad][W+pad]: reflect pad bottom / left (impad :71-72), /255 (t :42),
 *       log(clamp(x + 1e-3, min = 1e-3)) (:127-128).
 * post: network output fp32 NCHW [B][3][Hp][Wp] -> restored float HWC [B][h][w][3] = clamp(out[:, :, :h, pad:], 0, 1)
 *       (:135-140); with gt (uint8 HWC [B][h][w][3]): gain = gray(gt/255)/gray(restored) with gray = 0.114 ch0 + 0.587 ch1 +
 *       0.299 ch2 (:142-144), clip, and psnr[b] = 10 log10(1 / mean((gt/255 - restored)^2)) (utils2.py:32-36). */
int glare_harness_preprocess_u8(const unsigned char* img_hwc, int B, int H, int W, int pad, float* out_nchw,
                                glare_stream_t stream);
/* SSIM of the evaluation scripts (calculate_ssim, code/utils/utils2.py:42-89, called on img_as_ubyte(target) / img_as_ubyte(restored),
 * infer_dataset_lol.py:152): this entry produces the two uint8-valued operands as fp32 planes in [0, 255] (restored rounded as
 * skimage's img_as_ubyte does); the windowed statistics are glare_ssim_forward_f32 (11-tap Gaussian sigma 1.5 = cv2.getGaussianKernel,
 * valid positions only, C1 = (0.01*255)^2, C2 = (0.03*255)^2), one call per image; its ssim mean over positions and channels is the metric. */
int glare_harness_ubyte_planes_f32(const float* restored_hwc, const unsigned char* gt_hwc, long long n, float* x255, float* y255,
                                   glare_stream_t stream);
/* img_as_ubyte(restored) (infer_dataset_lol.py:146-150, the image the loop saves): uint8 = rint(clip(x, 0, 1) * 255), n elements. */
int glare_harness_to_ubyte(const float* restored_hwc, long long n, unsigned char* out_u8, glare_stream_t stream);
size_t glare_harness_postprocess_workspace_bytes(int B);
int glare_harness_postprocess_f32(const float* out_nchw, const unsigned char* gt_hwc_or_null, int B, int h, int w, int Hp,
                                  int Wp, int pad, float* restored_hwc, double* psnr_or_null, void* workspace,
                                  size_t workspace_bytes, glare_stream_t stream);
/* ... and nonfinite_or_null[b] (int32 [B]) = the number of inf / NaN values of the network output inside image b's crop, counted BEFORE
 * the clamp: torch.clamp turns +inf into 1.0, so an fp16 overflow upstream can leave a finite PSNR behind (the reference masks NaNs only
 * in its training loss, VQLLFLOWD_model.py:214-217); glare_amd.infer re-runs a flagged image in bf16 and lists it. */
int glare_harness_postprocess_flagged_f32(const float* out_nchw, const unsigned char* gt_hwc_or_null, int B, int h, int w, int Hp,
                                          int Wp, int pad, float* restored_hwc, double* psnr_or_null, int* nonfinite_or_null,
                                          void* workspace, size_t workspace_bytes, glare_stream_t stream);

/* ---- f1: the rest of the stage-3 loss (VQLLFLOWD_model.py:209-223) -------------------------------------------------
 * glare_clamp01_f32 / _backward: sr = clamp(rec, 0, 1) with NaN -> 0 (:209-215) and torch.clamp's gradient mask.
 * MS-SSIM (modules/pytorch_msssim/__init__.py:21-98) on NHWC fp32 images: per pyramid level glare_ssim_forward_f32 writes the
 *   five windowed moments per output pixel (moments: [B][OH][OW][C][5], OH = H - window + 1) and ssim_cs_out[2] = (mean
 *   ssim_map, mean cs_map); glare_avgpool2_f32 builds the next level (F.avg_pool2d 2x2).  glare_ssim_backward_f32 turns the
 *   gradients of the ten level scalars (g_sim_cs_5_5: d/d sim[0..4] then d/d cs[0..4]) into the image gradient of `level`,
 *   adding the pooled gradient of the next level (g_next: [B][H/2][W/2][C] or NULL); scratch_maps: [B][OH][OW][C][3].
 *   window_host: the 1-D normalised Gaussian (<= 11 taps) exactly as create_window() builds it; C1, C2 as ssim() does.
 * VGG16-feature perceptual loss (modules/losses.py:12-40): convolutions = glare_conv2d_bf16 (+ReLU); here the 2x2 max-pool
 *   (backward to the first maximum in scan order, as ATen) and F.mse_loss of two bf16 feature maps with grad 2(a-b)/n. */
int glare_clamp01_f32(const float* x, float* y, long long n, glare_stream_t stream);
int glare_clamp01_backward_f32(const float* x, const float* g, float* gx, long long n, glare_stream_t stream);
int glare_avgpool2_f32(const float* x_nhwc, float* y_nhwc, int B, int H, int W, int C, glare_stream_t stream);
size_t glare_ssim_workspace_bytes(void);
int glare_ssim_forward_f32(const float* x_nhwc, const float* y_nhwc, int B, int H, int W, int C, const float* window_host,
                           int window_size, float C1, float C2, float* moments, float* ssim_cs_out, void* workspace,
                           size_t workspace_bytes, glare_stream_t stream);
int glare_ssim_backward_f32(const float* x_nhwc, const float* y_nhwc, const float* moments, int B, int H, int W, int C,
                            const float* window_host, int window_size, float C1, float C2, const float* g_sim_cs_5_5,
                            int level, const float* g_next_or_null, float* scratch_maps, float* gx, glare_stream_t stream);
int glare_maxpool2_bf16(const void* x_nhwc, void* y_nhwc, int B, int H, int W, int C, glare_stream_t stream);
int glare_maxpool2_backward_bf16(const void* x_nhwc, const void* g_nhwc, void* gx_nhwc, int B, int H, int W, int C,
                                 glare_stream_t stream);
int glare_mse_loss_bf16(const void* a, const void* b, long long n, float* loss_out, void* grad_a_or_null, void* workspace,
                        size_t workspace_bytes, glare_stream_t stream);
/* grad_a = round16(2 (a - b) / n * g_dev[0]): the upstream scalar gradient (incl. any loss scale) is multiplied in fp32 BEFORE the
 * 16-bit rounding, as F.mse_loss under autocast (fp32) + the cast's backward do (losses.py:33-39 behind scaler.scale(loss).backward()) */
int glare_mse_backward_bf16(const void* a, const void* b, long long n, const float* g_dev, void* grad_a, glare_stream_t stream);

/* ---- f4: LPIPS (AlexNet) of the evaluation loop ---------------------------------------------------------------------------
 * Replaces `lpips.LPIPS(net='alex').forward(tA, tB)` as called by Measure.lpips (code/Measure.py:17-30; infer_dataset_lol.py:153).
 * The `lpips` package (Zhang et al. 2018, PyPI lpips 0.1.x) is a third-party dependency absent from the reference tree; its published
 * forward is restated by oracle/torch_ref.py LPIPSAlex.  All fp32, NCHW, as the package computes it (csrc/metrics.hip):
 * glare_conv2d_direct_f32: out = act(bias + conv(x; w [Cout][Cin][k][k], stride, zero padding)); in_shift / in_scale (both or neither,
 *   fp32 [Cin]): the input is (x - shift) / scale, padded with zeros AFTER the scaling (LPIPS' ScalingLayer in front of conv1).
 *   Any Cin / Cout / stride / padding, k <= 16.   glare_maxpool2d_f32: nn.MaxPool2d(k, stride) (no padding, floor mode).
 * glare_lpips_tap_f32: one feature tap -- both feature maps unit-normalised over the channels (x / (|x| + eps)), squared difference,
 *   the 1x1 head lin_w [C], spatial mean; dist_out[b] (fp64) = or += the tap's value (accumulate), deterministic two-level sum. */
int glare_conv2d_direct_f32(const float* x_nchw, const float* w_oihw, const float* bias_or_null, float* out_nchw, int B, int Cin, int H,
                            int W, int Cout, int ksize, int stride, int pad, int relu, const float* in_shift_or_null,
                            const float* in_scale_or_null, glare_stream_t stream);
int glare_maxpool2d_f32(const float* x_nchw, float* out_nchw, int B, int C, int H, int W, int ksize, int stride, glare_stream_t stream);
size_t glare_lpips_tap_workspace_bytes(int B, long long HW);
int glare_lpips_tap_f32(const float* feat0_nchw, const float* feat1_nchw, const float* lin_w, int B, int C, long long HW, float eps,
                        int accumulate, double* dist_out, void* workspace, size_t workspace_bytes, glare_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GLARE_HIP_H */
