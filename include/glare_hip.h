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
  int upsample;              /* nearest x2 on the input first                                      */
  int act;                   /* GLARE_ACT_*                                                        */
  int out_mode;              /* GLARE_OUT_*                                                        */
  long long plane_pitch;     /* planar modes: elements per plane (>= OH*OW); 0 = OH*OW             */
} glare_conv_desc;

/* Number of bf16 elements of the packed weight image for an OIHW [cout][cin_total][k][k] filter. */
long long glare_conv2d_packed_weight_elems(int cout, int cin_total, int ksize);
/* Packs fp32 OIHW weights (device) into the kernel's stage-ordered bf16 image (device). */
int glare_conv2d_pack_weight(const float* w_oihw, int cout, int cin_total, int ksize, void* packed_bf16,
                             glare_stream_t stream);
int glare_conv2d_bf16(const glare_conv_desc* desc_host, glare_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GLARE_HIP_H */
