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

#ifdef __cplusplus
}
#endif
#endif /* GLARE_HIP_H */
