// Internal: the general-shape DCNv2 kernels (dcn_generic.hip) behind glare_mdcn_forward_f32 / glare_mdcn_backward_f32.
#pragma once
#include <hip/hip_runtime.h>

int glare_mdcn_generic_check(int B, int C, int H, int W, int Co, int kh, int kw, int sh, int sw, int dh, int dw, int groups, int dg);
int glare_mdcn_generic_forward(const float* x, const float* offset, const float* mask, const float* weight, const float* bias, float* out,
                               int B, int C, int H, int W, int Co, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                               int groups, int dg, hipStream_t stream);
int glare_mdcn_generic_backward(const float* x, const float* offset, const float* mask, const float* weight, const float* grad_out,
                                float* grad_input, float* grad_offset, float* grad_mask, float* grad_weight, float* grad_bias, int B,
                                int C, int H, int W, int Co, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int groups,
                                int dg, hipStream_t stream);
