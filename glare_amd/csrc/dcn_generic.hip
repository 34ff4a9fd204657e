// General-shape modulated deformable convolution (DCNv2), forward and backward, in the reference's layouts.
//
// glare_mdcn_forward_f32 / glare_mdcn_backward_f32 are drop-ins for deform_conv_ext.modulated_deform_conv_forward / _backward
// (ops/dcn/src/deform_conv_ext.cpp:107-147), which accept ANY group / deformable_group / channel count / kernel / stride /
// padding / dilation (shape checks only: deform_conv_cuda.cpp:497-516).  The MFMA kernels of dcn.hip / dcn_bwd.hip cover the
// configurations GLARE runs (groups = 1, 32 or 64 channels per deformable group, Co a multiple of 64); everything else lands
// here instead of GLARE_ERR_UNSUPPORTED: plain fp32 HIP kernels, no MFMA, no `columns` buffer, no workspace.
//   forward : a workgroup owns 64 output pixels x 64 output channels of one conv group; per chunk of input channels the
//             masked bilinear samples go to LDS once ([c][tap][pixel]) and are contracted from there
//             (deform_conv_cuda_kernel.cu:468-497,571-633 + the addmm_ of deform_conv_cuda.cpp:551-554).
//   backward: (1) one thread per (sample, deformable group, tap, pixel) walks the group's channels: cv = sum_co W gO
//             (the `columns = W^T gO` GEMM, deform_conv_cuda.cpp:613-616), from it grad_offset / grad_mask written directly
//             (kernel.cu:695-767: deterministic) and grad_input scattered with fp32 atomics (kernel.cu:635-693, as the reference);
//             (2) one workgroup per (input channel, tap) x 16 output channels reduces grad_weight over samples and pixels
//             (deform_conv_cuda.cpp:640-672); (3) grad_bias.
// Arithmetic is fp32 throughout (the reference's), results agree with it to accumulation order.
#include "common.h"
#include "dcn_generic.h"

namespace {

struct GParams {
  const float* x;       // [B][C][H][W]
  const float* offset;  // [B][dg*2*K][Ho][Wo]
  const float* mask;    // [B][dg*K][Ho][Wo]
  const float* weight;  // [Co][C/groups][kh][kw]
  const float* bias;
  float* out;           // [B][Co][Ho][Wo]
  const float* gout;
  float *gx, *goff, *gmask, *gw, *gb;
  int B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, groups, dg, Ho, Wo;
  int cchunk;           // input channels sampled per LDS pass (forward)
};

// A sampling point (h, w) of a plane H x W as the kernels of dcn.hip see it: the linear index of its top-left corner, the two
// fractional parts and a validity predicate per corner (a corner outside the plane contributes zero: the per-corner zeroing
// of the reference's sampler, deform_conv_cuda_kernel.cu:468-497).  The interpolated value is sum_c wgt_c v_c with the four
// products of (1 - fy, fy) x (1 - fx, fx); its derivative along y (x) replaces the y (x) factor by (-1, +1) -- the reference
// tabulates the same two sums corner by corner in dmcn_get_coordinate_weight (:528-568).
struct Corners {
  int base;            // y0 * W + x0 (may be negative: only dereferenced where ok[] holds)
  float fy, fx;
  bool ok[4];          // (y0,x0) (y0,x1) (y1,x0) (y1,x1)
};

__device__ __forceinline__ Corners corners_at(int H, int W, float h, float w) {
  Corners c;
  const float y0f = floorf(h), x0f = floorf(w);
  const int y0 = (int)y0f, x0 = (int)x0f;
  c.base = y0 * W + x0;
  c.fy = h - y0f;
  c.fx = w - x0f;
  const bool top = y0 >= 0, bot = y0 + 1 < H, lef = x0 >= 0, rig = x0 + 1 < W;
  c.ok[0] = top && lef; c.ok[1] = top && rig; c.ok[2] = bot && lef; c.ok[3] = bot && rig;
  return c;
}

__device__ __forceinline__ void corner_values(const float* im, int W, const Corners& c, float (&v)[4]) {
  v[0] = c.ok[0] ? im[c.base] : 0.f;
  v[1] = c.ok[1] ? im[c.base + 1] : 0.f;
  v[2] = c.ok[2] ? im[c.base + W] : 0.f;
  v[3] = c.ok[3] ? im[c.base + W + 1] : 0.f;
}

__device__ __forceinline__ float bilinear(const float* im, int H, int W, float h, float w) {
  const Corners c = corners_at(H, W, h, w);
  float v[4];
  corner_values(im, W, c, v);
  const float ty = 1.f - c.fy, tx = 1.f - c.fx;
  return ty * tx * v[0] + ty * c.fx * v[1] + c.fy * tx * v[2] + c.fy * c.fx * v[3];
}

// d(sample)/d(coordinate): dir 0 = along y, 1 = along x; zero outside the open interval the forward samples in
__device__ __forceinline__ float coordinate_weight(const float* im, int H, int W, float ah, float aw, int dir) {
  if (!(ah > -1.f && ah < (float)H && aw > -1.f && aw < (float)W)) return 0.f;
  const Corners c = corners_at(H, W, ah, aw);
  float v[4];
  corner_values(im, W, c, v);
  const float tx = 1.f - c.fx, ty = 1.f - c.fy;
  // accumulated in the corner order (y0,x0), (y0,x1), (y1,x0), (y1,x1)
  if (dir == 0) return ((-tx * v[0] - c.fx * v[1]) + tx * v[2]) + c.fx * v[3];
  return ((-ty * v[0] + ty * v[1]) - c.fy * v[2]) + c.fy * v[3];
}

constexpr int GP = 64;    // output pixels per workgroup
constexpr int GCO = 64;   // output channels per workgroup (16 per thread)

__global__ __launch_bounds__(256) void dcn_generic_fwd_kernel(const GParams p) {
  extern __shared__ float col[];   // [cchunk][K][GP]
  const int K = p.kh * p.kw, HWo = p.Ho * p.Wo, Cg = p.C / p.groups, Cog = p.Co / p.groups, cpg = p.C / p.dg;
  const int co_tiles = (Cog + GCO - 1) / GCO;
  const int grp = blockIdx.z / co_tiles, cot = blockIdx.z % co_tiles, b = blockIdx.y;
  const int pix0 = blockIdx.x * GP;
  const int tid = threadIdx.x, px = tid & 63, cq = tid >> 6;            // thread: pixel px, output channels co0 + cq*16 + j
  const int n = pix0 + px;
  const int co0 = grp * Cog + cot * GCO + cq * 16;
  const int co_end = min(grp * Cog + Cog, grp * Cog + cot * GCO + GCO);
  float acc[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = 0.f;
  const float* xb = p.x + (size_t)b * p.C * p.H * p.W;
  const float* offb = p.offset + (size_t)b * p.dg * 2 * K * HWo;
  const float* mb = p.mask + (size_t)b * p.dg * K * HWo;
  for (int c0 = 0; c0 < Cg; c0 += p.cchunk) {
    const int nc = min(p.cchunk, Cg - c0);
    __syncthreads();
    for (int e = tid; e < nc * K * GP; e += 256) {                       // masked samples of this channel chunk
      const int pp = e % GP, k = (e / GP) % K, cl = e / (GP * K);
      const int nn = pix0 + pp;
      float v = 0.f;
      if (nn < HWo) {
        const int c = grp * Cg + c0 + cl, dgi = c / cpg;
        const int ho = nn / p.Wo, wo = nn % p.Wo, i = k / p.kw, j = k % p.kw;
        const float oh = offb[((size_t)dgi * 2 * K + 2 * k) * HWo + nn], ow = offb[((size_t)dgi * 2 * K + 2 * k + 1) * HWo + nn];
        const float h_im = ho * p.sh - p.ph + i * p.dh + oh, w_im = wo * p.sw - p.pw + j * p.dw + ow;
        if (h_im > -1.f && w_im > -1.f && h_im < (float)p.H && w_im < (float)p.W)        // kernel.cu:618
          v = bilinear(xb + (size_t)c * p.H * p.W, p.H, p.W, h_im, w_im) * mb[((size_t)dgi * K + k) * HWo + nn];
      }
      col[e] = v;
    }
    __syncthreads();
    for (int cl = 0; cl < nc; ++cl)
      for (int k = 0; k < K; ++k) {
        const float s = col[(cl * K + k) * GP + px];
#pragma unroll
        for (int j = 0; j < 16; ++j)
          if (co0 + j < co_end) acc[j] = fmaf(p.weight[((size_t)(co0 + j) * Cg + c0 + cl) * K + k], s, acc[j]);
      }
  }
  if (n < HWo)
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (co0 + j < co_end) p.out[((size_t)b * p.Co + co0 + j) * HWo + n] = acc[j] + (p.bias ? p.bias[co0 + j] : 0.f);
}

// one thread per (b, deformable group, tap, pixel): grad_offset (2 values), grad_mask, grad_input scatter
__global__ __launch_bounds__(256) void dcn_generic_bwd_data_kernel(const GParams p) {
  const int K = p.kh * p.kw, HWo = p.Ho * p.Wo, Cg = p.C / p.groups, Cog = p.Co / p.groups, cpg = p.C / p.dg;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)p.B * p.dg * K * HWo) return;
  const int n = (int)(t % HWo), k = (int)((t / HWo) % K), dgi = (int)((t / ((long long)HWo * K)) % p.dg);
  const int b = (int)(t / ((long long)HWo * K * p.dg));
  const int ho = n / p.Wo, wo = n % p.Wo, i = k / p.kw, j = k % p.kw;
  const float oh = p.offset[(((size_t)b * p.dg + dgi) * 2 * K + 2 * k) * HWo + n];
  const float ow = p.offset[(((size_t)b * p.dg + dgi) * 2 * K + 2 * k + 1) * HWo + n];
  const float m = p.mask[(((size_t)b * p.dg + dgi) * K + k) * HWo + n];
  const float ah = ho * p.sh - p.ph + i * p.dh + oh, aw = wo * p.sw - p.pw + j * p.dw + ow;
  const bool inside = !(ah <= -1.f || aw <= -1.f || ah >= (float)p.H || aw >= (float)p.W);
  const int hl = (int)floorf(ah), wl = (int)floorf(aw);
  float g_h = 0.f, g_w = 0.f, g_m = 0.f;
  for (int cc = 0; cc < cpg; ++cc) {
    const int c = dgi * cpg + cc, grp = c / Cg, cl = c % Cg;
    float cv = 0.f;                                                       // columns[c*K + k][n] = sum_co W[co][cl][k] gO[b][co][n]
    for (int co = 0; co < Cog; ++co)
      cv = fmaf(p.weight[((size_t)(grp * Cog + co) * Cg + cl) * K + k], p.gout[((size_t)b * p.Co + grp * Cog + co) * HWo + n], cv);
    const float* im = p.x + ((size_t)b * p.C + c) * p.H * p.W;
    if (inside) {
      g_m += cv * bilinear(im, p.H, p.W, ah, aw);                         // kernel.cu:737-743
      g_h += coordinate_weight(im, p.H, p.W, ah, aw, 0) * cv * m;         // :744-749
      g_w += coordinate_weight(im, p.H, p.W, ah, aw, 1) * cv * m;
      if (p.gx) {                                                         // col2im, kernel.cu:635-693: the four bilinear corners
        float* gim = p.gx + ((size_t)b * p.C + c) * p.H * p.W;
        const float top = cv * m;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
          for (int dx = 0; dx < 2; ++dx) {
            const int y = hl + dy, x = wl + dx;
            if (y >= 0 && y < p.H && x >= 0 && x < p.W) {
              const float wy = dy ? (ah + 1 - y) : (y + 1 - ah), wx = dx ? (aw + 1 - x) : (x + 1 - aw);   // dmcn_get_gradient_weight
              atomicAdd(gim + y * p.W + x, wy * wx * top);
            }
          }
      }
    }
  }
  p.goff[(((size_t)b * p.dg + dgi) * 2 * K + 2 * k) * HWo + n] = g_h;
  p.goff[(((size_t)b * p.dg + dgi) * 2 * K + 2 * k + 1) * HWo + n] = g_w;
  p.gmask[(((size_t)b * p.dg + dgi) * K + k) * HWo + n] = g_m;
}

// grad_weight[co][cl][k] += sum_{b, n} gO[b][co][n] * sample(b, c, k, n) * mask: workgroup = (c, k) x 16 output channels
__global__ __launch_bounds__(256) void dcn_generic_bwd_weight_kernel(const GParams p) {
  __shared__ float red[4][16];
  const int K = p.kh * p.kw, HWo = p.Ho * p.Wo, Cg = p.C / p.groups, Cog = p.Co / p.groups, cpg = p.C / p.dg;
  const int c = blockIdx.x / K, k = blockIdx.x % K, grp = c / Cg, cl = c % Cg, dgi = c / cpg;
  const int co0 = grp * Cog + blockIdx.y * 16, co_end = min(grp * Cog + Cog, co0 + 16);
  const int i = k / p.kw, j = k % p.kw;
  float acc[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  for (long long e = threadIdx.x; e < (long long)p.B * HWo; e += 256) {
    const int b = (int)(e / HWo), n = (int)(e % HWo), ho = n / p.Wo, wo = n % p.Wo;
    const float oh = p.offset[(((size_t)b * p.dg + dgi) * 2 * K + 2 * k) * HWo + n];
    const float ow = p.offset[(((size_t)b * p.dg + dgi) * 2 * K + 2 * k + 1) * HWo + n];
    const float h_im = ho * p.sh - p.ph + i * p.dh + oh, w_im = wo * p.sw - p.pw + j * p.dw + ow;
    float s = 0.f;
    if (h_im > -1.f && w_im > -1.f && h_im < (float)p.H && w_im < (float)p.W)
      s = bilinear(p.x + ((size_t)b * p.C + c) * p.H * p.W, p.H, p.W, h_im, w_im) * p.mask[(((size_t)b * p.dg + dgi) * K + k) * HWo + n];
#pragma unroll
    for (int q = 0; q < 16; ++q)
      if (co0 + q < co_end) acc[q] = fmaf(p.gout[((size_t)b * p.Co + co0 + q) * HWo + n], s, acc[q]);
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const float v = wave_sum(acc[q]);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][q] = v;
  }
  __syncthreads();
  if (threadIdx.x < 16 && co0 + (int)threadIdx.x < co_end) {
    const int q = threadIdx.x;
    p.gw[((size_t)(co0 + q) * Cg + cl) * K + k] += (red[0][q] + red[1][q]) + (red[2][q] + red[3][q]);
  }
}

__global__ __launch_bounds__(256) void dcn_generic_bwd_bias_kernel(const GParams p) {
  __shared__ float red[4];
  const int co = blockIdx.x, HWo = p.Ho * p.Wo;
  float s = 0.f;
  for (long long e = threadIdx.x; e < (long long)p.B * HWo; e += 256)
    s += p.gout[((size_t)(e / HWo) * p.Co + co) * HWo + e % HWo];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) p.gb[co] += (red[0] + red[1]) + (red[2] + red[3]);
}

bool fill(GParams& p, int B, int C, int H, int W, int Co, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int groups,
          int dg) {
  p.B = B; p.C = C; p.H = H; p.W = W; p.Co = Co; p.kh = kh; p.kw = kw; p.sh = sh; p.sw = sw; p.ph = ph; p.pw = pw;
  p.dh = dh; p.dw = dw; p.groups = groups; p.dg = dg;
  p.Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1;
  p.Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1;
  return p.Ho > 0 && p.Wo > 0;
}

}  // namespace

// shape_check of the reference (deform_conv_cuda.cpp:497-516): channels divisible by both group counts
int glare_mdcn_generic_check(int B, int C, int H, int W, int Co, int kh, int kw, int sh, int sw, int dh, int dw, int groups, int dg) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || Co <= 0 || kh <= 0 || kw <= 0 || sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0 || dg <= 0 ||
      groups <= 0)
    return GLARE_ERR_INVALID;
  if (C % dg || C % groups || Co % groups) return GLARE_ERR_INVALID;
  if ((long long)kh * kw * GP * 4 > 48 * 1024) return GLARE_ERR_UNSUPPORTED;   // one channel's samples must fit the LDS pass
  return GLARE_OK;
}

int glare_mdcn_generic_forward(const float* x, const float* offset, const float* mask, const float* weight, const float* bias, float* out,
                               int B, int C, int H, int W, int Co, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                               int groups, int dg, hipStream_t stream) {
  GParams p{};
  p.x = x; p.offset = offset; p.mask = mask; p.weight = weight; p.bias = bias; p.out = out;
  if (!fill(p, B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, groups, dg)) return GLARE_ERR_INVALID;
  const int K = kh * kw, Cg = C / groups, Cog = Co / groups;
  p.cchunk = max(1, min(min(8, Cg), (48 * 1024) / (K * GP * 4)));
  const size_t lds = (size_t)p.cchunk * K * GP * sizeof(float);
  const dim3 grid((unsigned)cdiv(p.Ho * p.Wo, GP), (unsigned)B, (unsigned)(groups * cdiv(Cog, GCO)));
  if (grid.y > 65535u || grid.z > 65535u) return GLARE_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(dcn_generic_fwd_kernel, grid, dim3(256), lds, stream, p);
  return glare_launch_status();
}

int glare_mdcn_generic_backward(const float* x, const float* offset, const float* mask, const float* weight, const float* grad_out,
                                float* grad_input, float* grad_offset, float* grad_mask, float* grad_weight, float* grad_bias, int B,
                                int C, int H, int W, int Co, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int groups,
                                int dg, hipStream_t stream) {
  GParams p{};
  p.x = x; p.offset = offset; p.mask = mask; p.weight = weight; p.gout = grad_out;
  p.gx = grad_input; p.goff = grad_offset; p.gmask = grad_mask; p.gw = grad_weight; p.gb = grad_bias;
  if (!fill(p, B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, groups, dg)) return GLARE_ERR_INVALID;
  const int K = kh * kw, Cog = Co / groups;
  const long long HWo = (long long)p.Ho * p.Wo;
  if (grad_input &&   // overwritten (deform_conv_cuda.cpp:601-603 views the caller's zeroed buffer per sample): zero, then scatter
      hipMemsetAsync(grad_input, 0, sizeof(float) * (size_t)B * C * H * W, stream) != hipSuccess)
    return GLARE_ERR_LAUNCH;
  const long long nthreads = (long long)B * dg * K * HWo;
  if (cdivll(nthreads, 256) > 0x7fffffffLL || (long long)C * K > 0x7fffffffLL) return GLARE_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(dcn_generic_bwd_data_kernel, dim3((unsigned)cdivll(nthreads, 256)), dim3(256), 0, stream, p);
  hipLaunchKernelGGL(dcn_generic_bwd_weight_kernel, dim3((unsigned)(C * K), (unsigned)cdiv(Cog, 16)), dim3(256), 0, stream, p);
  if (grad_bias) hipLaunchKernelGGL(dcn_generic_bwd_bias_kernel, dim3((unsigned)Co), dim3(256), 0, stream, p);
  return glare_launch_status();
}
