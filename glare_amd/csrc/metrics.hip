// LPIPS (AlexNet variant) of the evaluation loop, on the device.
//
// Reference call sites: code/Measure.py:17-30 (`lpips.LPIPS(net='alex')`, `self.model.forward(tA, tB)` on images scaled to [-1, 1])
// from code/infer_dataset_lol.py:151-153, infer_dataset_lolv2-real.py:150, test_stage3.py:205.  The algorithm lives in the third-party
// package `lpips` (Zhang et al., CVPR 2018; PyPI lpips 0.1.x, unpinned by the reference), absent from /root/reference and from this
// image; its published forward (version 0.1, spatial = False) is restated by oracle/torch_ref.py `LPIPSAlex` and computed here:
//   scaling layer -> five AlexNet feature taps (conv 11x11 s4 p2, maxpool 3 s2, conv 5x5 p2, maxpool, three 3x3 convs, ReLU after
//   each conv) -> per tap: unit-normalise the channel vector of both images (x / (|x| + 1e-10)), squared difference, 1x1 "lin"
//   head (C -> 1, no bias), spatial mean; the distance is the sum over the taps.
// Everything is fp32 as in the package (a metric, not a hot path: ~6.5 GFLOP per image pair at 400 x 600): a direct NCHW convolution
// with the filter block of 32 output channels staged through LDS, a 3x3 / stride-2 max pool, and a two-level deterministic
// reduction per tap (no atomics).
#include "common.h"

namespace {

constexpr int MC_PIX = 64, MC_CO = 32, MC_COT = 8;      // pixels / output channels per block, output channels per thread
constexpr int MC_LDS_FLOATS = 8192;                      // filter block: MC_CO x (cin chunk x k x k) floats

// out[b][co][oy][ox] = act(bias[co] + sum_{ci,ky,kx} w[co][ci][ky][kx] x[b][ci][oy*s - p + ky][ox*s - p + kx]); optional per-channel
// input affine (x - shift[ci]) / scale[ci] applied to IN-RANGE taps only (= zero padding of the scaled tensor: LPIPS' scaling layer)
__global__ __launch_bounds__(256) void conv_direct_f32_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bias, float* __restrict__ out, int Cin, int H,
                                                              int W, int Cout, int OH, int OW, int k, int stride, int pad, int relu,
                                                              int ci_chunk, const float* __restrict__ in_shift,
                                                              const float* __restrict__ in_scale) {
  __shared__ float wl[MC_LDS_FLOATS];
  const int b = blockIdx.z, co0 = blockIdx.y * MC_CO;
  const int px = threadIdx.x & (MC_PIX - 1), cg = threadIdx.x / MC_PIX;     // cg: which 8 of the block's 32 output channels (wave-uniform)
  const int p = blockIdx.x * MC_PIX + px;
  const bool live = p < OH * OW;
  const int oy = live ? p / OW : 0, ox = live ? p - oy * OW : 0;
  const int iy0 = oy * stride - pad, ix0 = ox * stride - pad;
  const int kk = k * k;
  float acc[MC_COT];
#pragma unroll
  for (int e = 0; e < MC_COT; ++e) acc[e] = 0.f;
  const float* xb = x + (size_t)b * Cin * H * W;
  for (int c0 = 0; c0 < Cin; c0 += ci_chunk) {
    const int nc = min(ci_chunk, Cin - c0);
    __syncthreads();
    // filter block [co 0..31][ci 0..nc)[kk] -> LDS as [ci][kk][co]: the 8 channels of a thread are two broadcast 16-B reads
    for (int i = threadIdx.x; i < MC_CO * nc * kk; i += 256) {
      const int co = i / (nc * kk), r = i - co * nc * kk;
      wl[r * MC_CO + co] = (co0 + co < Cout) ? w[((size_t)(co0 + co) * Cin + c0) * kk + r] : 0.f;
    }
    __syncthreads();
    for (int ci = 0; ci < nc; ++ci) {
      const float* xc = xb + (size_t)(c0 + ci) * H * W;
      const float sh = in_shift ? in_shift[c0 + ci] : 0.f, sc = in_scale ? in_scale[c0 + ci] : 1.f;
      for (int ky = 0; ky < k; ++ky) {
        const int iy = iy0 + ky;
        const bool yok = live && iy >= 0 && iy < H;
        for (int kx = 0; kx < k; ++kx) {
          const int ix = ix0 + kx;
          float v = 0.f;
          if (yok && ix >= 0 && ix < W) v = (xc[(size_t)iy * W + ix] - sh) / sc;
          const float* wr = wl + ((ci * kk + ky * k + kx) * MC_CO + cg * MC_COT);
          const f32x4 w0 = *reinterpret_cast<const f32x4*>(wr), w1 = *reinterpret_cast<const f32x4*>(wr + 4);
          acc[0] = fmaf(v, w0[0], acc[0]); acc[1] = fmaf(v, w0[1], acc[1]); acc[2] = fmaf(v, w0[2], acc[2]); acc[3] = fmaf(v, w0[3], acc[3]);
          acc[4] = fmaf(v, w1[0], acc[4]); acc[5] = fmaf(v, w1[1], acc[5]); acc[6] = fmaf(v, w1[2], acc[6]); acc[7] = fmaf(v, w1[3], acc[7]);
        }
      }
    }
  }
  if (!live) return;
#pragma unroll
  for (int e = 0; e < MC_COT; ++e) {
    const int co = co0 + cg * MC_COT + e;
    if (co < Cout) {
      float v = acc[e] + (bias ? bias[co] : 0.f);
      if (relu) v = fmaxf(v, 0.f);
      out[((size_t)b * Cout + co) * OH * OW + p] = v;
    }
  }
}

// nn.MaxPool2d(k, stride), no padding, floor mode, NCHW fp32
__global__ __launch_bounds__(256) void maxpool_f32_kernel(const float* __restrict__ x, float* __restrict__ out, long long planes, int H,
                                                          int W, int OH, int OW, int k, int stride) {
  const long long total = planes * OH * OW;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int ox = (int)(i % OW), oy = (int)((i / OW) % OH);
    const long long pl = i / ((long long)OW * OH);
    const float* xp = x + pl * H * W + (size_t)(oy * stride) * W + ox * stride;
    float m = xp[0];
    for (int ky = 0; ky < k; ++ky)
      for (int kx = 0; kx < k; ++kx) m = fmaxf(m, xp[ky * W + kx]);
    out[i] = m;
  }
}

// one LPIPS tap: partial[b][blk] = sum over the block's pixels of sum_c lin[c] * (f0/(|f0| + eps) - f1/(|f1| + eps))^2
__global__ __launch_bounds__(256) void lpips_tap_kernel(const float* __restrict__ f0, const float* __restrict__ f1,
                                                        const float* __restrict__ lin, int C, int HW, float eps, float* __restrict__ partial) {
  __shared__ float red[4];
  const int b = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
  float acc = 0.f;
  if (p < HW) {
    const float* a = f0 + (size_t)b * C * HW + p;
    const float* c = f1 + (size_t)b * C * HW + p;
    float n0 = 0.f, n1 = 0.f;
    for (int ch = 0; ch < C; ++ch) {
      const float u = a[(size_t)ch * HW], v = c[(size_t)ch * HW];
      n0 = fmaf(u, u, n0);
      n1 = fmaf(v, v, n1);
    }
    const float d0 = sqrtf(n0) + eps, d1 = sqrtf(n1) + eps;
    for (int ch = 0; ch < C; ++ch) {
      const float d = a[(size_t)ch * HW] / d0 - c[(size_t)ch * HW] / d1;
      acc = fmaf(lin[ch], d * d, acc);
    }
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[(size_t)b * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// out[b] (+)= sum(partial[b][:]) / HW in fp64, one thread per image: the taps are accumulated by consecutive launches in a fixed order
__global__ void lpips_finish_kernel(const float* __restrict__ partial, int nblk, int HW, int B, int accumulate, double* __restrict__ out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double s = 0.0;
  for (int i = 0; i < nblk; ++i) s += (double)partial[(size_t)b * nblk + i];
  s /= (double)HW;
  out[b] = accumulate ? out[b] + s : s;
}

}  // namespace

extern "C" int glare_conv2d_direct_f32(const float* x_nchw, const float* w_oihw, const float* bias_or_null, float* out_nchw, int B, int Cin,
                                       int H, int W, int Cout, int ksize, int stride, int pad, int relu, const float* in_shift_or_null,
                                       const float* in_scale_or_null, glare_stream_t stream) {
  if (!x_nchw || !w_oihw || !out_nchw || B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0 || ksize <= 0 || stride <= 0 || pad < 0)
    return GLARE_ERR_INVALID;
  if ((in_shift_or_null == nullptr) != (in_scale_or_null == nullptr)) return GLARE_ERR_INVALID;
  const int OH = (H + 2 * pad - ksize) / stride + 1, OW = (W + 2 * pad - ksize) / stride + 1;
  if (OH <= 0 || OW <= 0) return GLARE_ERR_INVALID;
  const int kk = ksize * ksize;
  if (MC_CO * kk > MC_LDS_FLOATS || B > 65535) return GLARE_ERR_UNSUPPORTED;      // kernel <= 16 x 16
  int ci_chunk = MC_LDS_FLOATS / (MC_CO * kk);
  if (ci_chunk > Cin) ci_chunk = Cin;
  const dim3 grid((unsigned)cdiv(OH * OW, MC_PIX), (unsigned)cdiv(Cout, MC_CO), (unsigned)B);
  hipLaunchKernelGGL(conv_direct_f32_kernel, grid, dim3(256), 0, (hipStream_t)stream, x_nchw, w_oihw, bias_or_null, out_nchw, Cin, H, W, Cout,
                     OH, OW, ksize, stride, pad, relu, ci_chunk, in_shift_or_null, in_scale_or_null);
  return glare_launch_status();
}

extern "C" int glare_maxpool2d_f32(const float* x_nchw, float* out_nchw, int B, int C, int H, int W, int ksize, int stride,
                                   glare_stream_t stream) {
  if (!x_nchw || !out_nchw || B <= 0 || C <= 0 || ksize <= 0 || stride <= 0 || H < ksize || W < ksize) return GLARE_ERR_INVALID;
  const int OH = (H - ksize) / stride + 1, OW = (W - ksize) / stride + 1;
  const long long total = (long long)B * C * OH * OW;
  long long blocks = cdivll(total, 256);
  if (blocks > 65535) blocks = 65535;
  hipLaunchKernelGGL(maxpool_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x_nchw, out_nchw, (long long)B * C, H, W,
                     OH, OW, ksize, stride);
  return glare_launch_status();
}

extern "C" size_t glare_lpips_tap_workspace_bytes(int B, long long HW) {
  if (B <= 0 || HW <= 0) return 0;
  return (size_t)B * cdivll(HW, 256) * sizeof(float);
}

extern "C" int glare_lpips_tap_f32(const float* feat0_nchw, const float* feat1_nchw, const float* lin_w, int B, int C, long long HW,
                                   float eps, int accumulate, double* dist_out, void* workspace, size_t workspace_bytes,
                                   glare_stream_t stream) {
  if (!feat0_nchw || !feat1_nchw || !lin_w || !dist_out || B <= 0 || B > 65535 || C <= 0 || HW <= 0 || HW > 0x7fffffffLL)
    return GLARE_ERR_INVALID;
  if (!workspace || workspace_bytes < glare_lpips_tap_workspace_bytes(B, HW)) return GLARE_ERR_WORKSPACE;
  const int nblk = (int)cdivll(HW, 256);
  hipLaunchKernelGGL(lpips_tap_kernel, dim3((unsigned)nblk, (unsigned)B), dim3(256), 0, (hipStream_t)stream, feat0_nchw, feat1_nchw, lin_w, C,
                     (int)HW, eps, (float*)workspace);
  hipLaunchKernelGGL(lpips_finish_kernel, dim3((unsigned)cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, (const float*)workspace, nblk, (int)HW,
                     B, accumulate, dist_out);
  return glare_launch_status();
}
