// Stage-3 loss stack on the device: the terms VQLLFLOWDModel.optimize_parameters adds to the L1 loss
// (code/models/VQLLFLOWD_model.py:209-223) -- MS-SSIM (modules/pytorch_msssim/__init__.py:21-98) and the VGG16-feature
// perceptual loss (modules/losses.py:12-40; its convolutions run on conv_igemm.hip, this file holds the 2x2 max-pool and
// the feature MSE) -- each with its backward.  Images are NHWC fp32 [B][H][W][C] (C = 3), sizes are tiny (256 x 256
// crops): every kernel is a direct, bandwidth-trivial loop; reductions are two-level and atomic-free.
#include "common.h"

namespace {

constexpr int LT = 256;

// sr = clamp(rec, 0, 1), NaN -> 0 (VQLLFLOWD_model.py:209-215)
__global__ __launch_bounds__(LT) void clamp01_kernel(const float* __restrict__ x, float* __restrict__ y, long long n) {
  const long long i = (long long)blockIdx.x * LT + threadIdx.x;
  if (i >= n) return;
  const float v = x[i];
  y[i] = v == v ? fminf(fmaxf(v, 0.f), 1.f) : 0.f;
}
// torch.clamp's backward passes the gradient on the closed interval; the NaN positions were overwritten -> 0
__global__ __launch_bounds__(LT) void clamp01_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g, float* __restrict__ gx,
                                                         long long n) {
  const long long i = (long long)blockIdx.x * LT + threadIdx.x;
  if (i >= n) return;
  const float v = x[i];
  gx[i] = (v == v && v >= 0.f && v <= 1.f) ? g[i] : 0.f;
}

// F.avg_pool2d(x, (2, 2)): floor output size
__global__ __launch_bounds__(LT) void avgpool2_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W, int C) {
  const int OH = H / 2, OW = W / 2;
  const long long i = (long long)blockIdx.x * LT + threadIdx.x;
  if (i >= (long long)B * OH * OW * C) return;
  const int c = (int)(i % C), ox = (int)((i / C) % OW), oy = (int)((i / ((long long)C * OW)) % OH), b = (int)(i / ((long long)C * OW * OH));
  const float* p = x + (((long long)b * H + 2 * oy) * W + 2 * ox) * C + c;
  y[i] = (p[0] + p[C] + p[(long long)W * C] + p[(long long)W * C + C]) * 0.25f;
}

struct SsimGeom {
  int B, H, W, C, ws, OH, OW;
  float g[11];
  float C1, C2;
};

// per output pixel: the five windowed moments (kept for the backward) and the block partial sums of ssim_map and cs_map
__global__ __launch_bounds__(LT) void ssim_fwd_kernel(const float* __restrict__ x, const float* __restrict__ y, SsimGeom q,
                                                      float* __restrict__ mom, double* __restrict__ partial) {
  __shared__ double red[2][LT / 64];
  const long long n = (long long)q.B * q.OH * q.OW * q.C;
  double s_ssim = 0.0, s_cs = 0.0;
  for (long long i = (long long)blockIdx.x * LT + threadIdx.x; i < n; i += (long long)gridDim.x * LT) {
    const int c = (int)(i % q.C), ox = (int)((i / q.C) % q.OW), oy = (int)((i / ((long long)q.C * q.OW)) % q.OH);
    const int b = (int)(i / ((long long)q.C * q.OW * q.OH));
    float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
    for (int dy = 0; dy < q.ws; ++dy)
      for (int dx = 0; dx < q.ws; ++dx) {
        const long long p = (((long long)b * q.H + oy + dy) * q.W + ox + dx) * q.C + c;
        const float w = q.g[dy] * q.g[dx], a = x[p], bb = y[p];
        m1 = fmaf(w, a, m1); m2 = fmaf(w, bb, m2);
        e11 = fmaf(w, a * a, e11); e22 = fmaf(w, bb * bb, e22); e12 = fmaf(w, a * bb, e12);
      }
    float* mo = mom + i * 5;
    mo[0] = m1; mo[1] = m2; mo[2] = e11; mo[3] = e22; mo[4] = e12;
    const float s1 = e11 - m1 * m1, s2 = e22 - m2 * m2, s12 = e12 - m1 * m2;
    const float v1 = 2.f * s12 + q.C2, v2 = s1 + s2 + q.C2;
    s_cs += (double)(v1 / v2);
    s_ssim += (double)(((2.f * m1 * m2 + q.C1) * v1) / ((m1 * m1 + m2 * m2 + q.C1) * v2));
  }
  for (int o = 32; o > 0; o >>= 1) { s_ssim += __shfl_xor(s_ssim, o, 64); s_cs += __shfl_xor(s_cs, o, 64); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s_ssim; red[1][threadIdx.x >> 6] = s_cs; }
  __syncthreads();
  if (threadIdx.x < 2) {
    double t = 0.0;
    for (int w = 0; w < LT / 64; ++w) t += red[threadIdx.x][w];
    partial[(size_t)blockIdx.x * 2 + threadIdx.x] = t;
  }
}

__global__ void ssim_finalize_kernel(const double* __restrict__ partial, int blocks, double n, float* __restrict__ out2) {
  if (threadIdx.x >= 2 || blockIdx.x != 0) return;
  double t = 0.0;
  for (int k = 0; k < blocks; ++k) t += partial[(size_t)k * 2 + threadIdx.x];
  out2[threadIdx.x] = (float)(t / n);   // [0] = mean ssim_map, [1] = mean cs_map
}

// per output pixel: d(alpha * ssim + beta * cs)/n w.r.t. the x-dependent moments (mu1, E[xx], E[xy])
__global__ __launch_bounds__(LT) void ssim_bwd_maps_kernel(const float* __restrict__ mom, long long n, float C1, float C2,
                                                           const float* __restrict__ g_sim_cs, int level, float* __restrict__ gm) {
  const long long i = (long long)blockIdx.x * LT + threadIdx.x;
  if (i >= n) return;
  const float alpha = g_sim_cs[level] / (float)n, beta = g_sim_cs[5 + level] / (float)n;
  const float* mo = mom + i * 5;
  const float m1 = mo[0], m2 = mo[1], e11 = mo[2], e22 = mo[3], e12 = mo[4];
  const float s1 = e11 - m1 * m1, s2 = e22 - m2 * m2, s12 = e12 - m1 * m2;
  const float v1 = 2.f * s12 + C2, v2 = s1 + s2 + C2, a1 = 2.f * m1 * m2 + C1, a2 = m1 * m1 + m2 * m2 + C1;
  const float cs = v1 / v2, lum = a1 / a2, iv2 = 1.f / v2, ia2 = 1.f / a2;
  // d cs
  const float dcs_m1 = (-2.f * m2 - cs * (-2.f * m1)) * iv2;
  const float dcs_e11 = (-cs) * iv2;
  const float dcs_e12 = 2.f * iv2;
  // d lum (only through mu1)
  const float dl_m1 = (2.f * m2 - lum * 2.f * m1) * ia2;
  const float w_cs = alpha * lum + beta;   // ssim = lum * cs
  float* o = gm + i * 3;
  o[0] = alpha * dl_m1 * cs + w_cs * dcs_m1;
  o[1] = w_cs * dcs_e11;
  o[2] = w_cs * dcs_e12;
}

// per input pixel: adjoint of the windowed sums (+ the gradient arriving from the next, 2x-pooled level)
__global__ __launch_bounds__(LT) void ssim_bwd_gather_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                             const float* __restrict__ gm, SsimGeom q,
                                                             const float* __restrict__ g_next, int nH, int nW, float* __restrict__ gx) {
  const long long n = (long long)q.B * q.H * q.W * q.C;
  const long long i = (long long)blockIdx.x * LT + threadIdx.x;
  if (i >= n) return;
  const int c = (int)(i % q.C), px = (int)((i / q.C) % q.W), py = (int)((i / ((long long)q.C * q.W)) % q.H);
  const int b = (int)(i / ((long long)q.C * q.W * q.H));
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
  for (int dy = 0; dy < q.ws; ++dy) {
    const int oy = py - dy;
    if (oy < 0 || oy >= q.OH) continue;
    for (int dx = 0; dx < q.ws; ++dx) {
      const int ox = px - dx;
      if (ox < 0 || ox >= q.OW) continue;
      const float w = q.g[dy] * q.g[dx];
      const float* o = gm + ((((long long)b * q.OH + oy) * q.OW + ox) * q.C + c) * 3;
      a0 = fmaf(w, o[0], a0); a1 = fmaf(w, o[1], a1); a2 = fmaf(w, o[2], a2);
    }
  }
  float r = a0 + 2.f * x[i] * a1 + y[i] * a2;
  if (g_next && (py >> 1) < nH && (px >> 1) < nW) r += 0.25f * g_next[(((long long)b * nH + (py >> 1)) * nW + (px >> 1)) * q.C + c];
  gx[i] = r;
}

// 2x2 max-pool, stride 2 (vgg16.features[4], [9]) on NHWC bf16; backward routes to the FIRST maximum in scan order (ATen)
__global__ __launch_bounds__(LT) void maxpool2_kernel(const a16_t* __restrict__ x, a16_t* __restrict__ y, int B, int H, int W, int C) {
  const int OH = H / 2, OW = W / 2;
  const long long i = (long long)blockIdx.x * LT + threadIdx.x;
  if (i >= (long long)B * OH * OW * C) return;
  const int c = (int)(i % C), ox = (int)((i / C) % OW), oy = (int)((i / ((long long)C * OW)) % OH), b = (int)(i / ((long long)C * OW * OH));
  const a16_t* p = x + (((long long)b * H + 2 * oy) * W + 2 * ox) * C + c;
  const float v = fmaxf(fmaxf(a2f(p[0]), a2f(p[C])), fmaxf(a2f(p[(long long)W * C]), a2f(p[(long long)W * C + C])));
  y[i] = f2a(v);
}
__global__ __launch_bounds__(LT) void maxpool2_bwd_kernel(const a16_t* __restrict__ x, const a16_t* __restrict__ g,
                                                          a16_t* __restrict__ gx, int B, int H, int W, int C) {
  const int OH = H / 2, OW = W / 2;
  const long long i = (long long)blockIdx.x * LT + threadIdx.x;
  if (i >= (long long)B * H * W * C) return;
  const int c = (int)(i % C), px = (int)((i / C) % W), py = (int)((i / ((long long)C * W)) % H), b = (int)(i / ((long long)C * W * H));
  const int oy = py >> 1, ox = px >> 1;
  a16_t r = 0;
  if (oy < OH && ox < OW) {
    const a16_t* p = x + (((long long)b * H + 2 * oy) * W + 2 * ox) * C + c;
    const float v[4] = {a2f(p[0]), a2f(p[C]), a2f(p[(long long)W * C]), a2f(p[(long long)W * C + C])};
    int arg = 0;
    for (int k = 1; k < 4; ++k)
      if (v[k] > v[arg]) arg = k;
    if (arg == (py & 1) * 2 + (px & 1)) r = g[(((long long)b * OH + oy) * OW + ox) * C + c];
  }
  gx[i] = r;
}

// F.mse_loss of two bf16 feature maps: partial sums of (a-b)^2 and, optionally, grad_a = 2 (a-b) / n
__global__ __launch_bounds__(LT) void mse_kernel(const a16_t* __restrict__ a, const a16_t* __restrict__ b, long long n, float inv_n,
                                                 a16_t* __restrict__ ga, double* __restrict__ partial) {
  __shared__ double red[LT / 64];
  double acc = 0.0;
  for (long long i = (long long)blockIdx.x * LT + threadIdx.x; i < n; i += (long long)gridDim.x * LT) {
    const float d = a2f(a[i]) - a2f(b[i]);
    acc += (double)(d * d);
    if (ga) ga[i] = f2a(2.f * d * inv_n);
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < LT / 64; ++w) t += red[w];
    partial[blockIdx.x] = t;
  }
}
// backward of F.mse_loss w.r.t. a: grad_a = round16(2 (a - b) / n * g), g = the upstream scalar gradient ON THE DEVICE (it carries the
// GradScaler's loss scale under fp16: the product is formed in fp32 BEFORE the 16-bit rounding, as autocast's fp32 mse_loss does --
// rounded first, 2 d / n at n ~ 1e6 sits in fp16's subnormal range and loses most of the perceptual gradient)
__global__ __launch_bounds__(LT) void mse_bwd_kernel(const a16_t* __restrict__ a, const a16_t* __restrict__ b, long long n, float inv_n,
                                                     const float* __restrict__ g, a16_t* __restrict__ ga) {
  const float s = 2.f * inv_n * g[0];
  for (long long i = (long long)blockIdx.x * LT + threadIdx.x; i < n; i += (long long)gridDim.x * LT)
    ga[i] = f2a((a2f(a[i]) - a2f(b[i])) * s);
}
__global__ void mse_finalize_kernel(const double* __restrict__ partial, int blocks, double n, float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double t = 0.0;
  for (int k = 0; k < blocks; ++k) t += partial[k];
  out[0] = (float)(t / n);
}

int lblocks(long long n, int cap) {
  long long b = (n + LT - 1) / LT;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

int make_geom(SsimGeom& q, int B, int H, int W, int C, const float* window_host, int ws, float C1, float C2) {
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || ws <= 0 || ws > 11 || ws > H || ws > W || !window_host) return GLARE_ERR_INVALID;
  q.B = B; q.H = H; q.W = W; q.C = C; q.ws = ws; q.OH = H - ws + 1; q.OW = W - ws + 1; q.C1 = C1; q.C2 = C2;
  for (int i = 0; i < 11; ++i) q.g[i] = i < ws ? window_host[i] : 0.f;
  return GLARE_OK;
}

}  // namespace

#define ST(s) static_cast<hipStream_t>(s)

extern "C" int glare_clamp01_f32(const float* x, float* y, long long n, glare_stream_t stream) {
  if (n < 0) return GLARE_ERR_INVALID;
  if (n == 0) return GLARE_OK;
  if (!x || !y) return GLARE_ERR_INVALID;
  hipLaunchKernelGGL(clamp01_kernel, dim3((unsigned)cdivll(n, LT)), dim3(LT), 0, ST(stream), x, y, n);
  return glare_launch_status();
}
extern "C" int glare_clamp01_backward_f32(const float* x, const float* g, float* gx, long long n, glare_stream_t stream) {
  if (n < 0) return GLARE_ERR_INVALID;
  if (n == 0) return GLARE_OK;
  if (!x || !g || !gx) return GLARE_ERR_INVALID;
  hipLaunchKernelGGL(clamp01_bwd_kernel, dim3((unsigned)cdivll(n, LT)), dim3(LT), 0, ST(stream), x, g, gx, n);
  return glare_launch_status();
}
extern "C" int glare_avgpool2_f32(const float* x_nhwc, float* y_nhwc, int B, int H, int W, int C, glare_stream_t stream) {
  if (!x_nhwc || !y_nhwc || B <= 0 || H < 2 || W < 2 || C <= 0) return GLARE_ERR_INVALID;
  const long long n = (long long)B * (H / 2) * (W / 2) * C;
  hipLaunchKernelGGL(avgpool2_kernel, dim3((unsigned)cdivll(n, LT)), dim3(LT), 0, ST(stream), x_nhwc, y_nhwc, B, H, W, C);
  return glare_launch_status();
}

extern "C" size_t glare_ssim_workspace_bytes(void) { return (size_t)256 * 2 * sizeof(double); }

extern "C" int glare_ssim_forward_f32(const float* x_nhwc, const float* y_nhwc, int B, int H, int W, int C, const float* window_host,
                                      int window_size, float C1, float C2, float* moments, float* ssim_cs_out, void* workspace,
                                      size_t workspace_bytes, glare_stream_t stream) {
  SsimGeom q;
  const int rc = make_geom(q, B, H, W, C, window_host, window_size, C1, C2);
  if (rc != GLARE_OK) return rc;
  if (!x_nhwc || !y_nhwc || !moments || !ssim_cs_out) return GLARE_ERR_INVALID;
  if (!workspace || workspace_bytes < glare_ssim_workspace_bytes()) return GLARE_ERR_WORKSPACE;
  const long long n = (long long)B * q.OH * q.OW * C;
  const int blocks = lblocks(n, 256);
  hipLaunchKernelGGL(ssim_fwd_kernel, dim3(blocks), dim3(LT), 0, ST(stream), x_nhwc, y_nhwc, q, moments, static_cast<double*>(workspace));
  hipLaunchKernelGGL(ssim_finalize_kernel, dim3(1), dim3(64), 0, ST(stream), static_cast<const double*>(workspace), blocks, (double)n,
                     ssim_cs_out);
  return glare_launch_status();
}

extern "C" int glare_ssim_backward_f32(const float* x_nhwc, const float* y_nhwc, const float* moments, int B, int H, int W, int C,
                                       const float* window_host, int window_size, float C1, float C2, const float* g_sim_cs_5_5,
                                       int level, const float* g_next_or_null, float* scratch_maps, float* gx, glare_stream_t stream) {
  SsimGeom q;
  const int rc = make_geom(q, B, H, W, C, window_host, window_size, C1, C2);
  if (rc != GLARE_OK) return rc;
  if (!x_nhwc || !y_nhwc || !moments || !g_sim_cs_5_5 || !scratch_maps || !gx || level < 0 || level > 4) return GLARE_ERR_INVALID;
  const long long no = (long long)B * q.OH * q.OW * C, ni = (long long)B * H * W * C;
  hipLaunchKernelGGL(ssim_bwd_maps_kernel, dim3((unsigned)cdivll(no, LT)), dim3(LT), 0, ST(stream), moments, no, C1, C2, g_sim_cs_5_5,
                     level, scratch_maps);
  hipLaunchKernelGGL(ssim_bwd_gather_kernel, dim3((unsigned)cdivll(ni, LT)), dim3(LT), 0, ST(stream), x_nhwc, y_nhwc, scratch_maps, q,
                     g_next_or_null, H / 2, W / 2, gx);
  return glare_launch_status();
}

extern "C" int glare_maxpool2_bf16(const void* x_nhwc, void* y_nhwc, int B, int H, int W, int C, glare_stream_t stream) {
  if (!x_nhwc || !y_nhwc || B <= 0 || H < 2 || W < 2 || C <= 0) return GLARE_ERR_INVALID;
  const long long n = (long long)B * (H / 2) * (W / 2) * C;
  hipLaunchKernelGGL(maxpool2_kernel, dim3((unsigned)cdivll(n, LT)), dim3(LT), 0, ST(stream), static_cast<const a16_t*>(x_nhwc),
                     static_cast<a16_t*>(y_nhwc), B, H, W, C);
  return glare_launch_status();
}
extern "C" int glare_maxpool2_backward_bf16(const void* x_nhwc, const void* g_nhwc, void* gx_nhwc, int B, int H, int W, int C,
                                            glare_stream_t stream) {
  if (!x_nhwc || !g_nhwc || !gx_nhwc || B <= 0 || H < 2 || W < 2 || C <= 0) return GLARE_ERR_INVALID;
  const long long n = (long long)B * H * W * C;
  hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3((unsigned)cdivll(n, LT)), dim3(LT), 0, ST(stream), static_cast<const a16_t*>(x_nhwc),
                     static_cast<const a16_t*>(g_nhwc), static_cast<a16_t*>(gx_nhwc), B, H, W, C);
  return glare_launch_status();
}

extern "C" int glare_mse_loss_bf16(const void* a, const void* b, long long n, float* loss_out, void* grad_a_or_null, void* workspace,
                                   size_t workspace_bytes, glare_stream_t stream) {
  if (n <= 0 || !a || !b || !loss_out) return GLARE_ERR_INVALID;
  if (!workspace || workspace_bytes < 256 * sizeof(double)) return GLARE_ERR_WORKSPACE;
  const int blocks = lblocks(n, 256);
  hipLaunchKernelGGL(mse_kernel, dim3(blocks), dim3(LT), 0, ST(stream), static_cast<const a16_t*>(a), static_cast<const a16_t*>(b), n,
                     1.0f / (float)n, static_cast<a16_t*>(grad_a_or_null), static_cast<double*>(workspace));
  hipLaunchKernelGGL(mse_finalize_kernel, dim3(1), dim3(64), 0, ST(stream), static_cast<const double*>(workspace), blocks, (double)n,
                     loss_out);
  return glare_launch_status();
}

extern "C" int glare_mse_backward_bf16(const void* a, const void* b, long long n, const float* g_dev, void* grad_a, glare_stream_t stream) {
  if (n <= 0 || !a || !b || !g_dev || !grad_a) return GLARE_ERR_INVALID;
  hipLaunchKernelGGL(mse_bwd_kernel, dim3(lblocks(n, 2048)), dim3(LT), 0, ST(stream), static_cast<const a16_t*>(a),
                     static_cast<const a16_t*>(b), n, 1.0f / (float)n, g_dev, static_cast<a16_t*>(grad_a));
  return glare_launch_status();
}
