// Backward of the conditional flow's normal direction (stage-2 objective, row a12): the adjoint of csrc/flow.hip's
// flow_fwd_pre / flow_h1 / flow_fwd_post / flow_nll_reduce.  Reference: autograd through FlowStep.normal_flow
// (FlowStep.py:75-98), CondAffineSeparatedAndCond.forward (FlowAffineCouplingsAblation.py:51-81) and GaussianDiag.logp
// (flow.py:76-95) under `loss.backward()` (LLFlow_model.py:231-236).
//
// The latent's gradient gz (fp32 [pixel][3]) is carried in place from the last step to the first; per coupling step
//   post_bwd : z''[1:] = (z'[1:] + shift) * scale, scale = sigmoid(raw+2)+eps  ->  gz'[1:], g(h4) (bf16, 8-channel
//              records for the MFMA data-gradient conv), including d(sum log scale)
//   (conv backward of the two coupling convs: conv_igemm.hip + gemm.hip)
//   h1_bwd   : adjoint of the 1-channel 3x3 conv on z'[0]: gz'[0] += ..., per-block partials of the [64][9] filter gradient
//   pre_bwd  : z' = (M z + t + shiftFt) * scaleFt  ->  gz, g(hF) (bf16 slice), per-block partials of gM (9) and gt (3)
#include "common.h"

namespace {

constexpr int FB_THREADS = 256;

__device__ __forceinline__ float sigmoid_acc(float x) { return 1.0f / (1.0f + expf(-x)); }

struct AffineParams {
  float M[9];
  float t[3];
};

__global__ __launch_bounds__(FB_THREADS) void flow_nll_bwd_kernel(const float* __restrict__ z, const float* __restrict__ mean,
                                                                  const float* __restrict__ g_logp, long long elems_per_sample,
                                                                  long long total, float* __restrict__ gz, float* __restrict__ gmean) {
  const long long i = (long long)blockIdx.x * FB_THREADS + threadIdx.x;
  if (i >= total) return;
  const float g = g_logp[i / elems_per_sample] * (z[i] - mean[i]);   // d logp / d mean = (z - mean)
  gz[i] = -g;
  gmean[i] = g;
}

__global__ __launch_bounds__(FB_THREADS) void flow_post_bwd_kernel(float* __restrict__ gz, const float* __restrict__ z_pre,
                                                                   const float* __restrict__ h4, const float* __restrict__ g_logdet,
                                                                   long long pix_per_sample, long long npix, float eps,
                                                                   a16_t* __restrict__ gh4) {
  const long long p = (long long)blockIdx.x * FB_THREADS + threadIdx.x;
  if (p >= npix) return;
  const float gl = g_logdet[p / pix_per_sample];
  const f32x4 h = *reinterpret_cast<const f32x4*>(h4 + p * 4);
  float out[4];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const float sg = sigmoid_acc(h[2 * c + 1] + 2.f), s = sg + eps;
    const float g = gz[p * 3 + 1 + c];
    const float gs = g * (z_pre[p * 3 + 1 + c] + h[2 * c]) + gl / s;   // through z'' and through log(scale)
    out[2 * c] = g * s;                       // d / d shift
    out[2 * c + 1] = gs * sg * (1.f - sg);    // d / d raw
    gz[p * 3 + 1 + c] = g * s;
  }
  u32x4 o = {pack_a2(out[0], out[1]), pack_a2(out[2], out[3]), 0u, 0u};
  *reinterpret_cast<u32x4*>(gh4 + p * 8) = o;
}

// g: masked gradient of h1's pre-activation, bf16 [pixel][g_pitch] channels [g_off, g_off+64)
__global__ __launch_bounds__(FB_THREADS) void flow_h1_bwd_kernel(float* __restrict__ gz, const a16_t* __restrict__ g, int g_pitch,
                                                                 int g_off, const float* __restrict__ z_pre,
                                                                 const float* __restrict__ wz, int B, int H, int W,
                                                                 float* __restrict__ gwz_partial) {
  __shared__ float wl[9][64];
  __shared__ float red[FB_THREADS / 64][576];
  for (int i = threadIdx.x; i < 576; i += FB_THREADS) wl[i % 9][i / 9] = wz[i];
  __syncthreads();
  float acc[8][9];
#pragma unroll
  for (int e = 0; e < 8; ++e)
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[e][t] = 0.f;
  const long long total = (long long)B * H * W * 8;
  const int grp = threadIdx.x & 7;
  for (long long base = (long long)blockIdx.x * FB_THREADS; base < total; base += (long long)gridDim.x * FB_THREADS) {
    const long long idx = base + threadIdx.x;
    const bool live = idx < total;
    const long long pix = live ? idx >> 3 : 0;
    const int x = (int)(pix % W), y = (int)((pix / W) % H);
    float dz = 0.f;
    if (live) {
      const u32x4 gs = *reinterpret_cast<const u32x4*>(g + pix * g_pitch + g_off + grp * 8);
      float ge[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) { ge[2 * e] = alo(gs[e]); ge[2 * e + 1] = ahi(gs[e]); }
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int dy = t / 3 - 1, dx = t % 3 - 1;
        // filter gradient: h1[p] saw z0[p + (dy,dx)] through tap t
        if (y + dy >= 0 && y + dy < H && x + dx >= 0 && x + dx < W) {
          const float zv = z_pre[(pix + (long long)dy * W + dx) * 3];
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e][t] = fmaf(ge[e], zv, acc[e][t]);
        }
        // data gradient: z0[p] fed h1[p - (dy,dx)] through tap t
        if (y - dy >= 0 && y - dy < H && x - dx >= 0 && x - dx < W) {
          const u32x4 gn = *reinterpret_cast<const u32x4*>(g + (pix - (long long)dy * W - dx) * g_pitch + g_off + grp * 8);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            dz = fmaf(alo(gn[e]), wl[t][grp * 8 + 2 * e], dz);
            dz = fmaf(ahi(gn[e]), wl[t][grp * 8 + 2 * e + 1], dz);
          }
        }
      }
    }
    dz += __shfl_xor(dz, 1, 64);
    dz += __shfl_xor(dz, 2, 64);
    dz += __shfl_xor(dz, 4, 64);
    if (live && grp == 0) gz[pix * 3] += dz;
  }
  // deterministic: lanes with the same channel group (lane & 7) by a fixed shuffle tree, then the 4 waves in order
#pragma unroll
  for (int e = 0; e < 8; ++e)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      float v = acc[e][t];
      v += __shfl_xor(v, 8, 64);
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if ((threadIdx.x & 63) < 8) red[threadIdx.x >> 6][(grp * 8 + e) * 9 + t] = v;
    }
  __syncthreads();
  for (int i = threadIdx.x; i < 576; i += FB_THREADS)
    gwz_partial[(size_t)blockIdx.x * 576 + i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
}

__global__ __launch_bounds__(FB_THREADS) void flow_pre_bwd_kernel(float* __restrict__ gz, const float* __restrict__ z_in,
                                                                  const float* __restrict__ hF, int f_pitch, int f_off,
                                                                  const float* __restrict__ g_logdet, long long pix_per_sample,
                                                                  long long npix, AffineParams ap, const float* __restrict__ mt_dev,
                                                                  float eps, a16_t* __restrict__ ghF, int gf_pitch, int gf_off,
                                                                  float* __restrict__ partial) {
  __shared__ float red[FB_THREADS / 64][12];
  if (mt_dev) {
#pragma unroll
    for (int i = 0; i < 9; ++i) ap.M[i] = mt_dev[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) ap.t[i] = mt_dev[9 + i];
  }
  float acc[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) acc[i] = 0.f;
  for (long long p = (long long)blockIdx.x * FB_THREADS + threadIdx.x; p < npix; p += (long long)gridDim.x * FB_THREADS) {
    const float gl = g_logdet[p / pix_per_sample];
    const float a[3] = {z_in[p * 3], z_in[p * 3 + 1], z_in[p * 3 + 2]};
    const float* f = hF + p * f_pitch + f_off;
    const f32x4 f0 = *reinterpret_cast<const f32x4*>(f);
    const f32x2 f1 = *reinterpret_cast<const f32x2*>(f + 4);
    const float sh[3] = {f0[0], f0[2], f1[0]}, raw[3] = {f0[1], f0[3], f1[1]};
    float gy[3], gf[6];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float y = fmaf(ap.M[3 * i], a[0], fmaf(ap.M[3 * i + 1], a[1], fmaf(ap.M[3 * i + 2], a[2], ap.t[i])));
      const float sg = sigmoid_acc(raw[i] + 2.f), s = sg + eps;
      const float g = gz[p * 3 + i];
      const float gs = g * (y + sh[i]) + gl / s;
      gy[i] = g * s;
      gf[2 * i] = g * s;
      gf[2 * i + 1] = gs * sg * (1.f - sg);
    }
    u32x4 o = {pack_a2(gf[0], gf[1]), pack_a2(gf[2], gf[3]), pack_a2(gf[4], gf[5]), 0u};
    *reinterpret_cast<u32x4*>(ghF + p * gf_pitch + gf_off) = o;
#pragma unroll
    for (int j = 0; j < 3; ++j) gz[p * 3 + j] = ap.M[j] * gy[0] + ap.M[3 + j] * gy[1] + ap.M[6 + j] * gy[2];   // M^T gy
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[3 * i + j] = fmaf(gy[i], a[j], acc[3 * i + j]);
      acc[9 + i] += gy[i];
    }
  }
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    const float v = wave_sum(acc[i]);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < 12)
    partial[(size_t)blockIdx.x * 12 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

int fb_blocks(long long items, int cap) {
  long long b = (items + FB_THREADS - 1) / FB_THREADS;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

#define ST(s) static_cast<hipStream_t>(s)

extern "C" int glare_flow_bwd_blocks(long long n_pixels) { return fb_blocks(n_pixels * 8, 256); }

extern "C" int glare_flow_nll_backward_f32(const float* z, const float* mean, const float* g_logp_per_sample, int B,
                                           long long pixels_per_sample, float* gz, float* gmean, glare_stream_t stream) {
  if (!z || !mean || !g_logp_per_sample || !gz || !gmean || B <= 0 || pixels_per_sample <= 0) return GLARE_ERR_INVALID;
  const long long total = (long long)B * pixels_per_sample * 3;
  hipLaunchKernelGGL(flow_nll_bwd_kernel, dim3((unsigned)cdivll(total, FB_THREADS)), dim3(FB_THREADS), 0, ST(stream), z, mean,
                     g_logp_per_sample, pixels_per_sample * 3, total, gz, gmean);
  return glare_launch_status();
}

extern "C" int glare_flow_fwd_post_backward_f32(float* gz, const float* z_pre, const float* h4, const float* g_logdet_per_sample, int B,
                                                long long pixels_per_sample, float eps, void* gh4_bf16x8, glare_stream_t stream) {
  if (!gz || !z_pre || !h4 || !g_logdet_per_sample || !gh4_bf16x8 || B <= 0 || pixels_per_sample <= 0) return GLARE_ERR_INVALID;
  const long long npix = (long long)B * pixels_per_sample;
  hipLaunchKernelGGL(flow_post_bwd_kernel, dim3((unsigned)cdivll(npix, FB_THREADS)), dim3(FB_THREADS), 0, ST(stream), gz, z_pre, h4,
                     g_logdet_per_sample, pixels_per_sample, npix, eps, static_cast<a16_t*>(gh4_bf16x8));
  return glare_launch_status();
}

extern "C" int glare_flow_h1_backward_f32(float* gz, const void* gh1_bf16, int g_pitch, int g_off, const float* z_pre,
                                          const float* wz_64x9, int B, int H, int W, float* gwz_partial, glare_stream_t stream) {
  if (!gz || !gh1_bf16 || !z_pre || !wz_64x9 || !gwz_partial || B <= 0 || H <= 0 || W <= 0) return GLARE_ERR_INVALID;
  if ((g_pitch % 8) || (g_off % 8) || g_off + 64 > g_pitch) return GLARE_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(flow_h1_bwd_kernel, dim3(glare_flow_bwd_blocks((long long)B * H * W)), dim3(FB_THREADS), 0, ST(stream), gz,
                     static_cast<const a16_t*>(gh1_bf16), g_pitch, g_off, z_pre, wz_64x9, B, H, W, gwz_partial);
  return glare_launch_status();
}

extern "C" int glare_flow_fwd_pre_backward_f32(float* gz, const float* z_in, const float* hF, int hF_pitch, int hF_off,
                                               const float* g_logdet_per_sample, int B, long long pixels_per_sample,
                                               const float* M_3x3_host, const float* t_3_host, float eps, void* ghF_bf16,
                                               int ghF_pitch, int ghF_off, float* gMt_partial, glare_stream_t stream) {
  if (!gz || !z_in || !hF || !g_logdet_per_sample || !M_3x3_host || !t_3_host || !ghF_bf16 || !gMt_partial || B <= 0 ||
      pixels_per_sample <= 0)
    return GLARE_ERR_INVALID;
  if ((hF_pitch % 4) || (hF_off % 4) || hF_off + 6 > hF_pitch || (ghF_pitch % 8) || (ghF_off % 8) || ghF_off + 8 > ghF_pitch)
    return GLARE_ERR_UNSUPPORTED;
  AffineParams ap;
  for (int i = 0; i < 9; ++i) ap.M[i] = M_3x3_host[i];
  for (int i = 0; i < 3; ++i) ap.t[i] = t_3_host[i];
  const long long npix = (long long)B * pixels_per_sample;
  hipLaunchKernelGGL(flow_pre_bwd_kernel, dim3(glare_flow_bwd_blocks(npix)), dim3(FB_THREADS), 0, ST(stream), gz, z_in, hF, hF_pitch,
                     hF_off, g_logdet_per_sample, pixels_per_sample, npix, ap, (const float*)nullptr, eps,
                     static_cast<a16_t*>(ghF_bf16), ghF_pitch, ghF_off, gMt_partial);
  return glare_launch_status();
}

extern "C" int glare_flow_fwd_pre_backward_dev_f32(float* gz, const float* z_in, const float* hF, int hF_pitch, int hF_off,
                                                   const float* g_logdet_per_sample, int B, long long pixels_per_sample,
                                                   const float* Mt_12_device, float eps, void* ghF_bf16, int ghF_pitch, int ghF_off,
                                                   float* gMt_partial, glare_stream_t stream) {
  if (!gz || !z_in || !hF || !g_logdet_per_sample || !Mt_12_device || !ghF_bf16 || !gMt_partial || B <= 0 || pixels_per_sample <= 0)
    return GLARE_ERR_INVALID;
  if ((hF_pitch % 4) || (hF_off % 4) || hF_off + 6 > hF_pitch || (ghF_pitch % 8) || (ghF_off % 8) || ghF_off + 8 > ghF_pitch)
    return GLARE_ERR_UNSUPPORTED;
  AffineParams ap = {};
  const long long npix = (long long)B * pixels_per_sample;
  hipLaunchKernelGGL(flow_pre_bwd_kernel, dim3(glare_flow_bwd_blocks(npix)), dim3(FB_THREADS), 0, ST(stream), gz, z_in, hF, hF_pitch,
                     hF_off, g_logdet_per_sample, pixels_per_sample, npix, ap, Mt_12_device, eps, static_cast<a16_t*>(ghF_bf16),
                     ghF_pitch, ghF_off, gMt_partial);
  return glare_launch_status();
}
