// Conditional normalizing flow, reverse direction (sampling path), per coupling step.
//
// Reference: FlowUpsamplerNet.decode (FlowUpsamplerNet.py:290-326) -> FlowStep.reverse_flow
// (FlowStep.py:100-119) -> CondAffineSeparatedAndCond.forward(reverse=True)
// (FlowAffineCouplingsAblation.py:83-110), InvertibleConv1x1 (Permutations.py:45-59),
// ActNorm2d (FlowActNorms.py:81-100).  The reference runs ~20 tiny torch ops plus a host-side
// fp64 3x3 inverse and slogdet per step; here the latent state z (3 channels, fp32, token-major
// [pixel][3] -- the layout the codebook search consumes) is touched by two fused kernels per step:
//
//   flow_h1:   h1 = relu( ftA[:, 64s:64s+64] + conv3x3(z[:, 0] -> 64) )        bf16 [pixel][64]
//              ftA is the z-INDEPENDENT part of fAffine's first conv (65 -> 64 splits into a
//              64-channel conditional part, batched for all 24 steps into one MFMA conv before
//              the loop, and this 1-channel part); ActNorm is folded into weights and bias.
//   (two MFMA convs: 1x1 64->64 + relu, 3x3 64->4, csrc/conv_igemm.hip)
//   flow_tail: z[1:] = z[1:]/scale - shift;  z = z/scaleFt - shiftFt;  z = M z + t
//              with scale = sigmoid(h+2)+1e-4 ("cross" split, thops.py:39-47) and (M, t) the
//              host-precomposed (fp64) invconv^-1, actnorm^-1 and any following coupling-free steps.
#include "common.h"

namespace {

constexpr int FL_THREADS = 256;

__device__ __forceinline__ float sigmoid_acc(float x) { return 1.0f / (1.0f + expf(-x)); }

struct TailParams {
  float M[9];
  float t[3];
};

// RAW: the pre-activation sum itself in fp32 (what fAffine[0]'s ActNorm sees at its data-dependent initialisation)
template <bool RAW>
__global__ __launch_bounds__(FL_THREADS) void flow_h1_kernel(const float* __restrict__ z, const float* __restrict__ ftA,
                                                             int a_pitch, int a_off, const float* __restrict__ wz,
                                                             a16_t* __restrict__ h1, float* __restrict__ raw, int B, int H, int W,
                                                             a16_t* __restrict__ h1_lo = nullptr) {
  __shared__ float wl[9][64];
  for (int i = threadIdx.x; i < 576; i += FL_THREADS) wl[i % 9][i / 9] = wz[i];  // wz is [64][9]
  __syncthreads();
  const long long total = (long long)B * H * W * 8;
  for (long long idx = (long long)blockIdx.x * FL_THREADS + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * FL_THREADS) {
    const int g = (int)(idx & 7);
    const long long pix = idx >> 3;
    const int x = (int)(pix % W), y = (int)((pix / W) % H);
    const float* a = ftA + pix * a_pitch + a_off + g * 8;
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(a), a1 = *reinterpret_cast<const f32x4*>(a + 4);
    float acc[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
      float v = 0.f;
      if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = z[(pix + (long long)(t / 3 - 1) * W + (t % 3 - 1)) * 3];
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = fmaf(v, wl[t][g * 8 + e], acc[e]);
    }
    if (RAW) {
      *reinterpret_cast<f32x4*>(raw + pix * 64 + g * 8) = f32x4{acc[0], acc[1], acc[2], acc[3]};
      *reinterpret_cast<f32x4*>(raw + pix * 64 + g * 8 + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
    } else {
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = pack_a2(fmaxf(acc[2 * e], 0.f), fmaxf(acc[2 * e + 1], 0.f));
      *reinterpret_cast<u32x4*>(h1 + pix * 64 + g * 8) = o;
      if (h1_lo) {   // the remainder half: h1 as the operand pair of an fp32-class 1x1 conv (glare_conv_desc.k_wrap)
        u32x4 l;
#pragma unroll
        for (int e = 0; e < 4; ++e) l[e] = pack_a2(fmaxf(acc[2 * e], 0.f) - alo(o[e]), fmaxf(acc[2 * e + 1], 0.f) - ahi(o[e]));
        *reinterpret_cast<u32x4*>(h1_lo + pix * 64 + g * 8) = l;
      }
    }
  }
}

// z = M z + t alone (a coupling-free step, FlowStep.py:83-88 with flow_coupling == "noCoupling")
__global__ __launch_bounds__(FL_THREADS) void flow_affine3_kernel(float* __restrict__ z, long long npix, TailParams tp) {
  for (long long p = (long long)blockIdx.x * FL_THREADS + threadIdx.x; p < npix; p += (long long)gridDim.x * FL_THREADS) {
    const float a0 = z[p * 3], a1 = z[p * 3 + 1], a2 = z[p * 3 + 2];
    z[p * 3] = fmaf(tp.M[0], a0, fmaf(tp.M[1], a1, fmaf(tp.M[2], a2, tp.t[0])));
    z[p * 3 + 1] = fmaf(tp.M[3], a0, fmaf(tp.M[4], a1, fmaf(tp.M[5], a2, tp.t[1])));
    z[p * 3 + 2] = fmaf(tp.M[6], a0, fmaf(tp.M[7], a1, fmaf(tp.M[8], a2, tp.t[2])));
  }
}

// ---- ActNorm data-dependent initialisation (FlowActNorms.py:32-46) -----------------------------------------------------------
//   bias = -mean(x);  logs = log(scale / (sqrt(mean((x + bias)^2)) + 1e-6))      per channel over (B, H, W)
// Two passes of one kernel over x fp32 [P][pitch] (the mean, then the centred second moment: the reference's own two-pass
// form), each block a fixed slice of the pixels, fp64 partials, a final single-block reduction in a fixed order:
// deterministic, no atomics.  C <= 64.
constexpr int AN_MAXC = 64;
__global__ __launch_bounds__(FL_THREADS) void actnorm_partial_kernel(const float* __restrict__ x, int pitch, int off, int C, long long P,
                                                                     const double* __restrict__ mean, double* __restrict__ partial) {
  __shared__ double red[FL_THREADS];
  const int lanes_per_pix = C <= 4 ? 4 : (C <= 16 ? 16 : 64);      // threads along the channel axis
  const int c = threadIdx.x % lanes_per_pix, pl = threadIdx.x / lanes_per_pix, ppi = FL_THREADS / lanes_per_pix;
  const long long per = (P + gridDim.x - 1) / gridDim.x;
  const long long p0 = (long long)blockIdx.x * per, p1 = min(P, p0 + per);
  const double mu = (mean && c < C) ? mean[c] : 0.0;
  double acc = 0.0;
  if (c < C)
    for (long long p = p0 + pl; p < p1; p += ppi) {
      const double v = (double)x[p * pitch + off + c] - mu;
      acc += mean ? v * v : v;
    }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x < lanes_per_pix && threadIdx.x < C) {
    double a = 0.0;
    for (int r = 0; r < ppi; ++r) a += red[r * lanes_per_pix + threadIdx.x];
    partial[(size_t)blockIdx.x * AN_MAXC + threadIdx.x] = a;
  }
}

// stage 0: mean[c] = sum / P.   stage 1: bias = -mean, logs = log(scale / (sqrt(sum / P) + 1e-6))
__global__ void actnorm_final_kernel(const double* __restrict__ partial, int nblocks, int C, long long P, int stage, float scale,
                                     double* __restrict__ mean, float* __restrict__ bias, float* __restrict__ logs) {
  const int c = threadIdx.x;
  if (c >= C) return;
  double a = 0.0;
  for (int b = 0; b < nblocks; ++b) a += partial[(size_t)b * AN_MAXC + c];
  a /= (double)P;
  if (stage == 0) {
    mean[c] = a;
  } else {
    bias[c] = (float)(-mean[c]);
    logs[c] = logf(scale / (sqrtf((float)a) + 1e-6f));
  }
}

__global__ __launch_bounds__(FL_THREADS) void flow_tail_kernel(float* __restrict__ z, const float* __restrict__ h4,
                                                               const float* __restrict__ hF, int f_pitch, int f_off,
                                                               long long npix, TailParams tp, float eps) {
  for (long long p = (long long)blockIdx.x * FL_THREADS + threadIdx.x; p < npix; p += (long long)gridDim.x * FL_THREADS) {
    float z0 = z[p * 3], z1 = z[p * 3 + 1], z2 = z[p * 3 + 2];
    const f32x4 h = *reinterpret_cast<const f32x4*>(h4 + p * 4);
    // self-conditional coupling on z[1:]  (FlowAffineCouplingsAblation.py:86-92)
    z1 = z1 / (sigmoid_acc(h[1] + 2.f) + eps) - h[0];
    z2 = z2 / (sigmoid_acc(h[3] + 2.f) + eps) - h[2];
    // feature-conditional affine on all channels  (:104-108)
    const float* f = hF + p * f_pitch + f_off;
    const f32x4 f0 = *reinterpret_cast<const f32x4*>(f);
    const f32x2 f1 = *reinterpret_cast<const f32x2*>(f + 4);
    z0 = z0 / (sigmoid_acc(f0[1] + 2.f) + eps) - f0[0];
    z1 = z1 / (sigmoid_acc(f0[3] + 2.f) + eps) - f0[2];
    z2 = z2 / (sigmoid_acc(f1[1] + 2.f) + eps) - f1[0];
    // invconv^-1, actnorm^-1 (+ following coupling-free steps), composed on the host
    z[p * 3] = fmaf(tp.M[0], z0, fmaf(tp.M[1], z1, fmaf(tp.M[2], z2, tp.t[0])));
    z[p * 3 + 1] = fmaf(tp.M[3], z0, fmaf(tp.M[4], z1, fmaf(tp.M[5], z2, tp.t[1])));
    z[p * 3 + 2] = fmaf(tp.M[6], z0, fmaf(tp.M[7], z1, fmaf(tp.M[8], z2, tp.t[2])));
  }
}


// ---- normal (training) direction, FlowStep.normal_flow (FlowStep.py:75-98) -------------------------------
// pre : z = M z + t  (host-composed actnorm . invconv of this step and the coupling-free steps before it),
//       then the feature-conditional affine z = (z + shiftFt) * scaleFt      (FlowAffineCouplingsAblation.py:55-59)
// post: z[1:] = (z[1:] + shift) * scale                                      (:74-77)
// Both add sum(log scale) of their pixels to a per-(sample, block) partial; the data-independent
// parts of the log-determinant (actnorm logs, slogdet W) are added on the host in fp64.
__global__ __launch_bounds__(FL_THREADS) void flow_fwd_pre_kernel(const float* zin, float* z, const float* __restrict__ hF, int f_pitch,
                                                                  int f_off, long long pix_per_sample, int blocks_per_sample,
                                                                  TailParams tp, const float* __restrict__ mt_dev, float eps,
                                                                  float* __restrict__ ld_partial) {
  __shared__ float red[FL_THREADS / 64];
  if (mt_dev) {  // the 3x3 + offset read from device memory (wave-uniform loads): no host round trip, graph-capturable
#pragma unroll
    for (int i = 0; i < 9; ++i) tp.M[i] = mt_dev[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) tp.t[i] = mt_dev[9 + i];
  }
  const int b = blockIdx.x / blocks_per_sample, blk = blockIdx.x % blocks_per_sample;
  float acc = 0.f;
  for (long long q = (long long)blk * FL_THREADS + threadIdx.x; q < pix_per_sample; q += (long long)blocks_per_sample * FL_THREADS) {
    const long long p = (long long)b * pix_per_sample + q;
    const float a0 = zin[p * 3], a1 = zin[p * 3 + 1], a2 = zin[p * 3 + 2];   // zin == z: in place (no __restrict__ on the pair)
    float z0 = fmaf(tp.M[0], a0, fmaf(tp.M[1], a1, fmaf(tp.M[2], a2, tp.t[0])));
    float z1 = fmaf(tp.M[3], a0, fmaf(tp.M[4], a1, fmaf(tp.M[5], a2, tp.t[1])));
    float z2 = fmaf(tp.M[6], a0, fmaf(tp.M[7], a1, fmaf(tp.M[8], a2, tp.t[2])));
    const float* f = hF + p * f_pitch + f_off;
    const f32x4 f0 = *reinterpret_cast<const f32x4*>(f);
    const f32x2 f1 = *reinterpret_cast<const f32x2*>(f + 4);
    const float s0 = sigmoid_acc(f0[1] + 2.f) + eps, s1 = sigmoid_acc(f0[3] + 2.f) + eps, s2 = sigmoid_acc(f1[1] + 2.f) + eps;
    z[p * 3] = (z0 + f0[0]) * s0;
    z[p * 3 + 1] = (z1 + f0[2]) * s1;
    z[p * 3 + 2] = (z2 + f1[0]) * s2;
    acc += logf(s0) + logf(s1) + logf(s2);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) ld_partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(FL_THREADS) void flow_fwd_post_kernel(const float* zin, float* z, const float* __restrict__ h4,
                                                                   long long pix_per_sample, int blocks_per_sample, float eps,
                                                                   float* __restrict__ ld_partial) {
  __shared__ float red[FL_THREADS / 64];
  const int b = blockIdx.x / blocks_per_sample, blk = blockIdx.x % blocks_per_sample;
  float acc = 0.f;
  for (long long q = (long long)blk * FL_THREADS + threadIdx.x; q < pix_per_sample; q += (long long)blocks_per_sample * FL_THREADS) {
    const long long p = (long long)b * pix_per_sample + q;
    const f32x4 h = *reinterpret_cast<const f32x4*>(h4 + p * 4);
    const float s1 = sigmoid_acc(h[1] + 2.f) + eps, s2 = sigmoid_acc(h[3] + 2.f) + eps;
    const float c0 = zin[p * 3], c1 = zin[p * 3 + 1], c2 = zin[p * 3 + 2];
    if (zin != z) z[p * 3] = c0;                // out of place: the pass-through channel travels too
    z[p * 3 + 1] = (c1 + h[0]) * s1;
    z[p * 3 + 2] = (c2 + h[2]) * s2;
    acc += logf(s1) + logf(s2);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) ld_partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// Gaussian log-likelihood of the latent, GaussianDiag.logp(mean, logs=0, z) (flow.py:76-95), and the final sum of
// the log-determinant partials: one block per sample.
//   out[b] = { sum over partial rows, sum_p -0.5 * ((z - mean)^2 + log(2 pi)) }
__global__ __launch_bounds__(FL_THREADS) void flow_nll_reduce_kernel(const float* __restrict__ z, const float* __restrict__ mean,
                                                                     const float* __restrict__ ld_partial, int n_rows,
                                                                     int row_len, int blocks_per_sample, long long elems_per_sample,
                                                                     double* __restrict__ out) {
  __shared__ double red[2][FL_THREADS / 64];
  const int b = blockIdx.x;
  double ld = 0.0, lp = 0.0;
  for (int i = threadIdx.x; i < n_rows * blocks_per_sample; i += FL_THREADS) {
    const int row = i / blocks_per_sample, k = i % blocks_per_sample;
    ld += ld_partial[(size_t)row * row_len + b * blocks_per_sample + k];
  }
  const float log2pi = 1.8378770664093453f;
  for (long long i = threadIdx.x; i < elems_per_sample; i += FL_THREADS) {
    const float d = z[(size_t)b * elems_per_sample + i] - mean[(size_t)b * elems_per_sample + i];
    lp += -0.5 * ((double)(d * d) + (double)log2pi);
  }
  for (int o = 32; o > 0; o >>= 1) { ld += __shfl_xor(ld, o, 64); lp += __shfl_xor(lp, o, 64); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = ld; red[1][threadIdx.x >> 6] = lp; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, c = 0.0;
    for (int w = 0; w < FL_THREADS / 64; ++w) { a += red[0][w]; c += red[1][w]; }
    out[2 * b] = a;
    out[2 * b + 1] = c;
  }
}

int fl_blocks(long long items) {
  long long b = (items + FL_THREADS - 1) / FL_THREADS;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace

extern "C" int glare_flow_h1_f32(const float* z_nhwc3, const float* ftA, int ftA_pitch, int ftA_off,
                                 const float* wz_64x9, void* h1_bf16, int B, int H, int W, glare_stream_t stream) {
  if (!z_nhwc3 || !ftA || !wz_64x9 || !h1_bf16 || B <= 0 || H <= 0 || W <= 0) return GLARE_ERR_INVALID;
  if ((ftA_pitch % 4) || (ftA_off % 4) || ftA_off + 64 > ftA_pitch) return GLARE_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(flow_h1_kernel<false>, dim3(fl_blocks((long long)B * H * W * 8)), dim3(FL_THREADS), 0, (hipStream_t)stream,
                     z_nhwc3, ftA, ftA_pitch, ftA_off, wz_64x9, (a16_t*)h1_bf16, (float*)nullptr, B, H, W);
  return glare_launch_status();
}

extern "C" int glare_flow_h1_pair_f32(const float* z_nhwc3, const float* ftA, int ftA_pitch, int ftA_off, const float* wz_64x9,
                                      void* h1_hi, void* h1_lo, int B, int H, int W, glare_stream_t stream) {
  if (!z_nhwc3 || !ftA || !wz_64x9 || !h1_hi || !h1_lo || B <= 0 || H <= 0 || W <= 0) return GLARE_ERR_INVALID;
  if ((ftA_pitch % 4) || (ftA_off % 4) || ftA_off + 64 > ftA_pitch) return GLARE_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(flow_h1_kernel<false>, dim3(fl_blocks((long long)B * H * W * 8)), dim3(FL_THREADS), 0, (hipStream_t)stream,
                     z_nhwc3, ftA, ftA_pitch, ftA_off, wz_64x9, (a16_t*)h1_hi, (float*)nullptr, B, H, W, (a16_t*)h1_lo);
  return glare_launch_status();
}

extern "C" int glare_flow_h1_raw_f32(const float* z_nhwc3, const float* ftA, int ftA_pitch, int ftA_off, const float* wz_64x9,
                                     float* raw_f32, int B, int H, int W, glare_stream_t stream) {
  if (!z_nhwc3 || !ftA || !wz_64x9 || !raw_f32 || B <= 0 || H <= 0 || W <= 0) return GLARE_ERR_INVALID;
  if ((ftA_pitch % 4) || (ftA_off % 4) || ftA_off + 64 > ftA_pitch) return GLARE_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(flow_h1_kernel<true>, dim3(fl_blocks((long long)B * H * W * 8)), dim3(FL_THREADS), 0, (hipStream_t)stream,
                     z_nhwc3, ftA, ftA_pitch, ftA_off, wz_64x9, (a16_t*)nullptr, raw_f32, B, H, W);
  return glare_launch_status();
}

extern "C" int glare_flow_affine3_f32(float* z_nhwc3, long long n_pixels, const float* M_3x3_host, const float* t_3_host,
                                      glare_stream_t stream) {
  if (!z_nhwc3 || !M_3x3_host || !t_3_host || n_pixels <= 0) return GLARE_ERR_INVALID;
  TailParams tp;
  for (int i = 0; i < 9; ++i) tp.M[i] = M_3x3_host[i];
  for (int i = 0; i < 3; ++i) tp.t[i] = t_3_host[i];
  hipLaunchKernelGGL(flow_affine3_kernel, dim3(fl_blocks(n_pixels)), dim3(FL_THREADS), 0, (hipStream_t)stream, z_nhwc3, n_pixels, tp);
  return glare_launch_status();
}

static int an_blocks(long long P) {
  long long b = P / 512;
  return (int)(b < 1 ? 1 : (b > 256 ? 256 : b));
}

extern "C" size_t glare_actnorm_init_workspace_bytes(long long n_pixels) {
  return n_pixels <= 0 ? 0 : ((size_t)an_blocks(n_pixels) * AN_MAXC + AN_MAXC) * sizeof(double);
}

extern "C" int glare_actnorm_init_f32(const float* x, int pitch, int off, int C, long long n_pixels, float scale, float* bias_out,
                                      float* logs_out, void* workspace, size_t workspace_bytes, glare_stream_t stream) {
  if (!x || !bias_out || !logs_out || n_pixels <= 0 || C <= 0 || pitch <= 0 || off < 0 || off + C > pitch) return GLARE_ERR_INVALID;
  if (C > AN_MAXC) return GLARE_ERR_UNSUPPORTED;
  if (!workspace || workspace_bytes < glare_actnorm_init_workspace_bytes(n_pixels)) return GLARE_ERR_WORKSPACE;
  const int nb = an_blocks(n_pixels);
  double* partial = static_cast<double*>(workspace);
  double* mean = partial + (size_t)nb * AN_MAXC;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(actnorm_partial_kernel, dim3(nb), dim3(FL_THREADS), 0, st, x, pitch, off, C, n_pixels, (const double*)nullptr, partial);
  hipLaunchKernelGGL(actnorm_final_kernel, dim3(1), dim3(AN_MAXC), 0, st, partial, nb, C, n_pixels, 0, scale, mean, bias_out, logs_out);
  hipLaunchKernelGGL(actnorm_partial_kernel, dim3(nb), dim3(FL_THREADS), 0, st, x, pitch, off, C, n_pixels, (const double*)mean, partial);
  hipLaunchKernelGGL(actnorm_final_kernel, dim3(1), dim3(AN_MAXC), 0, st, partial, nb, C, n_pixels, 1, scale, mean, bias_out, logs_out);
  return glare_launch_status();
}

extern "C" int glare_flow_tail_f32(float* z_nhwc3, const float* h4, const float* hF, int hF_pitch, int hF_off,
                                   long long n_pixels, const float* M_3x3_host, const float* t_3_host, float eps,
                                   glare_stream_t stream) {
  if (!z_nhwc3 || !h4 || !hF || !M_3x3_host || !t_3_host || n_pixels <= 0) return GLARE_ERR_INVALID;
  if ((hF_pitch % 4) || (hF_off % 4) || hF_off + 6 > hF_pitch) return GLARE_ERR_UNSUPPORTED;
  TailParams tp;
  for (int i = 0; i < 9; ++i) tp.M[i] = M_3x3_host[i];
  for (int i = 0; i < 3; ++i) tp.t[i] = t_3_host[i];
  hipLaunchKernelGGL(flow_tail_kernel, dim3(fl_blocks(n_pixels)), dim3(FL_THREADS), 0, (hipStream_t)stream, z_nhwc3, h4, hF,
                     hF_pitch, hF_off, n_pixels, tp, eps);
  return glare_launch_status();
}

extern "C" int glare_flow_blocks_per_sample(long long pixels_per_sample) {
  long long b = (pixels_per_sample + FL_THREADS - 1) / FL_THREADS;
  return (int)(b < 1 ? 1 : (b > 64 ? 64 : b));
}

extern "C" int glare_flow_fwd_pre_f32(float* z_nhwc3, const float* hF, int hF_pitch, int hF_off, int B,
                                      long long pixels_per_sample, const float* M_3x3_host, const float* t_3_host, float eps,
                                      float* logdet_partial, glare_stream_t stream) {
  if (!z_nhwc3 || !hF || !M_3x3_host || !t_3_host || !logdet_partial || B <= 0 || pixels_per_sample <= 0) return GLARE_ERR_INVALID;
  if ((hF_pitch % 4) || (hF_off % 4) || hF_off + 6 > hF_pitch) return GLARE_ERR_UNSUPPORTED;
  TailParams tp;
  for (int i = 0; i < 9; ++i) tp.M[i] = M_3x3_host[i];
  for (int i = 0; i < 3; ++i) tp.t[i] = t_3_host[i];
  const int bps = glare_flow_blocks_per_sample(pixels_per_sample);
  hipLaunchKernelGGL(flow_fwd_pre_kernel, dim3(B * bps), dim3(FL_THREADS), 0, (hipStream_t)stream, (const float*)z_nhwc3, z_nhwc3, hF,
                     hF_pitch, hF_off, pixels_per_sample, bps, tp, (const float*)nullptr, eps, logdet_partial);
  return glare_launch_status();
}

extern "C" int glare_flow_fwd_pre_dev_io_f32(const float* z_in, float* z_out, const float* hF, int hF_pitch, int hF_off, int B,
                                             long long pixels_per_sample, const float* Mt_12_device, float eps, float* logdet_partial,
                                             glare_stream_t stream) {
  if (!z_in || !z_out || !hF || !Mt_12_device || !logdet_partial || B <= 0 || pixels_per_sample <= 0) return GLARE_ERR_INVALID;
  if ((hF_pitch % 4) || (hF_off % 4) || hF_off + 6 > hF_pitch) return GLARE_ERR_UNSUPPORTED;
  TailParams tp = {};
  const int bps = glare_flow_blocks_per_sample(pixels_per_sample);
  hipLaunchKernelGGL(flow_fwd_pre_kernel, dim3(B * bps), dim3(FL_THREADS), 0, (hipStream_t)stream, z_in, z_out, hF, hF_pitch, hF_off,
                     pixels_per_sample, bps, tp, Mt_12_device, eps, logdet_partial);
  return glare_launch_status();
}

extern "C" int glare_flow_fwd_pre_dev_f32(float* z_nhwc3, const float* hF, int hF_pitch, int hF_off, int B,
                                          long long pixels_per_sample, const float* Mt_12_device, float eps, float* logdet_partial,
                                          glare_stream_t stream) {
  return glare_flow_fwd_pre_dev_io_f32(z_nhwc3, z_nhwc3, hF, hF_pitch, hF_off, B, pixels_per_sample, Mt_12_device, eps, logdet_partial,
                                       stream);
}

extern "C" int glare_flow_fwd_post_io_f32(const float* z_in, float* z_out, const float* h4, int B, long long pixels_per_sample, float eps,
                                          float* logdet_partial, glare_stream_t stream) {
  if (!z_in || !z_out || !h4 || !logdet_partial || B <= 0 || pixels_per_sample <= 0) return GLARE_ERR_INVALID;
  const int bps = glare_flow_blocks_per_sample(pixels_per_sample);
  hipLaunchKernelGGL(flow_fwd_post_kernel, dim3(B * bps), dim3(FL_THREADS), 0, (hipStream_t)stream, z_in, z_out, h4,
                     pixels_per_sample, bps, eps, logdet_partial);
  return glare_launch_status();
}

extern "C" int glare_flow_fwd_post_f32(float* z_nhwc3, const float* h4, int B, long long pixels_per_sample, float eps,
                                       float* logdet_partial, glare_stream_t stream) {
  return glare_flow_fwd_post_io_f32(z_nhwc3, z_nhwc3, h4, B, pixels_per_sample, eps, logdet_partial, stream);
}

extern "C" int glare_flow_nll_reduce_f32(const float* z_nhwc3, const float* mean_nhwc3, const float* logdet_partial,
                                         int n_partial_rows, int B, long long pixels_per_sample, double* out_2_per_sample,
                                         glare_stream_t stream) {
  if (!z_nhwc3 || !mean_nhwc3 || !logdet_partial || !out_2_per_sample || B <= 0 || pixels_per_sample <= 0 || n_partial_rows < 0)
    return GLARE_ERR_INVALID;
  const int bps = glare_flow_blocks_per_sample(pixels_per_sample);
  hipLaunchKernelGGL(flow_nll_reduce_kernel, dim3(B), dim3(FL_THREADS), 0, (hipStream_t)stream, z_nhwc3, mean_nhwc3, logdet_partial,
                     n_partial_rows, B * bps, bps, pixels_per_sample * 3, out_2_per_sample);
  return glare_launch_status();
}
