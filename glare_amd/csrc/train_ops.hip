// HBM-bound producers and element-wise kernels of the training step (rows a12 / a13): what `loss.backward()`
// (LLFlow_model.py:231-236, VQLLFLOWD_model.py:226-229) runs around the MFMA contractions.
//
//   im2col_t / transpose   K-contiguous operands for glare_gemm_nt_bf16 (weight gradients, attention backward)
//   dilate2 / pool2_sum    gradients through Downsample's stride 2 and Upsample's nearest x2
//                          (encoder_decoder.py:49-53,68-75), so that every data gradient is a stride-1 conv
//   act_bwd                g *= act'(y) for the activations the forward conv fuses (ReLU, sigmoid)
//   gn_bwd_*               backward of GroupNorm(32)+swish (encoder_decoder.py:29-35)
//   softmax2_rows, attn_ds row softmax / its backward for the materialised attention backward
//   adam                   torch.optim.Adam step (LLFlow_model.py:110-118) on flat fp32 buffers
#include "common.h"

namespace {

constexpr int TT = 64;  // transpose tile

// ---------------------------------------------------------------------------------------------------------
// colT[row_base + c*KK + tap][p] = x[b, oy*stride + ty - pad, ox*stride + tx - pad, c]   (0 outside the image)
// p = (b*OH + oy)*OW + ox, columns [P, ldp) zero.  x: bf16 NHWC [B][H][W][pitch] (channels [off, off+Ci)),
// optionally read through a nearest x2 upsample.  ones_row >= 0: that row becomes 1 for p < P (bias gradient).
struct Im2colParams {
  const a16_t* x;
  a16_t* col;
  long long ldp, P;
  int B, H, W, pitch, off, Ci, KS, stride, pad, ups, OH, OW, row_base, ones_row, vec_ok;
};

__global__ __launch_bounds__(256) void im2col_t_kernel(const Im2colParams p) {
  __shared__ __attribute__((aligned(16))) a16_t tile[TT][TT + 8];
  const int tid = threadIdx.x;
  const long long p0 = (long long)blockIdx.x * TT;
  const int c0 = blockIdx.y * TT, tap = blockIdx.z, ty = tap / p.KS, tx = tap % p.KS, KK = p.KS * p.KS;
  const int IH = p.ups ? 2 * p.H : p.H, IW = p.ups ? 2 * p.W : p.W;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int pl = pass * 32 + (tid >> 3), ch = (tid & 7) * 8;
    const long long pix = p0 + pl;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (pix < p.P && c0 + ch < p.Ci) {
      const int ox = (int)(pix % p.OW), oy = (int)((pix / p.OW) % p.OH), b = (int)(pix / ((long long)p.OW * p.OH));
      const int iy = oy * p.stride + ty - p.pad, ix = ox * p.stride + tx - p.pad;
      if (iy >= 0 && iy < IH && ix >= 0 && ix < IW) {
        const int sy = p.ups ? iy >> 1 : iy, sx = p.ups ? ix >> 1 : ix;
        const a16_t* src = p.x + (((long long)b * p.H + sy) * p.W + sx) * p.pitch + p.off + c0 + ch;
        if (p.vec_ok && c0 + ch + 8 <= p.Ci) {
          v = *reinterpret_cast<const u32x4*>(src);
        } else {
          a16_t e[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) e[i] = (c0 + ch + i < p.Ci) ? src[i] : (a16_t)0;
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = (uint32_t)e[2 * i] | ((uint32_t)e[2 * i + 1] << 16);
        }
      }
    }
    *reinterpret_cast<u32x4*>(&tile[pl][ch]) = v;
  }
  __syncthreads();
  const int cl = tid >> 2, sg = tid & 3, c = c0 + cl;
  if (c < p.Ci) {
    u32x4 o[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint32_t w = (uint32_t)tile[sg * 16 + 2 * i][cl] | ((uint32_t)tile[sg * 16 + 2 * i + 1][cl] << 16);
      o[i >> 2][i & 3] = w;
    }
    a16_t* dst = p.col + ((long long)p.row_base + (long long)c * KK + tap) * p.ldp + p0 + sg * 16;
    *reinterpret_cast<u32x4*>(dst) = o[0];
    *reinterpret_cast<u32x4*>(dst + 8) = o[1];
  }
  if (p.ones_row >= 0 && blockIdx.y == 0 && tap == 0 && tid < TT)
    p.col[(long long)p.ones_row * p.ldp + p0 + tid] = (p0 + tid < p.P) ? (a16_t)A16_ONE : (a16_t)0;
}

// same matrix from an fp32 tensor addressed by element strides (the NCHW image of conv_in, NHWC fp32 latents)
__global__ __launch_bounds__(256) void im2col_t_f32_kernel(const float* __restrict__ x, long long sb, long long sc, long long sy,
                                                          long long sx, a16_t* __restrict__ col, long long ldp, long long P,
                                                          int H, int W, int Ci, int KS, int pad, int row_base, int ones_row) {
  const long long pix = (long long)blockIdx.x * 256 + threadIdx.x;
  const int row = blockIdx.y, KK = KS * KS;
  if (pix >= ldp) return;
  if (row == Ci * KK) {
    if (ones_row >= 0) col[(long long)ones_row * ldp + pix] = pix < P ? (a16_t)A16_ONE : (a16_t)0;
    return;
  }
  const int c = row / KK, tap = row % KK, ty = tap / KS, tx = tap % KS;
  float v = 0.f;
  if (pix < P) {
    const int ox = (int)(pix % W), oy = (int)((pix / W) % H), b = (int)(pix / ((long long)W * H));
    const int iy = oy + ty - pad, ix = ox + tx - pad;
    if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = x[b * sb + c * sc + iy * sy + ix * sx];
  }
  col[((long long)row_base + row) * ldp + pix] = f2a(v);
}

// out[b][c][r] = in[b][r][c] (bf16), columns r in [R, ld_out) zero
__global__ __launch_bounds__(256) void transpose_kernel(const a16_t* __restrict__ in, long long ld_in, long long sb_in,
                                                        a16_t* __restrict__ out, long long ld_out, long long sb_out, long long R,
                                                        int C, int vec_ok) {
  __shared__ __attribute__((aligned(16))) a16_t tile[TT][TT + 8];
  const int tid = threadIdx.x;
  const long long r0 = (long long)blockIdx.x * TT;
  const int c0 = blockIdx.y * TT;
  in += (long long)blockIdx.z * sb_in;
  out += (long long)blockIdx.z * sb_out;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int rl = pass * 32 + (tid >> 3), ch = (tid & 7) * 8;
    const long long r = r0 + rl;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (r < R && c0 + ch < C) {
      const a16_t* src = in + r * ld_in + c0 + ch;
      if (vec_ok && c0 + ch + 8 <= C) {
        v = *reinterpret_cast<const u32x4*>(src);
      } else {
        a16_t e[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) e[i] = (c0 + ch + i < C) ? src[i] : (a16_t)0;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (uint32_t)e[2 * i] | ((uint32_t)e[2 * i + 1] << 16);
      }
    }
    *reinterpret_cast<u32x4*>(&tile[rl][ch]) = v;
  }
  __syncthreads();
  const int cl = tid >> 2, sg = tid & 3, c = c0 + cl;
  if (c < C) {
    u32x4 o[2];
#pragma unroll
    for (int i = 0; i < 8; ++i)
      o[i >> 2][i & 3] = (uint32_t)tile[sg * 16 + 2 * i][cl] | ((uint32_t)tile[sg * 16 + 2 * i + 1][cl] << 16);
    a16_t* dst = out + (long long)c * ld_out + r0 + sg * 16;
    *reinterpret_cast<u32x4*>(dst) = o[0];
    *reinterpret_cast<u32x4*>(dst + 8) = o[1];
  }
}

// ---------------------------------------------------------------------------------------------------------
// dilate2: out[b][2oy+1][2ox+1][c] = g[b][oy][ox][c], 0 elsewhere (out: [B][2OH][2OW][C]); a pad-1 3x3 conv of it with
// the flipped filter is the data gradient of the (0,1,0,1)-padded stride-2 conv (encoder_decoder.py:71-74)
__global__ __launch_bounds__(256) void dilate2_kernel(const a16_t* __restrict__ g, a16_t* __restrict__ out, int B, int OH, int OW,
                                                      int C8) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long n = (long long)B * 2 * OH * 2 * OW * C8;
  if (i >= n) return;
  const int c = (int)(i % C8);
  long long t = i / C8;
  const int x = (int)(t % (2 * OW));
  t /= 2 * OW;
  const int y = (int)(t % (2 * OH)), b = (int)(t / (2 * OH));
  u32x4 v = {0u, 0u, 0u, 0u};
  if ((x & 1) && (y & 1)) v = reinterpret_cast<const u32x4*>(g)[(((long long)b * OH + (y >> 1)) * OW + (x >> 1)) * C8 + c];
  reinterpret_cast<u32x4*>(out)[i] = v;
}

// pool2_sum: out[b][y][x][c] = sum of the 2x2 block of g[b][2y..][2x..][c]  (gradient of the nearest x2 upsample)
__global__ __launch_bounds__(256) void pool2_sum_kernel(const a16_t* __restrict__ g, a16_t* __restrict__ out, int B, int H, int W,
                                                        int C8) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long n = (long long)B * H * W * C8;
  if (i >= n) return;
  const int c = (int)(i % C8);
  long long t = i / C8;
  const int x = (int)(t % W);
  t /= W;
  const int y = (int)(t % H), b = (int)(t / H);
  const u32x4* gp = reinterpret_cast<const u32x4*>(g);
  const long long base = (((long long)b * 2 * H + 2 * y) * 2 * W + 2 * x) * C8 + c;
  const u32x4 a0 = gp[base], a1 = gp[base + C8], a2 = gp[base + 2LL * W * C8], a3 = gp[base + 2LL * W * C8 + C8];
  u32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e)
    o[e] = pack_a2((alo(a0[e]) + alo(a1[e])) + (alo(a2[e]) + alo(a3[e])), (ahi(a0[e]) + ahi(a1[e])) + (ahi(a2[e]) + ahi(a3[e])));
  reinterpret_cast<u32x4*>(out)[i] = o;
}

// g *= act'(y): ReLU (y > 0) or sigmoid (y (1 - y)); g, y bf16 or fp32 with independent pitches, C channels
template <typename TG, typename TY>
__global__ __launch_bounds__(256) void act_bwd_kernel(TG* __restrict__ g, int g_pitch, int g_off, const TY* __restrict__ y,
                                                      int y_pitch, int y_off, long long pixels, int C, int act) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= pixels * C) return;
  const long long px = i / C;
  const int c = (int)(i % C);
  TG* gp = g + px * g_pitch + g_off + c;
  const TY yv = y[px * y_pitch + y_off + c];
  float gv, yf;
  if constexpr (sizeof(TG) == 2) gv = a2f(*gp); else gv = *gp;
  if constexpr (sizeof(TY) == 2) yf = a2f(yv); else yf = yv;
  gv = act == GLARE_ACT_RELU ? (yf > 0.f ? gv : 0.f) : gv * yf * (1.f - yf);
  if constexpr (sizeof(TG) == 2) *gp = f2a(gv); else *gp = gv;
}

// fp32 <-> bf16 with independent channel pitches / offsets (the gradient of an fp32-output conv enters the MFMA path as bf16)
__global__ __launch_bounds__(256) void cast_f32_to_bf16_kernel(const float* __restrict__ in, int in_pitch, int in_off,
                                                               a16_t* __restrict__ out, int out_pitch, int out_off, long long pixels,
                                                               int C) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= pixels * C) return;
  const long long px = i / C;
  const int c = (int)(i % C);
  out[px * out_pitch + out_off + c] = f2a(in[px * in_pitch + in_off + c]);
}

__global__ __launch_bounds__(256) void cast_bf16_to_f32_kernel(const a16_t* __restrict__ in, int in_pitch, int in_off,
                                                               float* __restrict__ out, int out_pitch, int out_off, long long pixels,
                                                               int C) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= pixels * C) return;
  const long long px = i / C;
  const int c = (int)(i % C);
  out[px * out_pitch + out_off + c] = a2f(in[px * in_pitch + in_off + c]);
}

// ---------------------------------------------------------------------------------------------------------
// GroupNorm(32)+swish backward.  u = xh*gamma + beta, xh = (x-mean)*rstd, y = swish(u) or u.
//   du = dy * swish'(u);  s1[b,c] = sum du, s2[b,c] = sum du*xh;  dbeta = sum_b s1, dgamma = sum_b s2
//   a[b,g] = sum_{c in g} gamma s1 / m, q[b,g] = sum gamma s2 / m  (m = cpg*HW);  dx = rstd (du gamma - a - xh q)
constexpr int GNT = 256, GNG = 32;

__device__ __forceinline__ void gn_group_stats(const float* stats, int b, int splits, int g, long long HW, int cpg, float eps,
                                               float& mean, float& rstd) {
  double s = 0.0, q = 0.0;
  for (int i = 0; i < splits; ++i) {
    s += stats[((size_t)b * splits + i) * GNG * 2 + g * 2];
    q += stats[((size_t)b * splits + i) * GNG * 2 + g * 2 + 1];
  }
  const double n = (double)HW * cpg, m = s / n;
  double var = q / n - m * m;
  if (var < 0.0) var = 0.0;
  mean = (float)m;
  rstd = (float)(1.0 / sqrt(var + (double)eps));
}

__device__ __forceinline__ float swish_grad(float u) {
  const float s = sigmoidf_(u);
  return s * (1.f + u * (1.f - s));
}

// partial[b][split][c][2]
__global__ __launch_bounds__(GNT) void gn_bwd_reduce_kernel(const a16_t* __restrict__ x, int pitch, int off,
                                                            const a16_t* __restrict__ dy, const float* __restrict__ stats,
                                                            int fsplits, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ partial,
                                                            long long HW, int C, float eps, int swish, int splits) {
  __shared__ float mean_s[GNG], rstd_s[GNG];
  __shared__ float red[GNT][17];
  const int b = blockIdx.y, sp = blockIdx.x, cpg = C / GNG;
  if (threadIdx.x < GNG) gn_group_stats(stats, b, fsplits, threadIdx.x, HW, cpg, eps, mean_s[threadIdx.x], rstd_s[threadIdx.x]);
  __syncthreads();
  const int CP = C / 8, ppi = GNT / CP, chunk = threadIdx.x % CP, pl = threadIdx.x / CP;
  float sc[8], sh[8], s1[8], s2[8], mu[8], rs[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = chunk * 8 + e, g = c / cpg;
    mu[e] = mean_s[g]; rs[e] = rstd_s[g];
    sc[e] = gamma[c]; sh[e] = beta[c];
    s1[e] = s2[e] = 0.f;
  }
  const long long per = (HW + splits - 1) / splits, q0 = sp * per, q1 = min(HW, q0 + per);
  const a16_t* xb = x + (size_t)b * HW * pitch + off + chunk * 8;
  const a16_t* gb = dy + (size_t)b * HW * C + chunk * 8;
  auto accum = [&](const u32x4& xv, const u32x4& gv) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float xe = (e & 1) ? ahi(xv[e >> 1]) : alo(xv[e >> 1]);
      float ge = (e & 1) ? ahi(gv[e >> 1]) : alo(gv[e >> 1]);
      const float xh = (xe - mu[e]) * rs[e];
      if (swish) ge *= swish_grad(xh * sc[e] + sh[e]);
      s1[e] += ge;
      s2[e] += ge * xh;
    }
  };
  long long p = q0 + pl;
  for (; p + 3LL * ppi < q1; p += 4LL * ppi) {   // 8 independent 16-B loads in flight per lane; the sums keep the pixel order
    u32x4 xv[4], gv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      xv[k] = *reinterpret_cast<const u32x4*>(xb + (size_t)(p + (long long)k * ppi) * pitch);
      gv[k] = *reinterpret_cast<const u32x4*>(gb + (size_t)(p + (long long)k * ppi) * C);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) accum(xv[k], gv[k]);
  }
  for (; p < q1; p += ppi)
    accum(*reinterpret_cast<const u32x4*>(xb + (size_t)p * pitch), *reinterpret_cast<const u32x4*>(gb + (size_t)p * C));
#pragma unroll
  for (int e = 0; e < 8; ++e) { red[threadIdx.x][e] = s1[e]; red[threadIdx.x][8 + e] = s2[e]; }
  __syncthreads();
  // thread t < C: channel t; sums the ppi rows of its chunk
  for (int c = threadIdx.x; c < C; c += GNT) {
    const int ck = c / 8, e = c % 8;
    float a = 0.f, q = 0.f;
    for (int r = 0; r < ppi; ++r) { a += red[r * CP + ck][e]; q += red[r * CP + ck][8 + e]; }
    float* o = partial + (((size_t)b * splits + sp) * C + c) * 2;
    o[0] = a; o[1] = q;
  }
}

// per (image, block of CB channels): sums[b][c][2] (dbeta / dgamma contributions of image b), coef[b][g][2] = (a, q).
// Thread (cl, j) adds the splits s = j (mod J) of channel cb*CB + cl in ascending order, the J subtotals are combined in order
// j = 0..J-1: a fixed summation tree, whatever the launch.  CB is a multiple of the channels per group.
__global__ __launch_bounds__(GNT) void gn_bwd_finalize_kernel(const float* __restrict__ partial, const float* __restrict__ gamma,
                                                              float* __restrict__ sums, float* __restrict__ coef, long long HW,
                                                              int C, int splits, int CB) {
  __shared__ float sub[GNT][2];
  __shared__ float g1[2048], g2[2048];
  const int b = blockIdx.y, c0 = blockIdx.x * CB, cpg = C / GNG;
  const int J = CB <= GNT ? GNT / CB : 1;
  for (int cbase = 0; cbase < CB; cbase += GNT) {   // one turn unless CB > 256
    const int cl = cbase + (J > 1 ? threadIdx.x % CB : threadIdx.x), j = J > 1 ? threadIdx.x / CB : 0;
    float a = 0.f, q = 0.f;
    if (cl < CB && j < J) {
      for (int sp = j; sp < splits; sp += J) {
        const float2 o = *reinterpret_cast<const float2*>(partial + (((size_t)b * splits + sp) * C + c0 + cl) * 2);
        a += o.x; q += o.y;
      }
    }
    if (J > 1) {
      sub[threadIdx.x][0] = a; sub[threadIdx.x][1] = q;
      __syncthreads();
      if (j == 0 && cl < CB) {
        for (int k = 1; k < J; ++k) { a += sub[k * CB + cl][0]; q += sub[k * CB + cl][1]; }
      }
    }
    if (j == 0 && cl < CB) {
      const int c = c0 + cl;
      sums[((size_t)b * 2 + 0) * C + c] = a;   // [b][0][c] = dbeta part, [b][1][c] = dgamma part
      sums[((size_t)b * 2 + 1) * C + c] = q;
      g1[cl] = a * gamma[c]; g2[cl] = q * gamma[c];
    }
  }
  __syncthreads();
  const int gpb = CB / cpg;   // groups of this block
  if (threadIdx.x < gpb) {
    float a = 0.f, q = 0.f;
    for (int i = 0; i < cpg; ++i) { a += g1[threadIdx.x * cpg + i]; q += g2[threadIdx.x * cpg + i]; }
    const float inv = 1.f / ((float)HW * cpg);
    const int g = c0 / cpg + threadIdx.x;
    coef[((size_t)b * GNG + g) * 2] = a * inv;
    coef[((size_t)b * GNG + g) * 2 + 1] = q * inv;
  }
}

__global__ __launch_bounds__(GNT) void gn_bwd_apply_kernel(const a16_t* __restrict__ x, int pitch, int off,
                                                           const a16_t* __restrict__ dy, const float* __restrict__ stats,
                                                           int fsplits, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ coef,
                                                           a16_t* __restrict__ dx, long long HW, int C, float eps, int swish,
                                                           int blocks_per_image) {
  __shared__ float mean_s[GNG], rstd_s[GNG];
  const int b = blockIdx.x / blocks_per_image, blk = blockIdx.x % blocks_per_image, cpg = C / GNG;
  if (threadIdx.x < GNG) gn_group_stats(stats, b, fsplits, threadIdx.x, HW, cpg, eps, mean_s[threadIdx.x], rstd_s[threadIdx.x]);
  __syncthreads();
  const int CP = C / 8, ppi = GNT / CP, chunk = threadIdx.x % CP, pl = threadIdx.x / CP;
  float sc[8], sh[8], mu[8], rs[8], ca[8], cq[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = chunk * 8 + e, g = c / cpg;
    mu[e] = mean_s[g]; rs[e] = rstd_s[g]; sc[e] = gamma[c]; sh[e] = beta[c];
    ca[e] = coef[((size_t)b * GNG + g) * 2]; cq[e] = coef[((size_t)b * GNG + g) * 2 + 1];
  }
  const long long per = (HW + blocks_per_image - 1) / blocks_per_image, q0 = blk * per, q1 = min(HW, q0 + per);
  const a16_t* xb = x + (size_t)b * HW * pitch + off + chunk * 8;
  const a16_t* gb = dy + (size_t)b * HW * C + chunk * 8;
  a16_t* ob = dx + (size_t)b * HW * C + chunk * 8;
  for (long long p = q0 + pl; p < q1; p += ppi) {
    const u32x4 xv = *reinterpret_cast<const u32x4*>(xb + (size_t)p * pitch);
    const u32x4 gv = *reinterpret_cast<const u32x4*>(gb + (size_t)p * C);
    float r[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float xe = (e & 1) ? ahi(xv[e >> 1]) : alo(xv[e >> 1]);
      float ge = (e & 1) ? ahi(gv[e >> 1]) : alo(gv[e >> 1]);
      const float xh = (xe - mu[e]) * rs[e];
      if (swish) ge *= swish_grad(xh * sc[e] + sh[e]);
      r[e] = rs[e] * (ge * sc[e] - ca[e] - xh * cq[e]);
    }
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack_a2(r[2 * e], r[2 * e + 1]);
    *reinterpret_cast<u32x4*>(ob + (size_t)p * C) = o;
  }
}

// ---------------------------------------------------------------------------------------------------------
// P[i][j] = 2^(S[i][j] - max_j) / sum_j  (S = base-2 logits, fp32 [rows][lds]); bf16 out, columns [n, ldp) zero.
// One workgroup per row.
__global__ __launch_bounds__(256) void softmax2_rows_kernel(const float* __restrict__ S, long long lds, a16_t* __restrict__ P,
                                                            long long ldp, int n) {
  __shared__ float red[4];
  const long long row = blockIdx.x;
  const float* s = S + row * lds;
  float m = -INFINITY;
  for (int j = threadIdx.x; j < n; j += 256) m = fmaxf(m, s[j]);
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int j = threadIdx.x; j < n; j += 256) sum += exp2f(s[j] - m);
  sum = wave_sum(sum);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
  __syncthreads();
  const float inv = 1.f / ((red[0] + red[1]) + (red[2] + red[3]));
  a16_t* p = P + row * ldp;
  for (int j = threadIdx.x; j < ldp; j += 256) p[j] = j < n ? f2a(exp2f(s[j] - m) * inv) : (a16_t)0;
}

// dS[i][j] = scale * P[i][j] * (dP[i][j] - delta_i),  delta_i = sum_c dO[i][c] O[i][c];  bf16 out, pad columns zero
__global__ __launch_bounds__(256) void attn_ds_kernel(const a16_t* __restrict__ P, long long ldp, const float* __restrict__ dP,
                                                      long long lddp, const a16_t* __restrict__ dO, int ld_do,
                                                      const a16_t* __restrict__ O, int ld_o, int d, a16_t* __restrict__ dS,
                                                      long long ldds, int n, float scale) {
  __shared__ float red[4];
  const long long row = blockIdx.x;
  float acc = 0.f;
  for (int c = threadIdx.x; c < d; c += 256) acc += a2f(dO[row * ld_do + c]) * a2f(O[row * ld_o + c]);
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  const float delta = (red[0] + red[1]) + (red[2] + red[3]);
  for (int j = threadIdx.x; j < ldds; j += 256)
    dS[row * ldds + j] = j < n ? f2a(scale * a2f(P[row * ldp + j]) * (dP[row * lddp + j] - delta)) : (a16_t)0;
}

// ---------------------------------------------------------------------------------------------------------
// Mix backward (deformableDecoder_arch.py:587-590): out = s a + (1-s) b, s = sigmoid(w)
//   gb = (1-s) g, ga = s g (optional), partial[blk] = sum g (a - b)   (dw = s (1-s) * sum)
__global__ __launch_bounds__(256) void mix_bwd_kernel(const a16_t* __restrict__ g, const a16_t* __restrict__ a,
                                                      const a16_t* __restrict__ b, a16_t* __restrict__ ga, a16_t* __restrict__ gb,
                                                      long long n8, float s, const float* __restrict__ w_dev,
                                                      float* __restrict__ partial) {
  __shared__ float red[4];
  float post = 1.f;
  if (w_dev) {   // logit on the device: the partial sums leave already scaled by s (1 - s)
    s = 1.0f / (1.0f + expf(-w_dev[0]));
    post = s * (1.f - s);
  }
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
    const u32x4 gv = reinterpret_cast<const u32x4*>(g)[i], av = reinterpret_cast<const u32x4*>(a)[i], bv = reinterpret_cast<const u32x4*>(b)[i];
    u32x4 oa, ob;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float g0 = alo(gv[e]), g1 = ahi(gv[e]);
      acc += g0 * (alo(av[e]) - alo(bv[e])) + g1 * (ahi(av[e]) - ahi(bv[e]));
      oa[e] = pack_a2(s * g0, s * g1);
      ob[e] = pack_a2((1.f - s) * g0, (1.f - s) * g1);
    }
    if (ga) reinterpret_cast<u32x4*>(ga)[i] = oa;
    reinterpret_cast<u32x4*>(gb)[i] = ob;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = ((red[0] + red[1]) + (red[2] + red[3])) * post;
}

// Mean-rescale backward (deformableDecoder_arch.py:567): out = h + xw r, r = sum(h)/sum(xw) over the sample (or batch)
//   D = sum g xw;  gh = g + D/Sx;  gxw = g r - D Sh/Sx^2
// partial[b][blk][3] = (D, Sh, Sx) of a slice; coef[b][3] = (D/Sx, r, D Sh/Sx^2)
__global__ __launch_bounds__(256) void rescale_bwd_reduce_kernel(const a16_t* __restrict__ g, const a16_t* __restrict__ h,
                                                                 const float* __restrict__ xw, long long n_per_sample, int blocks,
                                                                 float* __restrict__ partial) {
  __shared__ float red[3][4];
  const int b = blockIdx.y;
  float d = 0.f, sh = 0.f, sx = 0.f;
  const long long base = (long long)b * n_per_sample;
  if ((n_per_sample & 7) == 0 && (base & 7) == 0) {   // 8 elements per lane and trip: 16-B loads of g / h, 2 x 16 B of xw
    const long long n8 = n_per_sample >> 3;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)blocks * 256) {
      const u32x4 gv = *reinterpret_cast<const u32x4*>(g + base + i * 8), hv = *reinterpret_cast<const u32x4*>(h + base + i * 8);
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(xw + base + i * 8), x1 = *reinterpret_cast<const f32x4*>(xw + base + i * 8 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xa = e < 2 ? x0[2 * e] : x1[2 * e - 4], xb = e < 2 ? x0[2 * e + 1] : x1[2 * e - 3];
        d += alo(gv[e]) * xa + ahi(gv[e]) * xb;
        sh += alo(hv[e]) + ahi(hv[e]);
        sx += xa + xb;
      }
    }
  } else {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_per_sample; i += (long long)blocks * 256) {
      const float x = xw[base + i];
      d += a2f(g[base + i]) * x;
      sh += a2f(h[base + i]);
      sx += x;
    }
  }
  d = wave_sum(d); sh = wave_sum(sh); sx = wave_sum(sx);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = d; red[1][threadIdx.x >> 6] = sh; red[2][threadIdx.x >> 6] = sx; }
  __syncthreads();
  if (threadIdx.x < 3)
    partial[((size_t)b * blocks + blockIdx.x) * 3 + threadIdx.x] =
        (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

// one wave: lane l adds the slices k = l (mod 64) in ascending order, the 64 subtotals are folded by a fixed butterfly
__global__ void rescale_bwd_finalize_kernel(const float* __restrict__ partial, int B, int blocks, int whole_batch,
                                            float* __restrict__ coef) {
  if (threadIdx.x >= 64 || blockIdx.x != 0) return;
  const int lane = threadIdx.x;
  double D = 0, Sh = 0, Sx = 0;
  for (int b = 0; b < B; ++b) {
    double d = 0, sh = 0, sx = 0;
    for (int k = lane; k < blocks; k += 64) {
      d += partial[((size_t)b * blocks + k) * 3];
      sh += partial[((size_t)b * blocks + k) * 3 + 1];
      sx += partial[((size_t)b * blocks + k) * 3 + 2];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      d += __shfl_xor(d, o, 64); sh += __shfl_xor(sh, o, 64); sx += __shfl_xor(sx, o, 64);
    }
    if (whole_batch) {
      D += d; Sh += sh; Sx += sx;
    } else if (lane == 0) {
      coef[b * 3] = (float)(d / sx); coef[b * 3 + 1] = (float)(sh / sx); coef[b * 3 + 2] = (float)(d * sh / (sx * sx));
    }
  }
  if (whole_batch && lane == 0)
    for (int b = 0; b < B; ++b) {
      coef[b * 3] = (float)(D / Sx); coef[b * 3 + 1] = (float)(Sh / Sx); coef[b * 3 + 2] = (float)(D * Sh / (Sx * Sx));
    }
}

__global__ __launch_bounds__(256) void rescale_bwd_apply_kernel(const a16_t* __restrict__ g, const float* __restrict__ coef,
                                                                long long n_per_sample, long long total, a16_t* __restrict__ gh,
                                                                float* __restrict__ gxw) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const float* c = coef + (i / n_per_sample) * 3;
  const float gv = a2f(g[i]);
  gh[i] = f2a(gv + c[0]);
  gxw[i] = gv * c[1] - c[2];
}

__global__ __launch_bounds__(256) void sigmoid_kernel(const float* __restrict__ x, float* __restrict__ y, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) y[i] = 1.0f / (1.0f + expf(-x[i]));
}

// Stage-3 pixel loss (VQLLFLOWD_model.py:209-217): sr = clamp(rec, 0, 1), NaN -> 0 and masked;
//   l1 = mean |sr - gt| over all elements; grad = sign(sr - gt) / n inside the clamp range, 0 outside / at NaN.
// rec NHWC fp32 [B][HW][3]; gt NCHW fp32 (the loader's layout); partial[blk] = sum |.|
__global__ __launch_bounds__(256) void l1_loss_kernel(const float* __restrict__ rec, const float* __restrict__ gt, long long HW, int C,
                                                      long long total, float inv_n, float* __restrict__ grad,
                                                      float* __restrict__ partial) {
  __shared__ float red[4];
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const long long pix = i / C, b = pix / HW, q = pix % HW;
    const float r = rec[i], t = gt[(b * C + c) * HW + q];
    float g = 0.f;
    if (r == r) {   // not NaN
      const float sr = fminf(fmaxf(r, 0.f), 1.f), d = sr - t;
      acc += fabsf(d);
      if (r >= 0.f && r <= 1.f) g = d > 0.f ? inv_n : (d < 0.f ? -inv_n : 0.f);   // torch.clamp: closed interval
    }
    grad[i] = g;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// out = a + b (+ c): the gradient accumulation at a fan-out point of the tape (an activation read by several consumers);
// bf16 in / out, fp32 add.  Keeps the accumulation on this library instead of the autograd engine's own add.
__global__ __launch_bounds__(256) void add_bf16_kernel(const a16_t* __restrict__ a, const a16_t* __restrict__ b,
                                                       const a16_t* __restrict__ c, a16_t* __restrict__ out, long long n8) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  const u32x4 va = reinterpret_cast<const u32x4*>(a)[i], vb = reinterpret_cast<const u32x4*>(b)[i];
  u32x4 vc = {0u, 0u, 0u, 0u};
  if (c) vc = reinterpret_cast<const u32x4*>(c)[i];
  u32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e)
    o[e] = pack_a2(alo(va[e]) + alo(vb[e]) + alo(vc[e]), ahi(va[e]) + ahi(vb[e]) + ahi(vc[e]));
  reinterpret_cast<u32x4*>(out)[i] = o;
}

// partial[blk][c] = sum over this block's pixel slice of g[p][c]  (bias gradient = column sums of the output gradient)
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const a16_t* __restrict__ g, int pitch, long long P, int C, int blocks,
                                                          float* __restrict__ partial) {
  const int c = blockIdx.y * 256 + threadIdx.x;
  if (c >= C) return;
  const long long per = (P + blocks - 1) / blocks, p0 = blockIdx.x * per, p1 = min(P, p0 + per);
  float acc = 0.f;
  for (long long q = p0; q < p1; ++q) acc += a2f(g[q * pitch + c]);
  partial[(size_t)blockIdx.x * C + c] = acc;
}

// torch.optim.Adam (no amsgrad, no weight decay unless wd != 0 -> L2 added to the gradient as torch does)
// step counter and bias corrections kept on the device, so that an optimizer step has no host-side state and the whole
// training step can be replayed from a hipGraph: state = {bc1, sqrt(bc2), lr multiplier}, step_dev = the 1-based step count
__global__ void adam_prepare_kernel(int* __restrict__ step_dev, float* __restrict__ state, float b1, float b2,
                                    const int* __restrict__ skip) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (skip && *skip) return;
  const int t = step_dev[0] + 1;
  step_dev[0] = t;
  state[0] = (float)(1.0 - pow((double)b1, t));
  state[1] = (float)sqrt(1.0 - pow((double)b2, t));
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long long n, float lr, float b1, float b2, float eps,
                                                   float wd, float bc1, float bc2_sqrt, float grad_scale,
                                                   const float* __restrict__ state_dev, const int* __restrict__ skip,
                                                   const float* __restrict__ loss_scale = nullptr) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (skip && *skip) return;   // GradScaler.step: inf / NaN somewhere in this step's gradients (wave-uniform scalar load)
  if (state_dev) { bc1 = state_dev[0]; bc2_sqrt = state_dev[1]; lr *= state_dev[2]; }
  if (loss_scale) grad_scale /= loss_scale[0];   // scaler.unscale_(): the loss was multiplied by the (device-resident) scale
  float gi = g[i] * grad_scale;
  if (wd != 0.f) gi += wd * w[i];
  const float mi = b1 * m[i] + (1.f - b1) * gi;
  const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
  m[i] = mi; v[i] = vi;
  const float denom = sqrtf(vi) / bc2_sqrt + eps;   // torch: (sqrt(v) / sqrt(bias_correction2)) + eps
  w[i] -= (lr / bc1) * (mi / denom);
}

// found |= any(!isfinite(g)): exponent all ones.  One atomicOr per block that saw one (OR: order-independent).
__global__ __launch_bounds__(256) void nonfinite_kernel(const float* __restrict__ g, long long n, int* __restrict__ found) {
  bool bad = false;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    bad |= (__float_as_uint(g[i]) & 0x7f800000u) == 0x7f800000u;
  if (__syncthreads_or(bad) && threadIdx.x == 0) atomicOr(found, 1);
}

__global__ void gradscaler_update_kernel(float* __restrict__ scale, int* __restrict__ tracker, const int* __restrict__ found,
                                         float growth, float backoff, int interval) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (*found) {
    *scale *= backoff;
    *tracker = 0;
  } else {
    const int t = *tracker + 1;
    if (t == interval) { *scale *= growth; *tracker = 0; } else { *tracker = t; }
  }
}

}  // namespace

#define ST(s) static_cast<hipStream_t>(s)

extern "C" int glare_im2col_t_bf16(const void* x_nhwc, int B, int H, int W, int pitch, int off, int Ci, int ksize, int stride,
                                   int pad, int upsample, void* colT, long long ldp, int row_base, int ones_row,
                                   glare_stream_t stream) {
  if (!x_nhwc || !colT || B <= 0 || H <= 0 || W <= 0 || Ci <= 0 || (ksize != 1 && ksize != 3) || (stride != 1 && stride != 2))
    return GLARE_ERR_INVALID;
  if (ldp % TT != 0) return GLARE_ERR_INVALID;
  Im2colParams p;
  const int IH = upsample ? 2 * H : H, IW = upsample ? 2 * W : W;
  p.OH = stride == 1 ? IH + 2 * pad - ksize + 1 : (IH + 1 - ksize) / 2 + 1;   // stride 2: (0,1,0,1) padding
  p.OW = stride == 1 ? IW + 2 * pad - ksize + 1 : (IW + 1 - ksize) / 2 + 1;
  p.P = (long long)B * p.OH * p.OW;
  if (ldp < p.P) return GLARE_ERR_INVALID;
  p.x = static_cast<const a16_t*>(x_nhwc); p.col = static_cast<a16_t*>(colT); p.ldp = ldp;
  p.B = B; p.H = H; p.W = W; p.pitch = pitch; p.off = off; p.Ci = Ci; p.KS = ksize; p.stride = stride; p.pad = pad; p.ups = upsample;
  p.row_base = row_base; p.ones_row = ones_row;
  p.vec_ok = (pitch % 8 == 0 && off % 8 == 0 && (reinterpret_cast<uintptr_t>(x_nhwc) & 15) == 0) ? 1 : 0;
  hipLaunchKernelGGL(im2col_t_kernel, dim3((unsigned)(ldp / TT), cdiv(Ci, TT), ksize * ksize), dim3(256), 0, ST(stream), p);
  return glare_launch_status();
}

extern "C" int glare_im2col_t_f32(const float* x, long long stride_b, long long stride_c, long long stride_y, long long stride_x,
                                  int B, int H, int W, int Ci, int ksize, int pad, void* colT, long long ldp, int row_base,
                                  int ones_row, glare_stream_t stream) {
  if (!x || !colT || B <= 0 || H <= 0 || W <= 0 || Ci <= 0 || (ksize != 1 && ksize != 3)) return GLARE_ERR_INVALID;
  const long long P = (long long)B * H * W;
  if (ldp < P) return GLARE_ERR_INVALID;
  hipLaunchKernelGGL(im2col_t_f32_kernel, dim3((unsigned)cdivll(ldp, 256), Ci * ksize * ksize + (ones_row >= 0 ? 1 : 0)), dim3(256), 0,
                     ST(stream), x, stride_b, stride_c, stride_y, stride_x, static_cast<a16_t*>(colT), ldp, P, H, W, Ci, ksize, pad,
                     row_base, ones_row);
  return glare_launch_status();
}

extern "C" int glare_transpose_bf16(const void* in, long long ld_in, long long batch_stride_in, void* out, long long ld_out,
                                    long long batch_stride_out, long long rows, int cols, int batch, glare_stream_t stream) {
  if (rows < 0 || cols < 0 || batch < 0) return GLARE_ERR_INVALID;
  if (rows == 0 || cols == 0 || batch == 0) return GLARE_OK;
  if (!in || !out || ld_out % TT != 0 || ld_out < rows || ld_in < cols) return GLARE_ERR_INVALID;
  const int vec_ok = (ld_in % 8 == 0 && batch_stride_in % 8 == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0) ? 1 : 0;
  if ((reinterpret_cast<uintptr_t>(out) & 15) != 0 || batch_stride_out % 8 != 0) return GLARE_ERR_INVALID;
  hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)(ld_out / TT), cdiv(cols, TT), batch), dim3(256), 0, ST(stream),
                     static_cast<const a16_t*>(in), ld_in, batch_stride_in, static_cast<a16_t*>(out), ld_out, batch_stride_out, rows,
                     cols, vec_ok);
  return glare_launch_status();
}

extern "C" int glare_dilate2_bf16(const void* g, void* out, int B, int OH, int OW, int C, glare_stream_t stream) {
  if (!g || !out || B <= 0 || OH <= 0 || OW <= 0 || C <= 0 || C % 8) return GLARE_ERR_INVALID;
  const long long n = (long long)B * 4 * OH * OW * (C / 8);
  hipLaunchKernelGGL(dilate2_kernel, dim3((unsigned)cdivll(n, 256)), dim3(256), 0, ST(stream), static_cast<const a16_t*>(g),
                     static_cast<a16_t*>(out), B, OH, OW, C / 8);
  return glare_launch_status();
}

extern "C" int glare_pool2_sum_bf16(const void* g, void* out, int B, int H, int W, int C, glare_stream_t stream) {
  if (!g || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8) return GLARE_ERR_INVALID;
  const long long n = (long long)B * H * W * (C / 8);
  hipLaunchKernelGGL(pool2_sum_kernel, dim3((unsigned)cdivll(n, 256)), dim3(256), 0, ST(stream), static_cast<const a16_t*>(g),
                     static_cast<a16_t*>(out), B, H, W, C / 8);
  return glare_launch_status();
}

extern "C" int glare_act_backward(void* g, int g_is_f32, int g_pitch, int g_off, const void* y, int y_is_f32, int y_pitch, int y_off,
                                  long long pixels, int C, int act, glare_stream_t stream) {
  if (pixels < 0 || C < 0) return GLARE_ERR_INVALID;
  if (pixels == 0 || C == 0) return GLARE_OK;
  if (!g || !y || (act != GLARE_ACT_RELU && act != GLARE_ACT_SIGMOID)) return GLARE_ERR_INVALID;
  const dim3 grid((unsigned)cdivll(pixels * C, 256));
  if (g_is_f32 && y_is_f32)
    hipLaunchKernelGGL((act_bwd_kernel<float, float>), grid, dim3(256), 0, ST(stream), (float*)g, g_pitch, g_off, (const float*)y, y_pitch, y_off, pixels, C, act);
  else if (g_is_f32)
    hipLaunchKernelGGL((act_bwd_kernel<float, a16_t>), grid, dim3(256), 0, ST(stream), (float*)g, g_pitch, g_off, (const a16_t*)y, y_pitch, y_off, pixels, C, act);
  else if (y_is_f32)
    hipLaunchKernelGGL((act_bwd_kernel<a16_t, float>), grid, dim3(256), 0, ST(stream), (a16_t*)g, g_pitch, g_off, (const float*)y, y_pitch, y_off, pixels, C, act);
  else
    hipLaunchKernelGGL((act_bwd_kernel<a16_t, a16_t>), grid, dim3(256), 0, ST(stream), (a16_t*)g, g_pitch, g_off, (const a16_t*)y, y_pitch, y_off, pixels, C, act);
  return glare_launch_status();
}

extern "C" int glare_cast_f32_bf16(const float* in, int in_pitch, int in_off, void* out, int out_pitch, int out_off, long long pixels,
                                   int C, glare_stream_t stream) {
  if (pixels < 0 || C < 0) return GLARE_ERR_INVALID;
  if (pixels == 0 || C == 0) return GLARE_OK;
  if (!in || !out) return GLARE_ERR_INVALID;
  hipLaunchKernelGGL(cast_f32_to_bf16_kernel, dim3((unsigned)cdivll(pixels * C, 256)), dim3(256), 0, ST(stream), in, in_pitch, in_off,
                     static_cast<a16_t*>(out), out_pitch, out_off, pixels, C);
  return glare_launch_status();
}

extern "C" int glare_cast_bf16_f32(const void* in, int in_pitch, int in_off, float* out, int out_pitch, int out_off, long long pixels,
                                   int C, glare_stream_t stream) {
  if (pixels < 0 || C < 0) return GLARE_ERR_INVALID;
  if (pixels == 0 || C == 0) return GLARE_OK;
  if (!in || !out) return GLARE_ERR_INVALID;
  hipLaunchKernelGGL(cast_bf16_to_f32_kernel, dim3((unsigned)cdivll(pixels * C, 256)), dim3(256), 0, ST(stream),
                     static_cast<const a16_t*>(in), in_pitch, in_off, out, out_pitch, out_off, pixels, C);
  return glare_launch_status();
}

static int gn_bwd_splits(long long HW) {
  long long s = HW / 128;   // the per-image finalize walks all splits serially (coalesced over channels)
  return (int)(s < 1 ? 1 : (s > 512 ? 512 : s));
}

extern "C" size_t glare_groupnorm_backward_workspace_bytes(int B, long long HW, int C) {
  if (B <= 0 || HW <= 0 || C <= 0) return 0;
  return ((size_t)B * gn_bwd_splits(HW) * C * 2 + (size_t)B * GNG * 2) * sizeof(float);
}

extern "C" int glare_groupnorm_swish_backward_bf16(const void* x, int in_pitch, int in_off, const void* dy, const float* stats,
                                                   int stat_splits, const float* gamma, const float* beta, void* dx,
                                                   float* dgamma_dbeta_per_image, int B, long long HW, int C, float eps, int swish,
                                                   void* workspace, size_t workspace_bytes, glare_stream_t stream) {
  if (!x || !dy || !stats || !gamma || !beta || !dx || !dgamma_dbeta_per_image || B <= 0 || HW <= 0 || C <= 0 || stat_splits <= 0)
    return GLARE_ERR_INVALID;
  if (C % 32 || C > 2048 || (GNT % (C / 8)) || in_pitch % 8 || in_off % 8) return GLARE_ERR_UNSUPPORTED;
  if (!workspace || workspace_bytes < glare_groupnorm_backward_workspace_bytes(B, HW, C)) return GLARE_ERR_WORKSPACE;
  const int splits = gn_bwd_splits(HW);
  float* partial = static_cast<float*>(workspace);
  float* coef = partial + (size_t)B * splits * C * 2;
  hipLaunchKernelGGL(gn_bwd_reduce_kernel, dim3(splits, B), dim3(GNT), 0, ST(stream), static_cast<const a16_t*>(x), in_pitch, in_off,
                     static_cast<const a16_t*>(dy), stats, stat_splits, gamma, beta, partial, HW, C, eps, swish, splits);
  const int CB = (C % 64 == 0 && 64 % (C / GNG) == 0) ? 64 : C;   // channel block of the finalize: whole groups
  hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(C / CB, B), dim3(GNT), 0, ST(stream), partial, gamma, dgamma_dbeta_per_image, coef, HW,
                     C, splits, CB);
  int bpi = (int)((HW * (C / 8) + 16 * GNT - 1) / (16 * GNT));
  if (bpi < 1) bpi = 1;
  hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3((unsigned)(bpi * B)), dim3(GNT), 0, ST(stream), static_cast<const a16_t*>(x), in_pitch,
                     in_off, static_cast<const a16_t*>(dy), stats, stat_splits, gamma, beta, coef, static_cast<a16_t*>(dx), HW, C, eps,
                     swish, bpi);
  return glare_launch_status();
}

extern "C" int glare_softmax2_rows_f32(const float* S, long long lds, void* P, long long ldp, long long rows, int n,
                                       glare_stream_t stream) {
  if (rows < 0 || n < 0) return GLARE_ERR_INVALID;
  if (rows == 0 || n == 0) return GLARE_OK;
  if (!S || !P || lds < n || ldp < n || rows > 0x7fffffffLL) return GLARE_ERR_INVALID;
  hipLaunchKernelGGL(softmax2_rows_kernel, dim3((unsigned)rows), dim3(256), 0, ST(stream), S, lds, static_cast<a16_t*>(P), ldp, n);
  return glare_launch_status();
}

extern "C" int glare_attention_ds_bf16(const void* P, long long ldp, const float* dP, long long lddp, const void* dO, int ld_do,
                                       const void* O, int ld_o, int d, void* dS, long long ldds, long long rows, int n, float scale,
                                       glare_stream_t stream) {
  if (rows < 0 || n < 0) return GLARE_ERR_INVALID;
  if (rows == 0 || n == 0) return GLARE_OK;
  if (!P || !dP || !dO || !O || !dS || ldp < n || lddp < n || ldds < n || rows > 0x7fffffffLL) return GLARE_ERR_INVALID;
  hipLaunchKernelGGL(attn_ds_kernel, dim3((unsigned)rows), dim3(256), 0, ST(stream), static_cast<const a16_t*>(P), ldp, dP, lddp,
                     static_cast<const a16_t*>(dO), ld_do, static_cast<const a16_t*>(O), ld_o, d, static_cast<a16_t*>(dS), ldds, n,
                     scale);
  return glare_launch_status();
}

extern "C" int glare_mix_backward_bf16(const void* g, const void* a, const void* b, void* ga_or_null, void* gb, long long n, float w,
                                       float* dw_out, void* workspace, size_t workspace_bytes, glare_stream_t stream) {
  if (n < 0 || n % 8) return GLARE_ERR_INVALID;
  if (!g || !a || !b || !gb || !dw_out) return GLARE_ERR_INVALID;
  const int blocks = (int)(cdivll(n / 8, 256) < 1 ? 1 : (cdivll(n / 8, 256) > 512 ? 512 : cdivll(n / 8, 256)));
  if (!workspace || workspace_bytes < (size_t)blocks * sizeof(float)) return GLARE_ERR_WORKSPACE;
  const float s = 1.0f / (1.0f + expf(-w));
  hipLaunchKernelGGL(mix_bwd_kernel, dim3(blocks), dim3(256), 0, ST(stream), static_cast<const a16_t*>(g), static_cast<const a16_t*>(a),
                     static_cast<const a16_t*>(b), static_cast<a16_t*>(ga_or_null), static_cast<a16_t*>(gb), n / 8, s,
                     (const float*)nullptr, static_cast<float*>(workspace));
  return glare_reduce_parts_f32(static_cast<const float*>(workspace), blocks, 1, s * (1.f - s), dw_out, 0, stream);
}

extern "C" int glare_mix_backward_dev_bf16(const void* g, const void* a, const void* b, void* ga_or_null, void* gb, long long n,
                                           const float* w_device, float* dw_out, void* workspace, size_t workspace_bytes,
                                           glare_stream_t stream) {
  if (n < 0 || n % 8) return GLARE_ERR_INVALID;
  if (!g || !a || !b || !gb || !dw_out || !w_device) return GLARE_ERR_INVALID;
  const int blocks = (int)(cdivll(n / 8, 256) < 1 ? 1 : (cdivll(n / 8, 256) > 512 ? 512 : cdivll(n / 8, 256)));
  if (!workspace || workspace_bytes < (size_t)blocks * sizeof(float)) return GLARE_ERR_WORKSPACE;
  hipLaunchKernelGGL(mix_bwd_kernel, dim3(blocks), dim3(256), 0, ST(stream), static_cast<const a16_t*>(g), static_cast<const a16_t*>(a),
                     static_cast<const a16_t*>(b), static_cast<a16_t*>(ga_or_null), static_cast<a16_t*>(gb), n / 8, 0.f, w_device,
                     static_cast<float*>(workspace));
  return glare_reduce_parts_f32(static_cast<const float*>(workspace), blocks, 1, 1.f, dw_out, 0, stream);
}

constexpr int RESCALE_BWD_BLOCKS = 512;   // slices per sample of the mean-rescale backward's sums (fixed: the order of the sums is)

extern "C" size_t glare_mean_rescale_backward_workspace_bytes(int B, long long n_per_sample) {
  if (B <= 0 || n_per_sample <= 0) return 0;
  return ((size_t)B * RESCALE_BWD_BLOCKS * 3 + (size_t)B * 3) * sizeof(float);
}

extern "C" int glare_mean_rescale_backward_bf16(const void* g, const void* h, const float* xw, void* gh, float* gxw, int B,
                                                long long n_per_sample, int whole_batch_mean, void* workspace, size_t workspace_bytes,
                                                glare_stream_t stream) {
  if (!g || !h || !xw || !gh || !gxw || B <= 0 || n_per_sample <= 0) return GLARE_ERR_INVALID;
  if (!workspace || workspace_bytes < glare_mean_rescale_backward_workspace_bytes(B, n_per_sample)) return GLARE_ERR_WORKSPACE;
  float* partial = static_cast<float*>(workspace);
  float* coef = partial + (size_t)B * RESCALE_BWD_BLOCKS * 3;
  hipLaunchKernelGGL(rescale_bwd_reduce_kernel, dim3(RESCALE_BWD_BLOCKS, B), dim3(256), 0, ST(stream), static_cast<const a16_t*>(g),
                     static_cast<const a16_t*>(h), xw, n_per_sample, RESCALE_BWD_BLOCKS, partial);
  hipLaunchKernelGGL(rescale_bwd_finalize_kernel, dim3(1), dim3(64), 0, ST(stream), partial, B, RESCALE_BWD_BLOCKS, whole_batch_mean, coef);
  const long long total = (long long)B * n_per_sample;
  hipLaunchKernelGGL(rescale_bwd_apply_kernel, dim3((unsigned)cdivll(total, 256)), dim3(256), 0, ST(stream), static_cast<const a16_t*>(g),
                     coef, n_per_sample, total, static_cast<a16_t*>(gh), gxw);
  return glare_launch_status();
}

extern "C" int glare_sigmoid_f32(const float* x, float* y, long long n, glare_stream_t stream) {
  if (n < 0) return GLARE_ERR_INVALID;
  if (n == 0) return GLARE_OK;
  if (!x || !y) return GLARE_ERR_INVALID;
  hipLaunchKernelGGL(sigmoid_kernel, dim3((unsigned)cdivll(n, 256)), dim3(256), 0, ST(stream), x, y, n);
  return glare_launch_status();
}

extern "C" int glare_l1_clamp_loss_f32(const float* rec_nhwc, const float* gt_nchw, int B, long long HW, int C, float* loss_out,
                                       float* grad_nhwc, void* workspace, size_t workspace_bytes, glare_stream_t stream) {
  if (!rec_nhwc || !gt_nchw || !loss_out || !grad_nhwc || B <= 0 || HW <= 0 || C <= 0) return GLARE_ERR_INVALID;
  const long long total = (long long)B * HW * C;
  const int blocks = (int)(cdivll(total, 1024) < 1 ? 1 : (cdivll(total, 1024) > 512 ? 512 : cdivll(total, 1024)));
  if (!workspace || workspace_bytes < (size_t)blocks * sizeof(float)) return GLARE_ERR_WORKSPACE;
  const float inv_n = 1.0f / (float)total;
  hipLaunchKernelGGL(l1_loss_kernel, dim3(blocks), dim3(256), 0, ST(stream), rec_nhwc, gt_nchw, HW, C, total, inv_n, grad_nhwc,
                     static_cast<float*>(workspace));
  return glare_reduce_parts_f32(static_cast<const float*>(workspace), blocks, 1, inv_n, loss_out, 0, stream);
}

extern "C" int glare_add_bf16(const void* a, const void* b, const void* c_or_null, void* out, long long n, glare_stream_t stream) {
  if (n < 0 || n % 8) return GLARE_ERR_INVALID;
  if (n == 0) return GLARE_OK;
  if (!a || !b || !out) return GLARE_ERR_INVALID;
  hipLaunchKernelGGL(add_bf16_kernel, dim3((unsigned)cdivll(n / 8, 256)), dim3(256), 0, ST(stream), static_cast<const a16_t*>(a),
                     static_cast<const a16_t*>(b), static_cast<const a16_t*>(c_or_null), static_cast<a16_t*>(out), n / 8);
  return glare_launch_status();
}

extern "C" int glare_colsum_bf16(const void* g, int pitch, long long P, int C, float* out, void* workspace, size_t workspace_bytes,
                                 glare_stream_t stream) {
  if (!g || !out || P <= 0 || C <= 0 || pitch < C) return GLARE_ERR_INVALID;
  const int blocks = (int)(cdivll(P, 512) < 1 ? 1 : (cdivll(P, 512) > 256 ? 256 : cdivll(P, 512)));
  if (!workspace || workspace_bytes < (size_t)blocks * C * sizeof(float)) return GLARE_ERR_WORKSPACE;
  hipLaunchKernelGGL(colsum_bf16_kernel, dim3(blocks, cdiv(C, 256)), dim3(256), 0, ST(stream), static_cast<const a16_t*>(g), pitch, P, C,
                     blocks, static_cast<float*>(workspace));
  return glare_reduce_parts_f32(static_cast<const float*>(workspace), blocks, C, 1.f, out, 0, stream);
}

extern "C" int glare_adam_step_f32(float* w, const float* grad, float* exp_avg, float* exp_avg_sq, long long n, float lr, float beta1,
                                   float beta2, float eps, float weight_decay, int step, float grad_scale, glare_stream_t stream) {
  if (n < 0 || step < 1) return GLARE_ERR_INVALID;
  if (n == 0) return GLARE_OK;
  if (!w || !grad || !exp_avg || !exp_avg_sq) return GLARE_ERR_INVALID;
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)cdivll(n, 256)), dim3(256), 0, ST(stream), w, grad, exp_avg, exp_avg_sq, n, lr, beta1,
                     beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2), grad_scale, (const float*)nullptr, (const int*)nullptr);
  return glare_launch_status();
}

extern "C" int glare_adam_prepare_guarded(int* step_device, float* state3_device, float beta1, float beta2,
                                          const int* skip_if_nonzero_device, glare_stream_t stream) {
  if (!step_device || !state3_device) return GLARE_ERR_INVALID;
  hipLaunchKernelGGL(adam_prepare_kernel, dim3(1), dim3(64), 0, ST(stream), step_device, state3_device, beta1, beta2,
                     skip_if_nonzero_device);
  return glare_launch_status();
}

extern "C" int glare_adam_prepare(int* step_device, float* state3_device, float beta1, float beta2, glare_stream_t stream) {
  return glare_adam_prepare_guarded(step_device, state3_device, beta1, beta2, nullptr, stream);
}

extern "C" int glare_grad_nonfinite_f32(const float* grad, long long n, int* found_device, glare_stream_t stream) {
  if (n < 0 || !found_device) return GLARE_ERR_INVALID;
  if (n == 0) return GLARE_OK;
  if (!grad) return GLARE_ERR_INVALID;
  const long long blocks = cdivll(n, 256 * 16);
  hipLaunchKernelGGL(nonfinite_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, ST(stream), grad, n, found_device);
  return glare_launch_status();
}

extern "C" int glare_gradscaler_update(float* scale_device, int* growth_tracker_device, const int* found_device, float growth_factor,
                                       float backoff_factor, int growth_interval, glare_stream_t stream) {
  if (!scale_device || !growth_tracker_device || !found_device || growth_interval < 1) return GLARE_ERR_INVALID;
  hipLaunchKernelGGL(gradscaler_update_kernel, dim3(1), dim3(64), 0, ST(stream), scale_device, growth_tracker_device, found_device,
                     growth_factor, backoff_factor, growth_interval);
  return glare_launch_status();
}

extern "C" int glare_adam_step_dev_guarded_f32(float* w, const float* grad, float* exp_avg, float* exp_avg_sq, long long n, float lr,
                                               float beta1, float beta2, float eps, float weight_decay, const float* state3_device,
                                               float grad_scale, const int* skip_if_nonzero_device, glare_stream_t stream) {
  if (n < 0) return GLARE_ERR_INVALID;
  if (n == 0) return GLARE_OK;
  if (!w || !grad || !exp_avg || !exp_avg_sq || !state3_device) return GLARE_ERR_INVALID;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)cdivll(n, 256)), dim3(256), 0, ST(stream), w, grad, exp_avg, exp_avg_sq, n, lr, beta1,
                     beta2, eps, weight_decay, 1.f, 1.f, grad_scale, state3_device, skip_if_nonzero_device);
  return glare_launch_status();
}

extern "C" int glare_adam_step_dev_scaled_f32(float* w, const float* grad, float* exp_avg, float* exp_avg_sq, long long n, float lr,
                                              float beta1, float beta2, float eps, float weight_decay, const float* state3_device,
                                              float grad_scale, const float* loss_scale_device, const int* skip_if_nonzero_device,
                                              glare_stream_t stream) {
  if (n < 0) return GLARE_ERR_INVALID;
  if (n == 0) return GLARE_OK;
  if (!w || !grad || !exp_avg || !exp_avg_sq || !state3_device || !loss_scale_device) return GLARE_ERR_INVALID;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)cdivll(n, 256)), dim3(256), 0, ST(stream), w, grad, exp_avg, exp_avg_sq, n, lr, beta1,
                     beta2, eps, weight_decay, 1.f, 1.f, grad_scale, state3_device, skip_if_nonzero_device, loss_scale_device);
  return glare_launch_status();
}

extern "C" int glare_adam_step_dev_f32(float* w, const float* grad, float* exp_avg, float* exp_avg_sq, long long n, float lr, float beta1,
                                       float beta2, float eps, float weight_decay, const float* state3_device, float grad_scale,
                                       glare_stream_t stream) {
  return glare_adam_step_dev_guarded_f32(w, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, state3_device, grad_scale,
                                         nullptr, stream);
}
