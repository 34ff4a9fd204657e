// GroupNorm(32 groups, eps) [+ swish] on NHWC bf16 activations.
//
// Replaces torch.nn.GroupNorm(num_groups=32, eps=1e-6) + x*sigmoid(x)
// (reference: encoder_decoder.py:29-35, used at :119-120,126-127,170 and norm_out :436-437,546-547;
// deformableDecoder_arch.py:573-574).  HBM-bound: one read for the statistics, one read + one
// write for the apply; every access is a 16-B (8-channel) chunk, lanes along the channel axis
// first so a wave reads whole contiguous pixels.
//   pass 1  (gn_stats_kernel):  per (image, split) partial sum / sum of squares per group, fp32
//           per-thread accumulation, LDS reduction, one fp32 pair per (image, split, group).
//   pass 2  (gn_apply_kernel):  combines the partials in fp64, y = (x-mean)*rstd*gamma + beta,
//           optional swish, bf16 store.
#include "common.h"

namespace {

constexpr int GN_THREADS = 256;
constexpr int GN_GROUPS = 32;

// partial layout: [B][splits][32][2] fp32
// ADD: y = x + addend (dense, pitch C) is formed, rounded to bf16, stored to `sum_out`, and the statistics are those of y --
// the residual add that is left of AttnBlock once proj_out is folded into v, fused with the next norm's statistics pass.
// LO: x is the hi half of a hi / lo pair (value = hi + lo, 22 mantissa bits; glare_conv_desc.out_lo), `xlo` its remainder half
template <bool ADD, bool LO = false>
__global__ __launch_bounds__(GN_THREADS) void gn_stats_kernel(const a16_t* __restrict__ x, float* __restrict__ partial,
                                                              long long HW, int C, int pitch, int off, int splits,
                                                              const a16_t* __restrict__ addend, a16_t* __restrict__ sum_out,
                                                              const a16_t* __restrict__ xlo = nullptr) {
  // deterministic block reduction (no float atomics: the statistics, and everything downstream, must not depend on
  // the order in which waves happen to arrive): per-thread sums -> per-channel sums -> per-group sums, fixed order
  __shared__ float vals[GN_THREADS][17];
  __shared__ float chs[2048][2];
  const int b = blockIdx.y, sp = blockIdx.x;
  const int CP = C / 8;                 // 16-B chunks per pixel
  const int ppi = GN_THREADS / CP;      // pixels per iteration (C <= 2048)
  const int chunk = threadIdx.x % CP, pl = threadIdx.x / CP;
  const long long per = (HW + splits - 1) / splits;
  const long long p0 = sp * per, p1 = min(HW, p0 + per);
  float s[8], q[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.f;
  {
    const a16_t* base = x + (size_t)b * HW * pitch + off + chunk * 8;
    const a16_t* abase = ADD ? addend + (size_t)b * HW * C + chunk * 8 : nullptr;
    a16_t* obase = ADD ? sum_out + (size_t)b * HW * C + chunk * 8 : nullptr;
    auto accum = [&](const u32x4& v) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float lo = alo(v[e]), hi = ahi(v[e]);
        s[2 * e] += lo; q[2 * e] += lo * lo;
        s[2 * e + 1] += hi; q[2 * e + 1] += hi * hi;
      }
    };
    long long p = p0 + pl;
    if (ADD) {
      auto add_store = [&](long long q) {
        const u32x4 a = *reinterpret_cast<const u32x4*>(base + (size_t)q * pitch);
        const u32x4 c = *reinterpret_cast<const u32x4*>(abase + (size_t)q * C);
        u32x4 y;
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = pack_a2(alo(a[e]) + alo(c[e]), ahi(a[e]) + ahi(c[e]));
        *reinterpret_cast<u32x4*>(obase + (size_t)q * C) = y;
        return y;
      };
      for (; p + ppi < p1; p += 2LL * ppi) {
        const u32x4 y0 = add_store(p), y1 = add_store(p + ppi);
        accum(y0); accum(y1);
      }
      for (; p < p1; p += ppi) accum(add_store(p));
    } else {
      if constexpr (LO) {
        const a16_t* lbase = xlo + (size_t)b * HW * pitch + off + chunk * 8;
        auto acc2 = [&](const u32x4& vh, const u32x4& vl) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float lo = alo(vh[e]) + alo(vl[e]), hi = ahi(vh[e]) + ahi(vl[e]);
            s[2 * e] += lo; q[2 * e] += lo * lo;
            s[2 * e + 1] += hi; q[2 * e + 1] += hi * hi;
          }
        };
        for (; p + ppi < p1; p += 2LL * ppi) {   // two pixels (four 16-B loads) in flight per lane
          const u32x4 h0 = *reinterpret_cast<const u32x4*>(base + (size_t)p * pitch), l0 = *reinterpret_cast<const u32x4*>(lbase + (size_t)p * pitch);
          const u32x4 h1 = *reinterpret_cast<const u32x4*>(base + (size_t)(p + ppi) * pitch), l1 = *reinterpret_cast<const u32x4*>(lbase + (size_t)(p + ppi) * pitch);
          acc2(h0, l0); acc2(h1, l1);
        }
        for (; p < p1; p += ppi)
          acc2(*reinterpret_cast<const u32x4*>(base + (size_t)p * pitch), *reinterpret_cast<const u32x4*>(lbase + (size_t)p * pitch));
      }
      for (; p + 3LL * ppi < p1; p += 4LL * ppi) {  // 4 independent 16-B loads in flight per lane
        const u32x4 v0 = *reinterpret_cast<const u32x4*>(base + (size_t)p * pitch);
        const u32x4 v1 = *reinterpret_cast<const u32x4*>(base + (size_t)(p + ppi) * pitch);
        const u32x4 v2 = *reinterpret_cast<const u32x4*>(base + (size_t)(p + 2LL * ppi) * pitch);
        const u32x4 v3 = *reinterpret_cast<const u32x4*>(base + (size_t)(p + 3LL * ppi) * pitch);
        accum(v0); accum(v1); accum(v2); accum(v3);
      }
      for (; p < p1; p += ppi) accum(*reinterpret_cast<const u32x4*>(base + (size_t)p * pitch));
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) { vals[threadIdx.x][e] = s[e]; vals[threadIdx.x][8 + e] = q[e]; }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += GN_THREADS) {
    const int ck = c / 8, e = c % 8;
    float a = 0.f, d = 0.f;
    for (int r = 0; r < ppi; ++r) { a += vals[r * CP + ck][e]; d += vals[r * CP + ck][8 + e]; }
    chs[c][0] = a; chs[c][1] = d;
  }
  __syncthreads();
  if (threadIdx.x < GN_GROUPS * 2) {
    const int g = threadIdx.x >> 1, k = threadIdx.x & 1, cpg = C / GN_GROUPS;
    float a = 0.f;
    for (int i = 0; i < cpg; ++i) a += chs[g * cpg + i][k];
    partial[((size_t)b * splits + sp) * GN_GROUPS * 2 + threadIdx.x] = a;
  }
}

// SWISH is a template parameter: as a run-time flag it compiled to one branch per bf16 pair of the streaming loop
template <bool SWISH, bool LO = false>
__global__ __launch_bounds__(GN_THREADS) void gn_apply_kernel(const a16_t* __restrict__ x, const float* __restrict__ partial,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              a16_t* __restrict__ y, long long HW, int C, int pitch, int off,
                                                              int splits, float eps, int blocks_per_image,
                                                              const a16_t* __restrict__ xlo = nullptr, a16_t* __restrict__ ylo = nullptr) {
  __shared__ float mean_s[GN_GROUPS], rstd_s[GN_GROUPS];
  const int b = blockIdx.x / blocks_per_image, blk = blockIdx.x % blocks_per_image;
  const int cpg = C / GN_GROUPS;
  if (threadIdx.x < GN_GROUPS) {
    double s = 0.0, q = 0.0;
    for (int i = 0; i < splits; ++i) {
      s += partial[((size_t)b * splits + i) * GN_GROUPS * 2 + threadIdx.x * 2];
      q += partial[((size_t)b * splits + i) * GN_GROUPS * 2 + threadIdx.x * 2 + 1];
    }
    const double n = (double)HW * cpg;
    const double m = s / n;
    double var = q / n - m * m;   // biased variance, as torch.nn.GroupNorm
    if (var < 0.0) var = 0.0;
    mean_s[threadIdx.x] = (float)m;
    rstd_s[threadIdx.x] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  const int CP = C / 8, ppi = GN_THREADS / CP;
  const int chunk = threadIdx.x % CP, pl = threadIdx.x / CP;
  if (pl >= ppi) return;
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = chunk * 8 + e, g = c / cpg;
    const float a = rstd_s[g] * gamma[c];
    sc[e] = a;
    sh[e] = beta[c] - mean_s[g] * a;
  }
  const long long per = (HW + blocks_per_image - 1) / blocks_per_image;
  const long long p0 = blk * per, p1 = min(HW, p0 + per);
  const a16_t* xb = x + (size_t)b * HW * pitch + off + chunk * 8;
  a16_t* yb = y + (size_t)b * HW * C + chunk * 8;
  auto apply = [&](const u32x4& v) {
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float lo = alo(v[e]) * sc[2 * e] + sh[2 * e];
      float hi = ahi(v[e]) * sc[2 * e + 1] + sh[2 * e + 1];
      if (SWISH) { lo = swishf_(lo); hi = swishf_(hi); }
      o[e] = pack_a2(lo, hi);
    }
    return o;
  };
  long long p = p0 + pl;
  if constexpr (LO) {     // hi / lo input: the normalised value is formed from hi + lo, rounded once on the way out
    const a16_t* lb = xlo + (size_t)b * HW * pitch + off + chunk * 8;
    a16_t* ylb = ylo ? ylo + (size_t)b * HW * C + chunk * 8 : nullptr;   // hi / lo OUTPUT too: the operand pair of an fp32-class conv
    auto apply2 = [&](const u32x4& vh, const u32x4& vl, long long pix) {
      u32x4 o, ol;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float lo = (alo(vh[e]) + alo(vl[e])) * sc[2 * e] + sh[2 * e];
        float hi = (ahi(vh[e]) + ahi(vl[e])) * sc[2 * e + 1] + sh[2 * e + 1];
        if (SWISH) { lo = swishf_(lo); hi = swishf_(hi); }
        o[e] = pack_a2(lo, hi);
        ol[e] = pack_a2(lo - alo(o[e]), hi - ahi(o[e]));
      }
      __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(yb + (size_t)pix * C));
      if (ylb) __builtin_nontemporal_store(ol, reinterpret_cast<u32x4*>(ylb + (size_t)pix * C));
    };
    for (; p + ppi < p1; p += 2LL * ppi) {   // two pixels (four 16-B loads) in flight per lane
      const u32x4 h0 = *reinterpret_cast<const u32x4*>(xb + (size_t)p * pitch), l0 = *reinterpret_cast<const u32x4*>(lb + (size_t)p * pitch);
      const u32x4 h1 = *reinterpret_cast<const u32x4*>(xb + (size_t)(p + ppi) * pitch), l1 = *reinterpret_cast<const u32x4*>(lb + (size_t)(p + ppi) * pitch);
      apply2(h0, l0, p);
      apply2(h1, l1, p + ppi);
    }
    for (; p < p1; p += ppi)
      apply2(*reinterpret_cast<const u32x4*>(xb + (size_t)p * pitch), *reinterpret_cast<const u32x4*>(lb + (size_t)p * pitch), p);
    return;
  }
  // nontemporal stores: +10 % on the apply pass (4.4 -> 4.9 TB/s at 8 x 420x620x128); a load depth of 8 instead of 4 loses 8 %
  constexpr int D = 4;                            // independent 16-B loads in flight per lane
  for (; p + (long long)(D - 1) * ppi < p1; p += (long long)D * ppi) {
    u32x4 v[D];
#pragma unroll
    for (int k = 0; k < D; ++k) v[k] = *reinterpret_cast<const u32x4*>(xb + (size_t)(p + (long long)k * ppi) * pitch);
#pragma unroll
    for (int k = 0; k < D; ++k) {
      __builtin_nontemporal_store(apply(v[k]), reinterpret_cast<u32x4*>(yb + (size_t)(p + (long long)k * ppi) * C));
    }
  }
  for (; p < p1; p += ppi) *reinterpret_cast<u32x4*>(yb + (size_t)p * C) = apply(*reinterpret_cast<const u32x4*>(xb + (size_t)p * pitch));
}

int gn_splits(long long HW) {
  long long s = HW / 128;  // >= 128 pixels per block, at most 64 partials per image for the apply prologue
  return (int)(s < 1 ? 1 : (s > 64 ? 64 : s));
}

// GroupNorm (no activation) in front of 1x1 convs as per-image filters.  With a[b,c] = rstd[b,g(c)] gamma[c] and
// d[b,c] = beta[c] - mean[b,g(c)] a[b,c], hn = a o x + d, so for AttnBlock on hn = GroupNorm(x) (encoder_decoder.py:146-188, after
// the key / value folds of AttnBlock._q_folded / _out_folded):
//   scores   q'_i . hn_j = (a o q'_i) . x_j + const_i  (the constant cancels in softmax_j)   =>  keys   = x
//   q''_i    = a o (Wq hn_i + bq) = (diag(a) Wq diag(a)) x_i + a o (Wq d + bq)
//   output   Wo (sum_j p_ij hn_j) + bo = (Wo diag(a)) (sum_j p_ij x_j) + (Wo d + bo)          =>  values = x
// One block per (output row o, image b) writes row o of both bf16 filters and the two bias entries; the normalised tensor is
// never materialised (11 GroupNorm apply passes of 133 MB per 8-image step).
__global__ __launch_bounds__(256) void attn_fold_kernel(const float* __restrict__ stats, int splits, long long HW, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps, const float* __restrict__ wq,
                                                        const float* __restrict__ bq, const float* __restrict__ wo,
                                                        const float* __restrict__ bo, a16_t* __restrict__ wq_out,
                                                        float* __restrict__ bq_out, a16_t* __restrict__ wo_out,
                                                        float* __restrict__ bo_out, int C, int feedback) {
  __shared__ float a_s[2048], d_s[2048];
  __shared__ float vq_s[2048], vo_s[2048];     // the two filter rows in fp32, rounded afterwards with error feedback (feedback != 0)
  __shared__ float red[2][4];
  const int o = blockIdx.x, b = blockIdx.y, cpg = C / GN_GROUPS;
  for (int c = threadIdx.x; c < C; c += 256) {
    const int g = c / cpg;
    double s = 0.0, q = 0.0;
    for (int i = 0; i < splits; ++i) {
      s += stats[((size_t)b * splits + i) * GN_GROUPS * 2 + g * 2];
      q += stats[((size_t)b * splits + i) * GN_GROUPS * 2 + g * 2 + 1];
    }
    const double n = (double)HW * cpg, m = s / n;
    double var = q / n - m * m;
    if (var < 0.0) var = 0.0;
    const float mean = (float)m, rstd = (float)(1.0 / sqrt(var + (double)eps));   // the two floats gn_apply_kernel uses
    const float a = rstd * gamma[c];
    a_s[c] = a;
    d_s[c] = beta[c] - mean * a;
  }
  __syncthreads();
  const float ao = a_s[o];
  float dq = 0.f, dv = 0.f;
  const size_t row = ((size_t)b * C + o) * C;
  for (int c = threadIdx.x; c < C; c += 256) {
    const float wqv = wq[(size_t)o * C + c], wov = wo[(size_t)o * C + c];
    if (feedback) {
      vq_s[c] = ao * wqv * a_s[c];
      vo_s[c] = wov * a_s[c];
    } else {
      wq_out[row + c] = f2a(ao * wqv * a_s[c]);
      wo_out[row + c] = f2a(wov * a_s[c]);
    }
    dq = fmaf(wqv, d_s[c], dq);
    dv = fmaf(wov, d_s[c], dv);
  }
  if (feedback) {
    // round 6: the per-image filters rounded like every other single-pass filter of inference (conv_igemm.hip filter_feedback_kernel):
    // q_c = round16(v_c + carry) along the input channels of this output row, one lane per filter (512 dependent steps, ~7 us per block)
    __syncthreads();
    if (threadIdx.x == 0 || threadIdx.x == 64) {
      const float* v = threadIdx.x == 0 ? vq_s : vo_s;
      a16_t* dst = (threadIdx.x == 0 ? wq_out : wo_out) + row;
      double carry = 0.0;
      for (int c = 0; c < C; ++c) {
        const double t = (double)v[c] + carry;
        const a16_t q = f2a((float)t);
        dst[c] = q;
        carry = t - (double)a2f(q);
      }
    }
  }
  dq = wave_sum(dq);
  dv = wave_sum(dv);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = dq; red[1][threadIdx.x >> 6] = dv; }
  __syncthreads();
  if (threadIdx.x == 0) {
    bq_out[(size_t)b * C + o] = ao * (((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) + bq[o]);
    bo_out[(size_t)b * C + o] = ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) + bo[o];
  }
}

}  // namespace

extern "C" size_t glare_groupnorm_workspace_bytes(int B, long long HW) {
  if (B <= 0 || HW <= 0) return 0;
  return (size_t)B * gn_splits(HW) * GN_GROUPS * 2 * sizeof(float);
}

extern "C" int glare_groupnorm_swish_bf16(const void* x, int in_pitch, int in_off, const float* gamma, const float* beta,
                                          void* y, int B, long long HW, int C, float eps, int swish, void* workspace,
                                          size_t workspace_bytes, glare_stream_t stream_) {
  if (!x || !gamma || !beta || !y || B <= 0 || HW <= 0 || C <= 0) return GLARE_ERR_INVALID;
  // 32 groups (encoder_decoder.py:35); 16-B chunks; a pixel must fit one 256-thread pass
  if (C % 32 || C % 8 || C > 2048 || (GN_THREADS % (C / 8)) || in_pitch % 8 || in_off % 8) return GLARE_ERR_UNSUPPORTED;
  if (in_off + C > in_pitch) return GLARE_ERR_INVALID;
  if (!workspace || workspace_bytes < glare_groupnorm_workspace_bytes(B, HW)) return GLARE_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  const int splits = gn_splits(HW);
  hipLaunchKernelGGL(gn_stats_kernel<false>, dim3(splits, B), dim3(GN_THREADS), 0, stream, (const a16_t*)x, (float*)workspace,
                     HW, C, in_pitch, in_off, splits, (const a16_t*)nullptr, (a16_t*)nullptr);
  int bpi = (int)((HW * (C / 8) + 16 * GN_THREADS - 1) / (16 * GN_THREADS));  // ~16 chunks per thread
  if (bpi < 1) bpi = 1;
  if (swish)
    hipLaunchKernelGGL(gn_apply_kernel<true>, dim3((unsigned)(bpi * B)), dim3(GN_THREADS), 0, stream, (const a16_t*)x,
                       (const float*)workspace, gamma, beta, (a16_t*)y, HW, C, in_pitch, in_off, splits, eps, bpi);
  else
    hipLaunchKernelGGL(gn_apply_kernel<false>, dim3((unsigned)(bpi * B)), dim3(GN_THREADS), 0, stream, (const a16_t*)x,
                       (const float*)workspace, gamma, beta, (a16_t*)y, HW, C, in_pitch, in_off, splits, eps, bpi);
  return glare_launch_status();
}

extern "C" int glare_add_groupnorm_stats_bf16(const void* a, const void* b, void* out, int B, long long HW, int C, void* stats,
                                              size_t stats_bytes, glare_stream_t stream_) {
  if (!a || !b || !out || !stats || B <= 0 || HW <= 0 || C <= 0) return GLARE_ERR_INVALID;
  if (C % 32 || C > 2048 || (GN_THREADS % (C / 8))) return GLARE_ERR_UNSUPPORTED;
  if (stats_bytes < glare_groupnorm_workspace_bytes(B, HW)) return GLARE_ERR_WORKSPACE;
  const int splits = gn_splits(HW);
  hipLaunchKernelGGL(gn_stats_kernel<true>, dim3(splits, B), dim3(GN_THREADS), 0, (hipStream_t)stream_, (const a16_t*)a,
                     (float*)stats, HW, C, C, 0, splits, (const a16_t*)b, (a16_t*)out);
  return glare_launch_status();
}

extern "C" int glare_groupnorm_apply_bf16(const void* x, int in_pitch, int in_off, const float* gamma, const float* beta,
                                          void* y, int B, long long HW, int C, float eps, int swish, const float* stats,
                                          int splits, glare_stream_t stream_) {
  if (!x || !gamma || !beta || !y || !stats || B <= 0 || HW <= 0 || C <= 0 || splits <= 0) return GLARE_ERR_INVALID;
  if (C % 32 || C % 8 || C > 2048 || (GN_THREADS % (C / 8)) || in_pitch % 8 || in_off % 8) return GLARE_ERR_UNSUPPORTED;
  if (in_off + C > in_pitch) return GLARE_ERR_INVALID;
  int bpi = (int)((HW * (C / 8) + 16 * GN_THREADS - 1) / (16 * GN_THREADS));
  if (bpi < 1) bpi = 1;
  if (swish)
    hipLaunchKernelGGL(gn_apply_kernel<true>, dim3((unsigned)(bpi * B)), dim3(GN_THREADS), 0, (hipStream_t)stream_, (const a16_t*)x,
                       stats, gamma, beta, (a16_t*)y, HW, C, in_pitch, in_off, splits, eps, bpi);
  else
    hipLaunchKernelGGL(gn_apply_kernel<false>, dim3((unsigned)(bpi * B)), dim3(GN_THREADS), 0, (hipStream_t)stream_, (const a16_t*)x,
                       stats, gamma, beta, (a16_t*)y, HW, C, in_pitch, in_off, splits, eps, bpi);
  return glare_launch_status();
}

// GroupNorm of a hi / lo pair (value = x_hi + x_lo, same pitch / offset; glare_conv_desc.out_lo): statistics and normalisation from
// the 22-bit value, one rounding on the way out.  stats != NULL: apply only (statistics from the producer's epilogue); stats == NULL:
// statistics pass first (workspace as glare_groupnorm_swish_bf16).
extern "C" int glare_groupnorm_hilo_bf16(const void* x_hi, const void* x_lo, int in_pitch, int in_off, const float* gamma,
                                         const float* beta, void* y, int B, long long HW, int C, float eps, int swish, const float* stats,
                                         int splits, void* workspace, size_t workspace_bytes, glare_stream_t stream_) {
  return glare_groupnorm_hilo_pair_bf16(x_hi, x_lo, in_pitch, in_off, gamma, beta, y, nullptr, B, HW, C, eps, swish, stats, splits, workspace,
                                        workspace_bytes, stream_);
}

// ... with the OUTPUT as a hi / lo pair as well (y_lo != NULL: dense [B][HW][C] like y): the activation operand pair of an fp32-class
// conv (glare_conv_desc.k_wrap).
extern "C" int glare_groupnorm_hilo_pair_bf16(const void* x_hi, const void* x_lo, int in_pitch, int in_off, const float* gamma,
                                              const float* beta, void* y, void* y_lo, int B, long long HW, int C, float eps, int swish,
                                              const float* stats, int splits, void* workspace, size_t workspace_bytes,
                                              glare_stream_t stream_) {
  if (!x_hi || !x_lo || !gamma || !beta || !y || B <= 0 || HW <= 0 || C <= 0) return GLARE_ERR_INVALID;
  if (C % 32 || C % 8 || C > 2048 || (GN_THREADS % (C / 8)) || in_pitch % 8 || in_off % 8) return GLARE_ERR_UNSUPPORTED;
  if (in_off + C > in_pitch) return GLARE_ERR_INVALID;
  hipStream_t stream = (hipStream_t)stream_;
  if (!stats) {
    if (!workspace || workspace_bytes < glare_groupnorm_workspace_bytes(B, HW)) return GLARE_ERR_WORKSPACE;
    splits = gn_splits(HW);
    hipLaunchKernelGGL((gn_stats_kernel<false, true>), dim3(splits, B), dim3(GN_THREADS), 0, stream, (const a16_t*)x_hi, (float*)workspace,
                       HW, C, in_pitch, in_off, splits, (const a16_t*)nullptr, (a16_t*)nullptr, (const a16_t*)x_lo);
    stats = (const float*)workspace;
  } else if (splits <= 0) {
    return GLARE_ERR_INVALID;
  }
  int bpi = (int)((HW * (C / 8) + 16 * GN_THREADS - 1) / (16 * GN_THREADS));
  if (bpi < 1) bpi = 1;
  if (swish)
    hipLaunchKernelGGL((gn_apply_kernel<true, true>), dim3((unsigned)(bpi * B)), dim3(GN_THREADS), 0, stream, (const a16_t*)x_hi, stats,
                       gamma, beta, (a16_t*)y, HW, C, in_pitch, in_off, splits, eps, bpi, (const a16_t*)x_lo, (a16_t*)y_lo);
  else
    hipLaunchKernelGGL((gn_apply_kernel<false, true>), dim3((unsigned)(bpi * B)), dim3(GN_THREADS), 0, stream, (const a16_t*)x_hi, stats,
                       gamma, beta, (a16_t*)y, HW, C, in_pitch, in_off, splits, eps, bpi, (const a16_t*)x_lo, (a16_t*)y_lo);
  return glare_launch_status();
}

// GroupNorm as a PROLOGUE of the consuming conv (conv_igemm_kernel<.., GNP>, round 4): per (image, channel) the pair
// (a, d) with y = swish(a x + d), computed exactly as gn_apply_kernel computes them (fp64 combine of the partial blocks, then
// a = rstd * gamma, d = beta - mean * a in fp32) -- the conv's loader applies them to its halo tile in LDS and the normalised tensor
// is never written.  out: fp32 [B][C][2].
__global__ __launch_bounds__(256) void gn_coeffs_kernel(const float* __restrict__ partial, int splits, long long HW, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps, float* __restrict__ out, int C) {
  __shared__ float mean_s[GN_GROUPS], rstd_s[GN_GROUPS];
  const int b = blockIdx.x, cpg = C / GN_GROUPS;
  if (threadIdx.x < GN_GROUPS) {
    double s = 0.0, q = 0.0;
    for (int i = 0; i < splits; ++i) {
      s += partial[((size_t)b * splits + i) * GN_GROUPS * 2 + threadIdx.x * 2];
      q += partial[((size_t)b * splits + i) * GN_GROUPS * 2 + threadIdx.x * 2 + 1];
    }
    const double n = (double)HW * cpg;
    const double m = s / n;
    double var = q / n - m * m;
    if (var < 0.0) var = 0.0;
    mean_s[threadIdx.x] = (float)m;
    rstd_s[threadIdx.x] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    const int g = c / cpg;
    const float a = rstd_s[g] * gamma[c];
    out[((size_t)b * C + c) * 2] = a;
    out[((size_t)b * C + c) * 2 + 1] = beta[c] - mean_s[g] * a;
  }
}

extern "C" int glare_groupnorm_coeffs_f32(const float* stats, int splits, int B, long long HW, int C, const float* gamma, const float* beta,
                                          float eps, float* coef_out, glare_stream_t stream) {
  if (!stats || !gamma || !beta || !coef_out || splits <= 0 || B <= 0 || HW <= 0 || C <= 0 || C % GN_GROUPS || C > 2048) return GLARE_ERR_INVALID;
  hipLaunchKernelGGL(gn_coeffs_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, stats, splits, HW, gamma, beta, eps, coef_out, C);
  return glare_launch_status();
}

// stats: the [B][splits][32][2] (sum, sum of squares) block of the tensor's GroupNorm (from a conv's fused statistics or
// glare_add_groupnorm_stats_bf16); wq / wo: fp32 [C][C] row-major (out, in), bq / bo fp32 [C]; outputs bf16 [B][C][C] and fp32 [B][C]
// for glare_conv1x1_ws_image_bf16.  C a multiple of 32, <= 2048.
extern "C" int glare_attn_fold_groupnorm_f32(const float* stats, int splits, int B, long long HW, int C, const float* gamma,
                                             const float* beta, float eps, const float* wq, const float* bq, const float* wo,
                                             const float* bo, void* wq_out, float* bq_out, void* wo_out, float* bo_out,
                                             int feedback, glare_stream_t stream) {
  if (!stats || !gamma || !beta || !wq || !bq || !wo || !bo || !wq_out || !bq_out || !wo_out || !bo_out) return GLARE_ERR_INVALID;
  if (splits <= 0 || B <= 0 || B > 65535 || HW <= 0 || C <= 0 || C % GN_GROUPS || C > 2048) return GLARE_ERR_INVALID;
  hipLaunchKernelGGL(attn_fold_kernel, dim3(C, B), dim3(256), 0, (hipStream_t)stream, stats, splits, HW, gamma, beta, eps, wq, bq, wo, bo,
                     (a16_t*)wq_out, bq_out, (a16_t*)wo_out, bo_out, C, feedback);
  return glare_launch_status();
}
