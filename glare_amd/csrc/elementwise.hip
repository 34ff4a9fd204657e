// HBM-bound glue of the AFT decoder and the layout conversions at the module boundary.
//   mix      Mix.forward: sigmoid(w)*a + (1-sigmoid(w))*b          (deformableDecoder_arch.py:587-590)
//   rescale  h + x_w * (mean(h) / mean(x_w))                      (deformableDecoder_arch.py:567)
//   nchw<->nhwc   the reference's module surface is NCHW fp32; the kernels work on NHWC
// All accesses are 16-B vectors; reductions are two-level (per-block fp32 partials, fp64 combine)
// and deterministic (no atomics).
#include "common.h"

namespace {

constexpr int EW_THREADS = 256;

__global__ __launch_bounds__(EW_THREADS) void mix_kernel(const a16_t* __restrict__ a, int a_pitch, int a_off,
                                                         const a16_t* __restrict__ b, int b_pitch, int b_off,
                                                         a16_t* __restrict__ out, int o_pitch, int o_off, long long npix,
                                                         int C, float f, const float* __restrict__ w_dev) {
  if (w_dev) f = 1.0f / (1.0f + expf(-w_dev[0]));   // the mixing logit read on the device: no host round trip
  const int CP = C / 8;
  const long long total = npix * CP;
  for (long long i = (long long)blockIdx.x * EW_THREADS + threadIdx.x; i < total; i += (long long)gridDim.x * EW_THREADS) {
    const long long p = i / CP;
    const int c = (int)(i % CP) * 8;
    const u32x4 va = *reinterpret_cast<const u32x4*>(a + p * a_pitch + a_off + c);
    const u32x4 vb = *reinterpret_cast<const u32x4*>(b + p * b_pitch + b_off + c);
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      o[e] = pack_a2(alo(va[e]) * f + alo(vb[e]) * (1.f - f), ahi(va[e]) * f + ahi(vb[e]) * (1.f - f));
    *reinterpret_cast<u32x4*>(out + p * o_pitch + o_off + c) = o;
  }
}

// partial[b][blk][2]: sum(h), sum(xw) over this block's slice of sample b
__global__ __launch_bounds__(EW_THREADS) void rescale_sum_kernel(const a16_t* __restrict__ h, const float* __restrict__ xw,
                                                                 float* __restrict__ partial, long long n_per_sample,
                                                                 int blocks_per_sample) {
  __shared__ float red[2][EW_THREADS / 64];
  const int b = blockIdx.x / blocks_per_sample, blk = blockIdx.x % blocks_per_sample;
  const long long nvec = n_per_sample / 8;
  const long long per = (nvec + blocks_per_sample - 1) / blocks_per_sample;
  const long long v0 = blk * per, v1 = min(nvec, v0 + per);
  const a16_t* hb = h + (size_t)b * n_per_sample;
  const float* xb = xw + (size_t)b * n_per_sample;
  float sh = 0.f, sx = 0.f;
  for (long long v = v0 + threadIdx.x; v < v1; v += EW_THREADS) {
    const u32x4 hv = *reinterpret_cast<const u32x4*>(hb + v * 8);
    const f32x4 x0 = *reinterpret_cast<const f32x4*>(xb + v * 8), x1 = *reinterpret_cast<const f32x4*>(xb + v * 8 + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) sh += alo(hv[e]) + ahi(hv[e]);
    sx += (x0[0] + x0[1]) + (x0[2] + x0[3]) + (x1[0] + x1[1]) + (x1[2] + x1[3]);
  }
  sh = wave_sum(sh);
  sx = wave_sum(sx);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sh; red[1][threadIdx.x >> 6] = sx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.f, c = 0.f;
    for (int w = 0; w < EW_THREADS / 64; ++w) { a += red[0][w]; c += red[1][w]; }
    partial[(size_t)blockIdx.x * 2] = a;
    partial[(size_t)blockIdx.x * 2 + 1] = c;
  }
}

__global__ __launch_bounds__(EW_THREADS) void rescale_apply_kernel(const a16_t* __restrict__ h, const float* __restrict__ xw,
                                                                   const float* __restrict__ partial, a16_t* __restrict__ out,
                                                                   long long n_per_sample, int blocks_per_sample, int B,
                                                                   int whole_batch) {
  __shared__ float ratio_s;
  const int b = blockIdx.x / blocks_per_sample, blk = blockIdx.x % blocks_per_sample;
  if (threadIdx.x == 0) {
    double sh = 0.0, sx = 0.0;
    const int b0 = whole_batch ? 0 : b, b1 = whole_batch ? B : b + 1;
    for (int i = b0 * blocks_per_sample; i < b1 * blocks_per_sample; ++i) { sh += partial[2 * i]; sx += partial[2 * i + 1]; }
    ratio_s = (float)(sh / sx);  // mean(h)/mean(x_w): equal element counts cancel
  }
  __syncthreads();
  const float r = ratio_s;
  const long long nvec = n_per_sample / 8;
  const long long per = (nvec + blocks_per_sample - 1) / blocks_per_sample;
  const long long v0 = blk * per, v1 = min(nvec, v0 + per);
  const a16_t* hb = h + (size_t)b * n_per_sample;
  const float* xb = xw + (size_t)b * n_per_sample;
  a16_t* ob = out + (size_t)b * n_per_sample;
  for (long long v = v0 + threadIdx.x; v < v1; v += EW_THREADS) {
    const u32x4 hv = *reinterpret_cast<const u32x4*>(hb + v * 8);
    const f32x4 x0 = *reinterpret_cast<const f32x4*>(xb + v * 8), x1 = *reinterpret_cast<const f32x4*>(xb + v * 8 + 4);
    u32x4 o;
    o[0] = pack_a2(alo(hv[0]) + x0[0] * r, ahi(hv[0]) + x0[1] * r);
    o[1] = pack_a2(alo(hv[1]) + x0[2] * r, ahi(hv[1]) + x0[3] * r);
    o[2] = pack_a2(alo(hv[2]) + x1[0] * r, ahi(hv[2]) + x1[1] * r);
    o[3] = pack_a2(alo(hv[3]) + x1[2] * r, ahi(hv[3]) + x1[3] * r);
    *reinterpret_cast<u32x4*>(ob + v * 8) = o;
  }
}

// ---- round 6: the mean rescale without its own statistics pass -------------------------------------------------------------------
// sum(h) comes from the kernel that PRODUCES h (the Mix in front of the warp), sum(x_w) from the DCN's epilogue (dcn.hip, per-tile
// sums of its fp32 outputs, tiles cut per image), a one-block-per-image kernel turns the two partial sets into the ratio, and the apply pass reads x_w as
// the 16-bit tensor the DCN wrote (or fp32).  Same arithmetic as rescale_sum / rescale_apply above: fp64 combine, ratio in fp32.

// Mix.forward + partial[b][blk] = sum of the block's ROUNDED outputs (what rescale_sum_kernel reads back); dense tensors of n_per_sample elements
__global__ __launch_bounds__(EW_THREADS) void mix_sum_kernel(const a16_t* __restrict__ a, const a16_t* __restrict__ b, a16_t* __restrict__ out,
                                                             long long n_per_sample, int blocks_per_sample, float f,
                                                             const float* __restrict__ w_dev, float* __restrict__ partial) {
  __shared__ float red[EW_THREADS / 64];
  if (w_dev) f = 1.0f / (1.0f + expf(-w_dev[0]));
  const int bi = blockIdx.x / blocks_per_sample, blk = blockIdx.x % blocks_per_sample;
  const long long nvec = n_per_sample / 8;
  const long long per = (nvec + blocks_per_sample - 1) / blocks_per_sample;
  const long long v0 = blk * per, v1 = min(nvec, v0 + per);
  const size_t base = (size_t)bi * n_per_sample;
  float sh = 0.f;
  for (long long v = v0 + threadIdx.x; v < v1; v += EW_THREADS) {
    const u32x4 va = *reinterpret_cast<const u32x4*>(a + base + v * 8);
    const u32x4 vb = *reinterpret_cast<const u32x4*>(b + base + v * 8);
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[e] = pack_a2(alo(va[e]) * f + alo(vb[e]) * (1.f - f), ahi(va[e]) * f + ahi(vb[e]) * (1.f - f));
      sh += alo(o[e]) + ahi(o[e]);
    }
    *reinterpret_cast<u32x4*>(out + base + v * 8) = o;
  }
  sh = wave_sum(sh);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sh;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ratio[b] = sum(h_b) / sum(x_b) (whole_batch: one ratio from all images, written to every slot).  h_part: [B][h_bps]; x_part: the
// DCN's per-image tile sums [B][x_tpi].  One block per image; every thread sums a fixed strided subset in fp64, the block combines in a
// fixed order: an image's ratio depends on its own tiles only -- the same in a batch of 8 as alone.
__global__ __launch_bounds__(256) void rescale_ratio_kernel(const float* __restrict__ h_part, int h_bps, const float* __restrict__ x_part,
                                                            int x_tpi, int B, int whole_batch, float* __restrict__ ratio) {
  __shared__ double red[2][256];
  const int b = blockIdx.x;
  const int b0 = whole_batch ? 0 : b, b1 = whole_batch ? B : b + 1;
  double sh = 0.0, sx = 0.0;
  for (long long i = (long long)b0 * h_bps + threadIdx.x; i < (long long)b1 * h_bps; i += 256) sh += (double)h_part[i];
  for (long long i = (long long)b0 * x_tpi + threadIdx.x; i < (long long)b1 * x_tpi; i += 256) sx += (double)x_part[i];
  red[0][threadIdx.x] = sh;
  red[1][threadIdx.x] = sx;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) { red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[1][threadIdx.x] += red[1][threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) ratio[b] = (float)(red[0][0] / red[1][0]);
}

template <bool X16>
__global__ __launch_bounds__(EW_THREADS) void rescale_apply2_kernel(const a16_t* __restrict__ h, const void* __restrict__ xw_,
                                                                    const float* __restrict__ ratio, a16_t* __restrict__ out,
                                                                    long long n_per_sample, int blocks_per_sample) {
  const int b = blockIdx.x / blocks_per_sample, blk = blockIdx.x % blocks_per_sample;
  const float r = ratio[b];
  const long long nvec = n_per_sample / 8;
  const long long per = (nvec + blocks_per_sample - 1) / blocks_per_sample;
  const long long v0 = blk * per, v1 = min(nvec, v0 + per);
  const size_t base = (size_t)b * n_per_sample;
  for (long long v = v0 + threadIdx.x; v < v1; v += EW_THREADS) {
    const u32x4 hv = *reinterpret_cast<const u32x4*>(h + base + v * 8);
    u32x4 o;
    if constexpr (X16) {
      const u32x4 xv = *reinterpret_cast<const u32x4*>(reinterpret_cast<const a16_t*>(xw_) + base + v * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = pack_a2(alo(hv[e]) + alo(xv[e]) * r, ahi(hv[e]) + ahi(xv[e]) * r);
    } else {
      const float* xb = reinterpret_cast<const float*>(xw_) + base + v * 8;
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(xb), x1 = *reinterpret_cast<const f32x4*>(xb + 4);
      o[0] = pack_a2(alo(hv[0]) + x0[0] * r, ahi(hv[0]) + x0[1] * r);
      o[1] = pack_a2(alo(hv[1]) + x0[2] * r, ahi(hv[1]) + x0[3] * r);
      o[2] = pack_a2(alo(hv[2]) + x1[0] * r, ahi(hv[2]) + x1[1] * r);
      o[3] = pack_a2(alo(hv[3]) + x1[2] * r, ahi(hv[3]) + x1[3] * r);
    }
    *reinterpret_cast<u32x4*>(out + base + v * 8) = o;
  }
}

// [B][C][HW] (fp32) <-> [B][HW][C] (fp32 or bf16) through a 32 x 33 LDS tile.
template <bool TO_NHWC, bool BF16>
__global__ __launch_bounds__(256) void layout_kernel(const void* __restrict__ src, void* __restrict__ dst, int C,
                                                     long long HW, int pitch, int off) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const long long p0 = (long long)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  if (TO_NHWC) {
    const float* s = reinterpret_cast<const float*>(src) + (size_t)b * C * HW;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = c0 + ty + 8 * k;
      const long long p = p0 + tx;
      tile[ty + 8 * k][tx] = (c < C && p < HW) ? s[(size_t)c * HW + p] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long long p = p0 + ty + 8 * k;
      const int c = c0 + tx;
      if (c < C && p < HW) {
        const size_t o = ((size_t)b * HW + p) * pitch + off + c;
        if (BF16) reinterpret_cast<a16_t*>(dst)[o] = f2a(tile[tx][ty + 8 * k]);
        else reinterpret_cast<float*>(dst)[o] = tile[tx][ty + 8 * k];
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long long p = p0 + ty + 8 * k;
      const int c = c0 + tx;
      float v = 0.f;
      if (c < C && p < HW) {
        const size_t o = ((size_t)b * HW + p) * pitch + off + c;
        v = BF16 ? a2f(reinterpret_cast<const a16_t*>(src)[o]) : reinterpret_cast<const float*>(src)[o];
      }
      tile[ty + 8 * k][tx] = v;
    }
    __syncthreads();
    float* d = reinterpret_cast<float*>(dst) + (size_t)b * C * HW;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = c0 + ty + 8 * k;
      const long long p = p0 + tx;
      if (c < C && p < HW) d[(size_t)c * HW + p] = tile[tx][ty + 8 * k];
    }
  }
}

int ew_blocks(long long work_items) {
  long long b = (work_items + EW_THREADS - 1) / EW_THREADS;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

int rescale_bps(long long n_per_sample) {
  long long b = n_per_sample / (8LL * EW_THREADS * 8);
  return (int)(b < 1 ? 1 : (b > 512 ? 512 : b));
}

// fp32 -> hi / lo pair: hi = round16(v), lo = round16(v - hi)  (the conv_in output opening the encoder's residual stream)
__global__ __launch_bounds__(EW_THREADS) void split_hilo_kernel(const float* __restrict__ src, long long n8, a16_t* __restrict__ hi,
                                                                a16_t* __restrict__ lo) {
  for (long long i = (long long)blockIdx.x * EW_THREADS + threadIdx.x; i < n8; i += (long long)gridDim.x * EW_THREADS) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(src + i * 8), c = *reinterpret_cast<const f32x4*>(src + i * 8 + 4);
    const float v[8] = {a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]};
    u32x4 h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      h[e] = pack_a2(v[2 * e], v[2 * e + 1]);
      l[e] = pack_a2(v[2 * e] - alo(h[e]), v[2 * e + 1] - ahi(h[e]));
    }
    *reinterpret_cast<u32x4*>(hi + i * 8) = h;
    *reinterpret_cast<u32x4*>(lo + i * 8) = l;
  }
}

}  // namespace

extern "C" int glare_split_hilo_f32(const float* src, long long n, void* hi_bf16, void* lo_bf16, glare_stream_t stream) {
  if (!src || !hi_bf16 || !lo_bf16 || n <= 0 || (n % 8)) return GLARE_ERR_INVALID;
  hipLaunchKernelGGL(split_hilo_kernel, dim3(ew_blocks(n / 8)), dim3(EW_THREADS), 0, (hipStream_t)stream, src, n / 8, (a16_t*)hi_bf16,
                     (a16_t*)lo_bf16);
  return glare_launch_status();
}

extern "C" int glare_mix_bf16(const void* a, int a_pitch, int a_off, const void* b, int b_pitch, int b_off, void* out,
                              int out_pitch, int out_off, long long n_pixels, int C, float mix_w, glare_stream_t stream) {
  if (!a || !b || !out || n_pixels <= 0 || C <= 0) return GLARE_ERR_INVALID;
  if ((C | a_pitch | a_off | b_pitch | b_off | out_pitch | out_off) % 8) return GLARE_ERR_UNSUPPORTED;
  const float f = 1.0f / (1.0f + expf(-mix_w));
  hipLaunchKernelGGL(mix_kernel, dim3(ew_blocks(n_pixels * (C / 8))), dim3(EW_THREADS), 0, (hipStream_t)stream,
                     (const a16_t*)a, a_pitch, a_off, (const a16_t*)b, b_pitch, b_off, (a16_t*)out, out_pitch, out_off,
                     n_pixels, C, f, (const float*)nullptr);
  return glare_launch_status();
}

extern "C" int glare_mix_dev_bf16(const void* a, int a_pitch, int a_off, const void* b, int b_pitch, int b_off, void* out,
                                  int out_pitch, int out_off, long long n_pixels, int C, const float* mix_w_device,
                                  glare_stream_t stream) {
  if (!a || !b || !out || !mix_w_device || n_pixels <= 0 || C <= 0) return GLARE_ERR_INVALID;
  if ((C | a_pitch | a_off | b_pitch | b_off | out_pitch | out_off) % 8) return GLARE_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(mix_kernel, dim3(ew_blocks(n_pixels * (C / 8))), dim3(EW_THREADS), 0, (hipStream_t)stream,
                     (const a16_t*)a, a_pitch, a_off, (const a16_t*)b, b_pitch, b_off, (a16_t*)out, out_pitch, out_off,
                     n_pixels, C, 0.f, mix_w_device);
  return glare_launch_status();
}

extern "C" size_t glare_mean_rescale_workspace_bytes(int B, long long n_per_sample) {
  if (B <= 0 || n_per_sample <= 0) return 0;
  return (size_t)B * rescale_bps(n_per_sample) * 2 * sizeof(float);
}

extern "C" int glare_mean_rescale_bf16(const void* h, const float* xw, void* out, int B, long long n_per_sample,
                                       int whole_batch_mean, void* workspace, size_t workspace_bytes,
                                       glare_stream_t stream_) {
  if (!h || !xw || !out || B <= 0 || n_per_sample <= 0) return GLARE_ERR_INVALID;
  if (n_per_sample % 8) return GLARE_ERR_UNSUPPORTED;
  if (!workspace || workspace_bytes < glare_mean_rescale_workspace_bytes(B, n_per_sample)) return GLARE_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  const int bps = rescale_bps(n_per_sample);
  hipLaunchKernelGGL(rescale_sum_kernel, dim3(B * bps), dim3(EW_THREADS), 0, stream, (const a16_t*)h, xw, (float*)workspace,
                     n_per_sample, bps);
  hipLaunchKernelGGL(rescale_apply_kernel, dim3(B * bps), dim3(EW_THREADS), 0, stream, (const a16_t*)h, xw,
                     (const float*)workspace, (a16_t*)out, n_per_sample, bps, B, whole_batch_mean);
  return glare_launch_status();
}

extern "C" int glare_mix_sum_blocks(long long n_per_sample) { return n_per_sample <= 0 ? 0 : rescale_bps(n_per_sample); }

extern "C" int glare_mix_sum_bf16(const void* a, const void* b, void* out, int B, long long n_per_sample, float mix_w,
                                  const float* mix_w_dev_or_null, float* sum_partial, glare_stream_t stream) {
  if (!a || !b || !out || !sum_partial || B <= 0 || n_per_sample <= 0) return GLARE_ERR_INVALID;
  if (n_per_sample % 8) return GLARE_ERR_UNSUPPORTED;
  const int bps = rescale_bps(n_per_sample);
  const float f = 1.0f / (1.0f + expf(-mix_w));
  hipLaunchKernelGGL(mix_sum_kernel, dim3(B * bps), dim3(EW_THREADS), 0, (hipStream_t)stream, (const a16_t*)a, (const a16_t*)b, (a16_t*)out,
                     n_per_sample, bps, f, mix_w_dev_or_null, sum_partial);
  return glare_launch_status();
}

extern "C" int glare_mean_rescale_fused_bf16(const void* h, const void* xw, int xw_is_16bit, void* out, int B, long long n_per_sample,
                                             long long pixels_per_sample, const float* h_sum_partial, const float* xw_tile_sums, int tile_pixels,
                                             int whole_batch_mean, float* ratio_scratch, glare_stream_t stream_) {
  if (!h || !xw || !out || !h_sum_partial || !xw_tile_sums || !ratio_scratch || B <= 0 || n_per_sample <= 0 || pixels_per_sample <= 0 ||
      tile_pixels <= 0)
    return GLARE_ERR_INVALID;
  if (n_per_sample % 8) return GLARE_ERR_UNSUPPORTED;
  hipStream_t stream = (hipStream_t)stream_;
  const int bps = rescale_bps(n_per_sample);
  hipLaunchKernelGGL(rescale_ratio_kernel, dim3(B), dim3(256), 0, stream, h_sum_partial, bps, xw_tile_sums,
                     (int)cdivll(pixels_per_sample, tile_pixels), B, whole_batch_mean, ratio_scratch);
  if (xw_is_16bit)
    hipLaunchKernelGGL(rescale_apply2_kernel<true>, dim3(B * bps), dim3(EW_THREADS), 0, stream, (const a16_t*)h, xw, (const float*)ratio_scratch,
                       (a16_t*)out, n_per_sample, bps);
  else
    hipLaunchKernelGGL(rescale_apply2_kernel<false>, dim3(B * bps), dim3(EW_THREADS), 0, stream, (const a16_t*)h, xw, (const float*)ratio_scratch,
                       (a16_t*)out, n_per_sample, bps);
  return glare_launch_status();
}

extern "C" int glare_nchw_to_nhwc(const float* src_nchw, void* dst_nhwc, int B, int C, long long HW, int dst_pitch,
                                  int dst_off, int dst_is_bf16, glare_stream_t stream) {
  if (!src_nchw || !dst_nhwc || B <= 0 || C <= 0 || HW <= 0 || dst_off + C > dst_pitch) return GLARE_ERR_INVALID;
  const dim3 grid((unsigned)((HW + 31) / 32), (unsigned)((C + 31) / 32), (unsigned)B);
  if (dst_is_bf16)
    hipLaunchKernelGGL((layout_kernel<true, true>), grid, dim3(256), 0, (hipStream_t)stream, src_nchw, dst_nhwc, C, HW,
                       dst_pitch, dst_off);
  else
    hipLaunchKernelGGL((layout_kernel<true, false>), grid, dim3(256), 0, (hipStream_t)stream, src_nchw, dst_nhwc, C, HW,
                       dst_pitch, dst_off);
  return glare_launch_status();
}

extern "C" int glare_nhwc_to_nchw(const void* src_nhwc, float* dst_nchw, int B, int C, long long HW, int src_pitch,
                                  int src_off, int src_is_bf16, glare_stream_t stream) {
  if (!src_nhwc || !dst_nchw || B <= 0 || C <= 0 || HW <= 0 || src_off + C > src_pitch) return GLARE_ERR_INVALID;
  const dim3 grid((unsigned)((HW + 31) / 32), (unsigned)((C + 31) / 32), (unsigned)B);
  if (src_is_bf16)
    hipLaunchKernelGGL((layout_kernel<false, true>), grid, dim3(256), 0, (hipStream_t)stream, src_nhwc, dst_nchw, C, HW,
                       src_pitch, src_off);
  else
    hipLaunchKernelGGL((layout_kernel<false, false>), grid, dim3(256), 0, (hipStream_t)stream, src_nhwc, dst_nchw, C, HW,
                       src_pitch, src_off);
  return glare_launch_status();
}
