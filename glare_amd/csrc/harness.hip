// Pre/post-processing of the inference harness on the device: the steps
// code/infer_dataset_lol.py:113-153 runs per image on the host with numpy / cv2, batched here so that the images cross
// PCIe once as uint8 and only one PSNR per image comes back.
//   pre : reflect-pad 20 px bottom / left (impad, :71-72), /255 (t, :42), log(clamp(x + 1e-3, min = 1e-3)) (:127-128)
//   post: crop [:, :, :h, 20:], clamp [0,1] (:135-140), gain = gray(GT/255) / gray(out) with the cv2.COLOR_BGR2GRAY weights
//         applied to RGB-ordered data, i.e. 0.114 ch0 + 0.587 ch1 + 0.299 ch2 (:142-144), clip, PSNR (utils2.py:32-36)
#include "common.h"

namespace {

constexpr int HB = 64;  // blocks per image of the two-level reductions

__global__ __launch_bounds__(256) void pre_kernel(const uint8_t* __restrict__ img, int B, int H, int W, int pad,
                                                  float* __restrict__ out) {
  const int Hp = H + pad, Wp = W + pad;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)B * 3 * Hp * Wp;
  if (i >= total) return;
  const int x = (int)(i % Wp), y = (int)((i / Wp) % Hp), c = (int)((i / ((long long)Wp * Hp)) % 3), b = (int)(i / ((long long)Wp * Hp * 3));
  // np.pad(..., 'reflect'): the edge is not repeated; pads longer than the image keep reflecting (period 2n - 2)
  const int py = y % (2 * H - 2), px = (x >= pad ? x - pad : pad - x) % (2 * W - 2);
  const int ys = py < H ? py : 2 * H - 2 - py;
  const int xs = px < W ? px : 2 * W - 2 - px;
  const float v = (float)img[(((long long)b * H + ys) * W + xs) * 3 + c];
  out[i] = logf(fmaxf(__fdiv_rn(v, 255.f) + 1e-3f, 1e-3f));
}

// pass 1: r = clamp(crop(out)) -> restored (NHWC), per-block sums of gray(r) and gray(gt/255)
__global__ __launch_bounds__(256) void post_crop_kernel(const float* __restrict__ out, const uint8_t* __restrict__ gt, int h, int w,
                                                        int Hp, int Wp, int pad, float* __restrict__ restored,
                                                        double* __restrict__ partial, int* __restrict__ bad_partial) {
  __shared__ double red[2][4];
  __shared__ int redc[4];
  const int b = blockIdx.y;
  const long long npix = (long long)h * w;
  double sr = 0.0, sg = 0.0;
  int bad = 0;       // non-finite values of the network output inside the crop, counted BEFORE the clamp: the clamp turns +inf into 1.0
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long long)HB * 256) {
    const int y = (int)(p / w), x = (int)(p % w);
    float r[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v = out[(((long long)b * 3 + c) * Hp + y) * Wp + x + pad];
      bad += (__float_as_uint(v) & 0x7f800000u) == 0x7f800000u;   // exponent all ones: inf or NaN (an fp16 overflow upstream)
      r[c] = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);     // torch.clamp (infer_dataset_lol.py:138): a NaN stays a NaN -- fminf / fmaxf
                                                      // would turn it into 0 and hide a broken image behind a finite PSNR
      restored[((long long)b * npix + p) * 3 + c] = r[c];
    }
    sr += (double)(0.114f * r[0] + 0.587f * r[1] + 0.299f * r[2]);
    if (gt) {
      const uint8_t* g = gt + ((long long)b * npix + p) * 3;
      sg += 0.114 * (g[0] / 255.0) + 0.587 * (g[1] / 255.0) + 0.299 * (g[2] / 255.0);
    }
  }
  for (int o = 32; o > 0; o >>= 1) { sr += __shfl_xor(sr, o, 64); sg += __shfl_xor(sg, o, 64); bad += __shfl_xor(bad, o, 64); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sr; red[1][threadIdx.x >> 6] = sg; redc[threadIdx.x >> 6] = bad; }
  __syncthreads();
  if (threadIdx.x < 2)
    partial[((size_t)b * HB + blockIdx.x) * 2 + threadIdx.x] =
        (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
  if (threadIdx.x == 2 && bad_partial) bad_partial[(size_t)b * HB + blockIdx.x] = (redc[0] + redc[1]) + (redc[2] + redc[3]);
}

__global__ void post_count_kernel(const int* __restrict__ bad_partial, int B, int* __restrict__ nonfinite) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int s = 0;
  for (int k = 0; k < HB; ++k) s += bad_partial[(size_t)b * HB + k];
  nonfinite[b] = s;
}

// pass 2: r = clip(r * gain), per-block sum of (gt/255 - r)^2
__global__ __launch_bounds__(256) void post_gain_kernel(float* __restrict__ restored, const uint8_t* __restrict__ gt, long long n,
                                                        const double* __restrict__ partial, int apply_gain,
                                                        double* __restrict__ mse_partial) {
  __shared__ double red[4];
  __shared__ float gain_s;
  const int b = blockIdx.y;
  if (threadIdx.x == 0) {
    double sr = 0.0, sg = 0.0;
    for (int k = 0; k < HB; ++k) { sr += partial[((size_t)b * HB + k) * 2]; sg += partial[((size_t)b * HB + k) * 2 + 1]; }
    gain_s = apply_gain ? (float)(sg / sr) : 1.f;      // ratio of the two means (same pixel count)
  }
  __syncthreads();
  const float gain = gain_s;
  double acc = 0.0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)HB * 256) {
    float r = restored[(long long)b * n + i];
    if (apply_gain) { r *= gain; r = r < 0.f ? 0.f : (r > 1.f ? 1.f : r); }   // np.clip: NaN propagates
    restored[(long long)b * n + i] = r;
    if (gt) {
      const double d = gt[(long long)b * n + i] / 255.0 - (double)r;
      acc += d * d;
    }
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) mse_partial[(size_t)b * HB + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void post_psnr_kernel(const double* __restrict__ mse_partial, long long n, int B, double* __restrict__ psnr) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double s = 0.0;
  for (int k = 0; k < HB; ++k) s += mse_partial[(size_t)b * HB + k];
  const double mse = s / (double)n;
  psnr[b] = mse == 0.0 ? 100.0 : 10.0 * log10(1.0 / mse);
}

// img_as_ubyte of both images as float planes in [0, 255] (the operands of calculate_ssim, infer_dataset_lol.py:152):
// restored float in [0,1] -> rint(clip * 255) (skimage's float -> uint8 conversion rounds to nearest even), the GT as is
__global__ __launch_bounds__(256) void ubyte_planes_kernel(const float* __restrict__ restored, const uint8_t* __restrict__ gt, long long n,
                                                           float* __restrict__ x255, float* __restrict__ y255) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  x255[i] = rintf(fminf(fmaxf(restored[i], 0.f), 1.f) * 255.f);
  y255[i] = (float)gt[i];
}

}  // namespace

// restored float in [0,1] -> uint8 = rint(clip * 255): img_as_ubyte(restored), the image the reference's loop hands to imwrite
__global__ __launch_bounds__(256) void to_ubyte_kernel(const float* __restrict__ restored, long long n, uint8_t* __restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = (uint8_t)rintf(fminf(fmaxf(restored[i], 0.f), 1.f) * 255.f);
}

extern "C" int glare_harness_to_ubyte(const float* restored_hwc, long long n, unsigned char* out_u8, glare_stream_t stream) {
  if (!restored_hwc || !out_u8 || n <= 0) return GLARE_ERR_INVALID;
  hipLaunchKernelGGL(to_ubyte_kernel, dim3((unsigned)cdivll(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), restored_hwc, n, out_u8);
  return glare_launch_status();
}

extern "C" int glare_harness_ubyte_planes_f32(const float* restored_hwc, const unsigned char* gt_hwc, long long n, float* x255, float* y255,
                                              glare_stream_t stream) {
  if (!restored_hwc || !gt_hwc || !x255 || !y255 || n <= 0) return GLARE_ERR_INVALID;
  hipLaunchKernelGGL(ubyte_planes_kernel, dim3((unsigned)cdivll(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), restored_hwc,
                     gt_hwc, n, x255, y255);
  return glare_launch_status();
}

extern "C" int glare_harness_preprocess_u8(const unsigned char* img_hwc, int B, int H, int W, int pad, float* out_nchw,
                                           glare_stream_t stream) {
  if (!img_hwc || !out_nchw || B <= 0 || H <= 1 || W <= 1 || pad < 0) return GLARE_ERR_INVALID;
  const long long total = (long long)B * 3 * (H + pad) * (W + pad);
  hipLaunchKernelGGL(pre_kernel, dim3((unsigned)cdivll(total, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), img_hwc, B, H, W,
                     pad, out_nchw);
  return glare_launch_status();
}

extern "C" size_t glare_harness_postprocess_workspace_bytes(int B) { return B <= 0 ? 0 : (size_t)B * HB * (3 * sizeof(double) + sizeof(int)); }

extern "C" int glare_harness_postprocess_flagged_f32(const float* out_nchw, const unsigned char* gt_hwc_or_null, int B, int h, int w,
                                                     int Hp, int Wp, int pad, float* restored_hwc, double* psnr_or_null,
                                                     int* nonfinite_or_null, void* workspace, size_t workspace_bytes, glare_stream_t stream) {
  if (!out_nchw || !restored_hwc || B <= 0 || h <= 0 || w <= 0 || h > Hp || w + pad > Wp || pad < 0) return GLARE_ERR_INVALID;
  if (psnr_or_null && !gt_hwc_or_null) return GLARE_ERR_INVALID;
  if (!workspace || workspace_bytes < glare_harness_postprocess_workspace_bytes(B)) return GLARE_ERR_WORKSPACE;
  double* partial = static_cast<double*>(workspace);
  double* mse_partial = partial + (size_t)B * HB * 2;
  int* bad_partial = reinterpret_cast<int*>(mse_partial + (size_t)B * HB);
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(post_crop_kernel, dim3(HB, B), dim3(256), 0, s, out_nchw, gt_hwc_or_null, h, w, Hp, Wp, pad, restored_hwc, partial,
                     nonfinite_or_null ? bad_partial : (int*)nullptr);
  const long long n = (long long)h * w * 3;
  hipLaunchKernelGGL(post_gain_kernel, dim3(HB, B), dim3(256), 0, s, restored_hwc, gt_hwc_or_null, n, partial, gt_hwc_or_null ? 1 : 0,
                     mse_partial);
  if (psnr_or_null) hipLaunchKernelGGL(post_psnr_kernel, dim3(cdiv(B, 64)), dim3(64), 0, s, mse_partial, n, B, psnr_or_null);
  if (nonfinite_or_null) hipLaunchKernelGGL(post_count_kernel, dim3(cdiv(B, 64)), dim3(64), 0, s, bad_partial, B, nonfinite_or_null);
  return glare_launch_status();
}

extern "C" int glare_harness_postprocess_f32(const float* out_nchw, const unsigned char* gt_hwc_or_null, int B, int h, int w,
                                             int Hp, int Wp, int pad, float* restored_hwc, double* psnr_or_null, void* workspace,
                                             size_t workspace_bytes, glare_stream_t stream) {
  return glare_harness_postprocess_flagged_f32(out_nchw, gt_hwc_or_null, B, h, w, Hp, Wp, pad, restored_hwc, psnr_or_null, nullptr, workspace,
                                               workspace_bytes, stream);
}
