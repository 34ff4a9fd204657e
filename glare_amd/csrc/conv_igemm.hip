// im2col-free direct convolution (3x3 / 1x1) as an implicit GEMM on bf16 MFMA.
//
// Replaces every torch.nn.Conv2d with >= 8 input channels on the path (reference:
// encoder_decoder.py:43-52,62-75,88-115,146-165,355,399,467,509; deformableDecoder_arch.py:282,
// 484; deform_conv.py:357-364 conv_offset; flow.py:13-70) -- cuDNN in the reference.
//
// Data layout: activations NHWC bf16, so the GEMM's K dimension (input channels) is contiguous
// per pixel and an MFMA A-fragment (32 pixels x 16 channels) is two dense 512-B runs of LDS.
//   out[pixel, co] = sum_{tap, ci} in[pixel + tap, ci] * w[tap][co][ci]
// One workgroup (4 waves) computes an 8 x 32 pixel tile x TN output channels.  Per stage it
// stages KC = 16*KSTEPS input channels of the (8*s + k - 1) x (32*s + k - 1) halo tile and the
// matching weights in LDS ONCE; the k*k taps are then just shifted LDS addresses -- no im2col
// buffer, no re-read of the input per tap.  LDS images are split in 8-channel planes
// ([kstep][khalf][position][8 ch]) so that a wave's ds_read_b128 fragment read is a dense,
// bank-conflict-free 512-B run per half-wave.  Both images are double-buffered and filled by
// LDS-DMA (buffer_load_dwordx4 ... lds through per-image descriptors: no staging VGPRs, no ds_write pass, zero padding by range check) one stage ahead of the
// MFMAs; 3 workgroups per CU (168 VGPRs, 47 KB LDS each) cover the DMA issue stalls and the barrier skew.
// The zero padding, the nearest x2 upsample (Upsample, encoder_decoder.py:50) and the
// asymmetric stride-2 padding (Downsample, encoder_decoder.py:71-73) are address arithmetic
// in the loader; bias, residual add, activation and layout conversion are the epilogue.
//
// KS == 2 is the SUB-PIXEL form of "nearest x2 upsample, then 3x3 conv" (Upsample, encoder_decoder.py:43-52): output pixel
// (2i+a, 2j+b) only ever sees input rows {i-1, i} (a = 0) or {i, i+1} (a = 1) of the LOW-resolution source, likewise for
// columns, so each of the four output phases (a, b) is a 2x2 conv of the source with the 3x3 taps that land on the same
// source pixel pre-summed (in fp32, then rounded to bf16 once): 16 tap-MACs per source pixel instead of 36, the same
// result up to that one rounding.  A workgroup computes one phase of an 8 x 32 SOURCE tile and scatters it to the
// interleaved output positions; the four phases of a tile are adjacent in the launch order (shared halo in L2).
#include "conv_igemm_kernel.h"

namespace {

// Packs OIHW fp32 weights into the stage-ordered bf16 image the kernel copies verbatim:
// [co_tile][stage][tap][kstep][khalf][TN][8], zero-filled outside Cout / Cin.
// dgrad != 0: `w` is the FORWARD filter [Cin_real][Cout][KS][KS] of which the data-gradient filter is wanted
// (w'[co][ci][tap] = w[ci][co][KK-1-tap], input channels ci >= Cin_real zero): no flip/transpose/pad pass on the host.
// One thread = one 16-B chunk (8 consecutive input channels of one (tap, output channel)): the index arithmetic once per chunk.
__device__ __forceinline__ void pack_weight_chunk(const float* __restrict__ w, a16_t* __restrict__ out, long long chunk, int Cout, int Cin,
                                                  int KS, int TN, int KSTEPS, int n_stages, int dgrad, int Cin_real) {
  long long t = chunk;
  const int n = t % TN; t /= TN;
  const int khalf = t % 2; t /= 2;
  const int ks = t % KSTEPS; t /= KSTEPS;
  const int KK = KS * KS;
  const int tap = t % KK; t /= KK;
  const int s = t % n_stages; t /= n_stages;
  const int ct = (int)t;
  const int co = ct * TN + n;
  const int ci0 = (s * KSTEPS + ks) * 16 + khalf * 8;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int ci = ci0 + e;
    v[e] = 0.f;
    if (dgrad) {
      if (co < Cout && ci < Cin_real) v[e] = w[((size_t)ci * Cout + co) * KK + (KK - 1 - tap)];
    } else if (co < Cout && ci < Cin) {
      v[e] = w[((size_t)co * Cin + ci) * KK + tap];
    }
  }
  u32x4 pk;
#pragma unroll
  for (int e = 0; e < 4; ++e) pk[e] = pack_a2(v[2 * e], v[2 * e + 1]);
  *reinterpret_cast<u32x4*>(out + chunk * 8) = pk;
}

constexpr int PACK_BLOCK_ELEMS = 2048;   // packed elements per 256-thread block (glare_pack_job.block_begin counts these)

// blockIdx.y = filter of a batch of equally shaped filters (consecutive in `w`, consecutive packed images in `out`).
__global__ void pack_weight_kernel(const float* __restrict__ w, a16_t* __restrict__ out, int Cout, int Cin, int KS,
                                   int TN, int KSTEPS, int n_stages, long long total, int dgrad, int Cin_real) {
  const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (c * 8 >= total) return;
  w += (size_t)blockIdx.y * (dgrad ? (size_t)Cin_real * Cout : (size_t)Cout * Cin) * KS * KS;
  out += (size_t)blockIdx.y * total;
  pack_weight_chunk(w, out, c, Cout, Cin, KS, TN, KSTEPS, n_stages, dgrad, Cin_real);
}

// Many filters of DIFFERENT shapes in one launch (the trainable convs of a training step, re-packed after the optimizer update):
// block b belongs to the last job whose block_begin <= b.
__global__ void pack_weight_multi_kernel(const glare_pack_job* __restrict__ jobs, int n_jobs) {
  int lo = 0, hi = n_jobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].block_begin <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const glare_pack_job j = jobs[lo];
  const long long c = ((long long)blockIdx.x - j.block_begin) * 256 + threadIdx.x;   // 16-B chunk of this job
  if (c * 8 >= j.total) return;
  if (j.kind == GLARE_PACK_PLAIN_BF16) {
    a16_t* o = static_cast<a16_t*>(j.out);
    for (long long i = c * 8; i < min(j.total, c * 8 + 8); ++i) o[i] = f2a(j.w[i]);
    return;
  }
  pack_weight_chunk(j.w, static_cast<a16_t*>(j.out), c, j.cout, j.cin, j.ksize, j.tn, j.ksteps, j.n_stages, j.kind == GLARE_PACK_DGRAD,
                    j.cin_real);
}

// Sub-pixel upsample filters: [phase = a*2+b][co_tile][stage][tap = r*2+c][khalf][TN][8] (KSTEPS = 1), where tap (r, c)
// of phase (a, b) is the sum of the 3x3 taps (ky, kx) that read the same source pixel: rows a=0: r=0 <- {0}, r=1 <- {1,2};
// a=1: r=0 <- {0,1}, r=1 <- {2}; columns alike with b.
// presummed: w is already the four phase filters, fp32 [phase][Cout][Cin][2][2] (round 6: summed and rounded with error feedback on the
// host side, ops.PackedConv(upsample_subpixel=True)); otherwise the 3x3 filter, summed here.
__global__ void pack_weight_subpix_kernel(const float* __restrict__ w, a16_t* __restrict__ out, int Cout, int Cin, int TN,
                                          int n_stages, int co_tiles, long long total, int presummed) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  long long t = i;
  const int e = t % 8; t /= 8;
  const int n = t % TN; t /= TN;
  const int khalf = t % 2; t /= 2;
  const int tap = t % 4; t /= 4;
  const int s = t % n_stages; t /= n_stages;
  const int ct = t % co_tiles; t /= co_tiles;
  const int phase = (int)t;
  const int a = phase >> 1, b = phase & 1, r = tap >> 1, c = tap & 1;
  const int ky0 = a == 0 ? (r == 0 ? 0 : 1) : (r == 0 ? 0 : 2), ky1 = a == 0 ? (r == 0 ? 0 : 2) : (r == 0 ? 1 : 2);
  const int kx0 = b == 0 ? (c == 0 ? 0 : 1) : (c == 0 ? 0 : 2), kx1 = b == 0 ? (c == 0 ? 0 : 2) : (c == 0 ? 1 : 2);
  const int co = ct * TN + n;
  const int ci = s * 16 + khalf * 8 + e;
  float v = 0.f;
  if (co < Cout && ci < Cin) {
    if (presummed) {
      v = w[((((size_t)phase * Cout + co) * Cin + ci) * 2 + r) * 2 + c];
    } else {
      const float* wp = w + ((size_t)co * Cin + ci) * 9;
      for (int ky = ky0; ky <= ky1; ++ky)
        for (int kx = kx0; kx <= kx1; ++kx) v += wp[ky * 3 + kx];
    }
  }
  out[i] = f2a(v);
}

// GroupNorm partial reduction: [b][part][Cout/4][2] -> the [B][1][32][2] partial format gn_apply consumes
__global__ __launch_bounds__(256) void gn_part_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, int nparts,
                                                             int Cout) {
  __shared__ float red[2][4];
  const int b = blockIdx.x / 32, g = blockIdx.x % 32;
  const int upg = Cout / 128;  // 4-channel units per group (cpg / 4)
  const int U = Cout / 4;
  float s = 0.f, q = 0.f;
  for (int i = threadIdx.x; i < nparts * upg; i += 256) {
    const int pt = i / upg, u = g * upg + i % upg;
    const float* src = part + (((size_t)b * nparts + pt) * U + u) * 2;
    s += src[0];
    q += src[1];
  }
  s = wave_sum(s);
  q = wave_sum(q);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = q; }
  __syncthreads();
  if (threadIdx.x == 0) {
    out[((size_t)b * 32 + g) * 2] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    out[((size_t)b * 32 + g) * 2 + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

struct Variant {
  int tn, ksteps;
};

// Measured and dropped: a 256-cout tile for the 1x1 convs (wave 128 px x 128 co, 256 accumulator registers, one workgroup per
// CU to halve the DMA pieces per MFMA): 250-267 TFLOP/s against 435-444 for the 128-cout tile at 512 -> 512/1024 -- like the
// 3x3 kernel, the 1x1 lives on the three workgroups per CU that cover its DMA issue stalls.
Variant pick_variant(int ksize, int cout, int cout_tile = 0) {
  Variant v;
  v.tn = cout > 64 ? 128 : (cout > 32 ? 64 : 32);
  if ((cout_tile == 32 || cout_tile == 64) && cout_tile < v.tn) v.tn = cout_tile;   // a narrower tile than the default: more workgroups
  v.ksteps = ksize == 1 ? 2 : 1;
  return v;
}

}  // namespace

extern "C" long long glare_conv2d_packed_weight_elems_tile(int cout, int cin_total, int ksize, int cout_tile) {
  if (cout <= 0 || cin_total <= 0 || (ksize != 1 && ksize != 3)) return GLARE_ERR_INVALID;
  const Variant v = pick_variant(ksize, cout, cout_tile);
  const int kc = 16 * v.ksteps;
  const long long stages = (cin_total + kc - 1) / kc, co_tiles = (cout + v.tn - 1) / v.tn;
  return co_tiles * stages * ksize * ksize * v.ksteps * 2 * v.tn * 8;
}

extern "C" long long glare_conv2d_packed_weight_elems(int cout, int cin_total, int ksize) {
  return glare_conv2d_packed_weight_elems_tile(cout, cin_total, ksize, 0);
}

// Workgroups of 8 x 32 output pixels x tile channels.  Measured on MI355X (tools/probes/conv_small_grid.py, 3x3, TFLOP/s with the
// 128 / 64 / 32 tile): 128->128 @256x256 B=1 610 / 718 / 616; 256->256 @128x128 B=1 426 / 572 / 632; 512->512 @64x64 B=1 254 /
// 336 / 412; 128->128 @320x320 B=2 800 / 880 / 710; 512->512 @80x80 B=2 669 / 729 / 645: the widest tile that still gives about
// four workgroups per CU with 128, or 1.5 with 64.
extern "C" int glare_conv2d_cout_tile(int B, int OH, int OW, int cout) {
  if (B <= 0 || OH <= 0 || OW <= 0 || cout <= 0) return 0;
  const int def = cout > 64 ? 128 : (cout > 32 ? 64 : 32);
  if (def < 128) return def;                       // only the 128-wide default was measured against narrower tiles
  const long long px = (long long)B * cdiv(OH, 8) * cdiv(OW, 32);
  if (px * cdiv(cout, 128) >= 1024) return 128;
  if (px * cdiv(cout, 64) >= 384) return 64;
  return 32;
}

// Rounding of a filter to the library's 16-bit format WITH ERROR FEEDBACK along the flattened (cin, ky, kx) axis of every output channel
// (round 6): q_i = round16(w_i + carry), carry = (w_i + carry) - q_i.  Every weight stays within one 16-bit ulp OF THE CHANNEL'S LARGEST
// WEIGHTS of its value (|error| <= 1/2 ulp + |carry|) and the SUM of an output channel's rounding errors stays below half such an ulp, where round-to-nearest leaves a random walk of ~sqrt(n) / 3.5 ulps:
// a filter's rounding error is the same perturbation at EVERY pixel (a coherent gain / offset error of the output channel), which is what
// stages D / E lose against the fp32 reference (tools/winograd_study.py: filters kept in fp32 0.0002-0.0018 dB, rounded 0.015-0.018;
// tools/filter_rounding_study.py: this rounding 0.001-0.008).  Output: fp32 values that ARE 16-bit numbers (the pack kernels' own
// rounding is then exact).  One thread per output channel; a pack-time kernel (once per weight set).
__global__ void filter_feedback_kernel(const float* __restrict__ w, float* __restrict__ out, int K, long long n) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  const float* src = w + (size_t)k * n;
  float* dst = out + (size_t)k * n;
  double carry = 0.0;
  auto step = [&](float v) {
    const double t = (double)v + carry;
    const float q = a2f(f2a((float)t));
    carry = t - (double)q;
    return q;
  };
  long long i = 0;
  if ((n & 3) == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {     // 16-B accesses, two in flight: the chain is the carry, not the loads
    for (; i + 8 <= n; i += 8) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(src + i), b = *reinterpret_cast<const f32x4*>(src + i + 4);
      f32x4 qa, qb;
#pragma unroll
      for (int e = 0; e < 4; ++e) qa[e] = step(a[e]);
#pragma unroll
      for (int e = 0; e < 4; ++e) qb[e] = step(b[e]);
      *reinterpret_cast<f32x4*>(dst + i) = qa;
      *reinterpret_cast<f32x4*>(dst + i + 4) = qb;
    }
  }
  for (; i < n; ++i) dst[i] = step(src[i]);
}

extern "C" int glare_filter_feedback_round_bf16(const float* w_oihw, float* out_oihw, int cout, long long elems_per_cout, glare_stream_t stream) {
  if (!w_oihw || !out_oihw || cout <= 0 || elems_per_cout <= 0) return GLARE_ERR_INVALID;
  hipLaunchKernelGGL(filter_feedback_kernel, dim3((unsigned)cdiv(cout, 64)), dim3(64), 0, (hipStream_t)stream, w_oihw, out_oihw, cout,
                     elems_per_cout);
  return glare_launch_status();
}

extern "C" int glare_conv2d_pack_weight(const float* w_oihw, int cout, int cin_total, int ksize, void* packed_bf16,
                                        glare_stream_t stream) {
  const long long total = glare_conv2d_packed_weight_elems(cout, cin_total, ksize);
  if (total < 0 || !w_oihw || !packed_bf16) return GLARE_ERR_INVALID;
  const Variant v = pick_variant(ksize, cout);
  const int kc = 16 * v.ksteps;
  const int stages = (cin_total + kc - 1) / kc;
  hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)((total + PACK_BLOCK_ELEMS - 1) / PACK_BLOCK_ELEMS)), dim3(256), 0, (hipStream_t)stream,
                     w_oihw, (a16_t*)packed_bf16, cout, cin_total, ksize, v.tn, v.ksteps, stages, total, 0, cin_total);
  return glare_launch_status();
}

extern "C" long long glare_conv2d_upsample_packed_weight_elems(int cout, int cin_total) {
  if (cout <= 0 || cin_total <= 0) return GLARE_ERR_INVALID;
  const Variant v = pick_variant(3, cout);
  const long long stages = (cin_total + 15) / 16, co_tiles = (cout + v.tn - 1) / v.tn;
  return 4 * co_tiles * stages * 4 * 2 * v.tn * 8;
}

extern "C" int glare_conv2d_pack_weight_upsample(const float* w_oihw, int cout, int cin_total, void* packed_bf16,
                                                 glare_stream_t stream) {
  const long long total = glare_conv2d_upsample_packed_weight_elems(cout, cin_total);
  if (total < 0 || !w_oihw || !packed_bf16) return GLARE_ERR_INVALID;
  const Variant v = pick_variant(3, cout);
  hipLaunchKernelGGL(pack_weight_subpix_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     w_oihw, (a16_t*)packed_bf16, cout, cin_total, v.tn, (cin_total + 15) / 16, (cout + v.tn - 1) / v.tn, total, 0);
  return glare_launch_status();
}

// The same image from the four phase filters themselves: w_phases fp32 [4][cout][cin][2][2], phase = 2 a + b the output sub-pixel, tap (r, c)
// the source pixel (the sums the entry above forms from the 3x3 filter) -- for callers that round the phase filters their own way.
extern "C" int glare_conv2d_pack_weight_upsample_phases(const float* w_phases, int cout, int cin_total, void* packed_bf16, glare_stream_t stream) {
  const long long total = glare_conv2d_upsample_packed_weight_elems(cout, cin_total);
  if (total < 0 || !w_phases || !packed_bf16) return GLARE_ERR_INVALID;
  const Variant v = pick_variant(3, cout);
  hipLaunchKernelGGL(pack_weight_subpix_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     w_phases, (a16_t*)packed_bf16, cout, cin_total, v.tn, (cin_total + 15) / 16, (cout + v.tn - 1) / v.tn, total, 1);
  return glare_launch_status();
}

extern "C" int glare_conv2d_pack_job_init(glare_pack_job* job, int kind, const float* w, int cout, int cin, int ksize, int dgrad_cout_padded,
                                          int cout_tile, void* packed) {
  if (!job || !w || !packed || cout <= 0 || cin <= 0) return GLARE_ERR_INVALID;
  job->w = w; job->out = packed; job->kind = kind; job->block_begin = 0;
  if (kind == GLARE_PACK_PLAIN_BF16) {   // the weight-stationary 1x1 kernel's filter: bf16 [cout][cin]
    job->cout = cout; job->cin = cin; job->ksize = 1; job->tn = job->ksteps = job->n_stages = 0; job->cin_real = cin;
    job->total = (long long)cout * cin;
    return GLARE_OK;
  }
  const bool dg = kind == GLARE_PACK_DGRAD;
  if (kind != GLARE_PACK_FORWARD && !dg) return GLARE_ERR_INVALID;
  if (dg && (dgrad_cout_padded < cout || (dgrad_cout_padded % 8))) return GLARE_ERR_INVALID;
  const int oc = dg ? cin : cout, ic = dg ? dgrad_cout_padded : cin;   // the packed conv's output / input channels
  const long long total = glare_conv2d_packed_weight_elems_tile(oc, ic, ksize, cout_tile);
  if (total <= 0) return GLARE_ERR_UNSUPPORTED;
  const Variant v = pick_variant(ksize, oc, cout_tile);
  const int kc = 16 * v.ksteps;
  job->cout = oc; job->cin = ic; job->ksize = ksize; job->tn = v.tn; job->ksteps = v.ksteps; job->n_stages = (ic + kc - 1) / kc;
  job->cin_real = dg ? cout : ic; job->total = total;
  return GLARE_OK;
}

extern "C" int glare_conv2d_pack_multi(const glare_pack_job* jobs_device, int n_jobs, long long total_blocks, glare_stream_t stream) {
  if (!jobs_device || n_jobs <= 0 || total_blocks <= 0 || total_blocks > 0x7fffffffLL) return GLARE_ERR_INVALID;
  hipLaunchKernelGGL(pack_weight_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, jobs_device, n_jobs);
  return glare_launch_status();
}

extern "C" int glare_conv2d_pack_weight_batched(const float* w_boihw, int batch, int cout, int cin, int ksize, int dgrad_cout_padded,
                                                int cout_tile, void* packed_bf16, glare_stream_t stream) {
  // `batch` filters of one shape in one launch; dgrad_cout_padded = 0: forward filters (glare_conv2d_pack_weight), > 0: the
  // data-gradient filters (glare_conv2d_pack_weight_dgrad with that padding); packed images are consecutive
  if (!w_boihw || !packed_bf16 || batch <= 0 || batch > 65535 || cout <= 0 || cin <= 0) return GLARE_ERR_INVALID;
  const bool dg = dgrad_cout_padded > 0;
  if (dg && (dgrad_cout_padded < cout || (dgrad_cout_padded % 8))) return GLARE_ERR_INVALID;
  const int oc = dg ? cin : cout, ic = dg ? dgrad_cout_padded : cin;   // the packed conv's output / input channels
  const long long total = glare_conv2d_packed_weight_elems_tile(oc, ic, ksize, cout_tile);
  if (total <= 0) return GLARE_ERR_UNSUPPORTED;
  const Variant v = pick_variant(ksize, oc, cout_tile);
  const int kc = 16 * v.ksteps;
  hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)((total + PACK_BLOCK_ELEMS - 1) / PACK_BLOCK_ELEMS), batch), dim3(256), 0, (hipStream_t)stream, w_boihw,
                     (a16_t*)packed_bf16, oc, ic, ksize, v.tn, v.ksteps, (ic + kc - 1) / kc, total, dg ? 1 : 0, dg ? cout : ic);
  return glare_launch_status();
}

extern "C" int glare_conv2d_pack_weight_dgrad(const float* w_oihw, int cout, int cin, int ksize, int cout_padded, void* packed_bf16,
                                              glare_stream_t stream) {
  // the data-gradient conv has cin output channels and cout_padded (>= cout, % 8 == 0) input channels
  if (!w_oihw || !packed_bf16 || cout <= 0 || cin <= 0 || cout_padded < cout || (cout_padded % 8)) return GLARE_ERR_INVALID;
  const long long total = glare_conv2d_packed_weight_elems(cin, cout_padded, ksize);
  if (total <= 0) return GLARE_ERR_UNSUPPORTED;
  const Variant v = pick_variant(ksize, cin);
  const int kc = 16 * v.ksteps;
  const int stages = (cout_padded + kc - 1) / kc;
  hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)((total + PACK_BLOCK_ELEMS - 1) / PACK_BLOCK_ELEMS)), dim3(256), 0, (hipStream_t)stream, w_oihw,
                     (a16_t*)packed_bf16, cin, cout_padded, ksize, v.tn, v.ksteps, stages, total, 1, cout);
  return glare_launch_status();
}

extern "C" int glare_conv2d_bf16(const glare_conv_desc* d, glare_stream_t stream_) {
  if (!d || !d->in || !d->weight_packed || !d->out) return GLARE_ERR_INVALID;
  if (d->B <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Cout <= 0) return GLARE_ERR_INVALID;
  if (d->ksize != 1 && d->ksize != 3) return GLARE_ERR_UNSUPPORTED;
  if (d->stride != 1 && d->stride != 2) return GLARE_ERR_UNSUPPORTED;
  if (d->stride == 2 && (d->ksize != 3 || d->upsample)) return GLARE_ERR_UNSUPPORTED;
  const bool subpix = d->upsample == 2;   // sub-pixel form: weights from glare_conv2d_pack_weight_upsample
  if (subpix && (d->ksize != 3 || d->stride != 1)) return GLARE_ERR_UNSUPPORTED;
  // 16-B channel chunks: every source's channel count / offset / pitch must be a multiple of 8
  if ((d->Cin % 8) || (d->in_pitch % 8) || (d->in_off % 8)) return GLARE_ERR_UNSUPPORTED;
  if (d->in2 && ((d->Cin2 % 8) || (d->in2_pitch % 8) || (d->in2_off % 8) || d->Cin2 <= 0)) return GLARE_ERR_UNSUPPORTED;
  if (d->out_mode < GLARE_OUT_NHWC_BF16 || d->out_mode > GLARE_OUT_PLANAR_BF16) return GLARE_ERR_INVALID;
  if (d->residual && d->out_mode > GLARE_OUT_NHWC_F32) return GLARE_ERR_UNSUPPORTED;
  if (d->in_off + d->Cin > d->in_pitch || d->out_off + d->Cout > d->out_pitch) return GLARE_ERR_INVALID;

  ConvParams p;
  p.in0 = (const a16_t*)d->in;
  p.in1 = (const a16_t*)d->in2;
  p.wpk = (const a16_t*)d->weight_packed;
  p.bias = d->bias;
  p.res = (const a16_t*)d->residual;
  p.out = d->out;
  p.B = d->B; p.H = d->H; p.W = d->W;
  p.IHs = (d->upsample && !subpix) ? 2 * d->H : d->H;   // sub-pixel form: the kernel works on the source grid
  p.IWs = (d->upsample && !subpix) ? 2 * d->W : d->W;
  if (d->stride == 2) {  // pad (0,1,0,1) then valid 3x3 s2 (encoder_decoder.py:71-73)
    p.OH = (p.IHs + 1 - 3) / 2 + 1;
    p.OW = (p.IWs + 1 - 3) / 2 + 1;
  } else {
    p.OH = p.IHs; p.OW = p.IWs;
  }
  if (d->k_wrap < 0 || d->k_wrap > 2 || (d->k_wrap == 2 && !(d->in2 && d->Cin2 == d->Cin))) return GLARE_ERR_INVALID;
  if (d->k_wrap == 2 && (d->ksize != 3 || d->gn_coef || d->upsample || d->Cin % 16)) return GLARE_ERR_UNSUPPORTED;
  p.k_wrap = d->k_wrap;
  p.Cin0 = d->Cin; p.Cin1 = d->in2 ? d->Cin2 : 0; p.CinTot = p.Cin0 + p.Cin1 + (p.k_wrap ? p.Cin0 : 0);
  p.p0 = d->in_pitch; p.o0 = d->in_off; p.p1 = d->in2_pitch; p.o1 = d->in2_off;
  p.Cout = d->Cout; p.opitch = d->out_pitch; p.ooff = d->out_off;
  p.rpitch = d->res_pitch; p.roff = d->res_off;
  p.upsample = subpix ? 0 : d->upsample; p.act = d->act; p.out_mode = d->out_mode;
  p.plane_pitch = d->plane_pitch > 0 ? d->plane_pitch : (long long)p.OH * p.OW;
  if (p.plane_pitch < (long long)p.OH * p.OW) return GLARE_ERR_INVALID;
  if (d->cout_tile != 0 && (d->upsample == 2 || (d->cout_tile != 32 && d->cout_tile != 64 && d->cout_tile != 128))) return GLARE_ERR_INVALID;
  const Variant v = pick_variant(d->ksize, d->Cout, d->cout_tile);
  if (d->gn_partial && v.tn == 32 && d->Cout > 32) return GLARE_ERR_UNSUPPORTED;   // the fused statistics' parts are per 4-row wave
                                                                                   // slab: the 32-wide tile's waves cover 2 rows
  const int kc = 16 * v.ksteps;
  // a stage must not straddle the two concatenated sources
  if ((p.in1 || p.k_wrap) && (p.Cin0 % kc)) return GLARE_ERR_UNSUPPORTED;
  if (p.k_wrap && (p.Cin1 % kc)) return GLARE_ERR_UNSUPPORTED;
  if (p.k_wrap && (subpix || d->upsample)) return GLARE_ERR_UNSUPPORTED;
  p.n_stages = (p.CinTot + kc - 1) / kc;
  {  // halo DMA through per-image buffer descriptors: 32-bit byte offsets inside one image of one source
    const long long b0 = (long long)p.H * p.W * p.p0 * 2, b1 = p.in1 ? (long long)p.H * p.W * p.p1 * 2 : 0;
    if (b0 >= 0x7ff00000LL || b1 >= 0x7ff00000LL) return GLARE_ERR_UNSUPPORTED;
    p.in0_bytes = (unsigned)b0;
    p.in1_bytes = (unsigned)b1;
    const long long br = d->residual ? (long long)p.OH * p.OW * p.rpitch * 2 : 0;   // one image of the residual (the L2 prefetch)
    p.res_bytes = (br > 0 && br < 0x7ff00000LL && !subpix) ? (unsigned)br : 0u;
    static const bool no_prefetch = getenv("GLARE_CONV_NO_RES_PREFETCH") != nullptr;
    if (no_prefetch) p.res_bytes = 0;
  }
  p.tiles_x = cdiv(p.OW, TW); p.tiles_y = 0; p.co_tiles = cdiv(p.Cout, v.tn);  // tiles_y / n_blocks: per variant, in launch()
  p.n_blocks = 0;
  p.gn_part = d->gn_partial;
  p.gn_nparts = 0;
  p.res_lo = (const a16_t*)d->residual_lo;
  p.out_lo = (a16_t*)d->out_lo;
  p.groups = d->groups > 1 ? d->groups : 1;
  p.g_in_step = d->group_in_step; p.g_out_step = d->group_out_step; p.g_bias_step = d->Cout; p.g_w_elems = 0;
  if (p.groups > 1) {
    if ((d->in2 && !(p.k_wrap && d->in2_pitch == d->in_pitch && d->Cin2 == d->Cin)) || d->residual || d->gn_partial || subpix || p.groups > 65535)
      return GLARE_ERR_UNSUPPORTED;
    if ((d->group_in_step % 8) || d->in_off + (long long)(p.groups - 1) * d->group_in_step + d->Cin > d->in_pitch ||
        d->out_off + (long long)(p.groups - 1) * d->group_out_step + d->Cout > d->out_pitch || d->group_out_step < 0 || d->group_in_step < 0)
      return GLARE_ERR_INVALID;
    p.g_w_elems = glare_conv2d_packed_weight_elems_tile(d->Cout, p.CinTot, d->ksize, d->cout_tile);
    if (d->in2 && d->in2_off + (long long)(p.groups - 1) * d->group_in_step + d->Cin2 > d->in2_pitch) return GLARE_ERR_INVALID;
  }
  p.gn_coef = d->gn_coef; p.gn_swish = d->gn_swish;
  if (p.gn_coef && (d->ksize != 3 || d->stride != 1 || d->in2 || p.k_wrap || d->upsample || p.groups > 1 || p.out_lo || d->in_off != 0 ||
                    d->Cin > 512))
    return GLARE_ERR_UNSUPPORTED;
  if (p.res_lo && !(p.res && p.out_lo)) return GLARE_ERR_INVALID;
  // 16-B records everywhere -> LDS-staged epilogue
  p.fast_epilogue = (d->out_mode == GLARE_OUT_NHWC_BF16) && !(p.Cout % 8) && !(p.opitch % 8) && !(p.ooff % 8) &&
                    (!p.res || (!(p.rpitch % 8) && !(p.roff % 8)));
  hipStream_t stream = (hipStream_t)stream_;

  if (p.out_lo) {   // hi / lo output (16-B records): the conditional encoder's residual stream, and every stored activation of the
                    // fp32-class convs (3x3: 128-wide tile; 1x1: 128- and 64-wide)
    if (subpix || d->upsample || !p.fast_epilogue) return GLARE_ERR_UNSUPPORTED;
    if (d->ksize == 1) return glare_conv_launch_k1(p, v.tn, true, stream);
    return d->stride == 2 ? glare_conv_launch_k3s2(p, v.tn, true, stream) : glare_conv_launch_k3s1(p, v.tn, true, stream);
  }
  if (subpix) {   // interleaved scatter lives in the LDS-staged epilogue only
    if (!p.fast_epilogue) return GLARE_ERR_UNSUPPORTED;
    return glare_conv_launch_k2(p, v.tn, stream);
  }
  if (d->ksize == 1) return glare_conv_launch_k1(p, v.tn, false, stream);
  if (d->stride == 2) return glare_conv_launch_k3s2(p, v.tn, false, stream);
  return glare_conv_launch_k3s1(p, v.tn, false, stream);
}

// Number of floats of the fused-GroupNorm partial buffer for a conv with this output geometry.
extern "C" long long glare_conv2d_gn_partial_elems(int B, int OH, int OW, int Cout) {
  if (B <= 0 || OH <= 0 || OW <= 0 || Cout <= 0) return GLARE_ERR_INVALID;
  const long long parts = (long long)cdiv(OW, TW) * cdiv(OH, 8) * 2;  // 8 x 32 tiles, 2 wave rows per tile
  return (long long)B * parts * (Cout / 4) * 2;
}

// The sub-pixel upsample conv (desc.upsample == 2) lays its partials on the SOURCE grid, four phases per tile.
extern "C" long long glare_conv2d_upsample_gn_partial_elems(int B, int H, int W, int Cout) {
  if (B <= 0 || H <= 0 || W <= 0 || Cout <= 0) return GLARE_ERR_INVALID;
  return 4 * glare_conv2d_gn_partial_elems(B, H, W, Cout);
}

extern "C" int glare_conv2d_upsample_gn_reduce(const float* gn_partial, float* stats_out, int B, int H, int W, int Cout,
                                               glare_stream_t stream) {
  if (!gn_partial || !stats_out || B <= 0 || Cout % 128) return GLARE_ERR_INVALID;
  const int parts = cdiv(W, TW) * cdiv(H, 8) * 2 * 4;
  hipLaunchKernelGGL(gn_part_reduce_kernel, dim3(B * 32), dim3(256), 0, (hipStream_t)stream, gn_partial, stats_out, parts, Cout);
  return glare_launch_status();
}

extern "C" int glare_conv2d_gn_reduce(const float* gn_partial, float* stats_out, int B, int OH, int OW, int Cout,
                                      glare_stream_t stream) {
  if (!gn_partial || !stats_out || B <= 0 || Cout % 128) return GLARE_ERR_INVALID;
  const int parts = cdiv(OW, TW) * cdiv(OH, 8) * 2;
  hipLaunchKernelGGL(gn_part_reduce_kernel, dim3(B * 32), dim3(256), 0, (hipStream_t)stream, gn_partial, stats_out, parts, Cout);
  return glare_launch_status();
}
