// Modulated deformable convolution (DCNv2), backward: all five gradients.
//
// Replaces modulated_deform_conv_cuda_backward (reference: ops/dcn/src/deform_conv_cuda.cpp:571-685),
// i.e. per sample: addmm_ (W^T . gO -> columns) -> col2im_coord kernel (deform_conv_cuda_kernel.cu:695-767)
// -> col2im kernel (:635-693, atomicAdd scatter) -> im2col kernel -> addmm_ (gO . col^T -> gW) -> addmm_ (gO . 1).
// As in the forward, the C*9*H*W `columns` buffer never exists; x is NHWC so a sampling corner's group
// channels are one contiguous run.  Two kernels with different decompositions:
//
//  dcn_bwd_data_kernel   (pixel-major: one workgroup = 64 output pixels, loop over (group, tap) stages)
//     cg[p, c]   = sum_co gO[p, co] * W[co, c, k]                 fp32 MFMA, gO tile resident in LDS
//     g_mask     = sum_c cg * bilinear(x)                         (kernel.cu:741-745)
//     g_offset   = sum_c cg * mask * d(bilinear)/d(h|w)           (kernel.cu:746-750, weights :528-568)
//     g_input   += cg * mask * corner weight                      (kernel.cu:677-691; fp32 atomics into NHWC)
//  dcn_bwd_weight_kernel (stage-major: one workgroup = one (group, tap) x a slice of the pixels)
//     gW[co, c, k] = sum_p gO[co, p] * mask * bilinear(x)[p, c]   fp32 MFMA with K = pixels; accumulators stay
//     in registers over the whole pixel slice; per-slice partials are reduced deterministically.
//  dcn_bwd_bias_kernel:  g_bias[co] = sum_{b,p} gO
//
// The reference accumulates grad_weight / grad_bias into caller-zeroed buffers and assigns the others
// (deform_conv.py:161-165); same here.  Like the reference, grad_input is order-nondeterministic (atomics).
#include "common.h"
#include "dcn_generic.h"

namespace {

constexpr int DB_THREADS = 256;
constexpr int DB_PIX = 64;
constexpr int DB_PITCH = DB_PIX + 1;

struct DcnBwdParams {
  const void* x;        // NHWC fp32 / bf16
  const float* offset;  // planar
  const float* mask;
  const float* gout;    // planar [B][Co][gout_plane]
  const float* wtb;     // data pass: [stage][Co][cpg]
  float* gx;            // NHWC fp32 [pix][C] (atomics) or NULL
  float* goff;          // planar like offset (dense planes of Ho*Wo)
  float* gmask;
  float* gw_partial;    // weight pass: [splits][Co][C][K]
  int B, C, H, W, Co, Ho, Wo;
  int kh, kw, sh, sw, ph, pw, dh, dw, dg, cpg;
  int xpitch, xoff;
  long long off_plane, mask_plane, off_bstride, mask_bstride, gout_plane;
  long long total_pix;
  int splits;
};

template <bool XBF16>
struct Corner8 {
  u32x4 v0, v1;
};
template <bool XBF16>
__device__ __forceinline__ void load8b(const void* base, long long elem_off, bool ok, Corner8<XBF16>& c) {
  const u32x4 z = {0u, 0u, 0u, 0u};
  if (XBF16) {
    c.v0 = ok ? *reinterpret_cast<const u32x4*>(reinterpret_cast<const a16_t*>(base) + elem_off) : z;
  } else {
    const float* p = reinterpret_cast<const float*>(base) + elem_off;
    c.v0 = ok ? *reinterpret_cast<const u32x4*>(p) : z;
    c.v1 = ok ? *reinterpret_cast<const u32x4*>(p + 4) : z;
  }
}
template <bool XBF16>
__device__ __forceinline__ float elem8(const Corner8<XBF16>& c, int e) {
  if (XBF16) return (e & 1) ? ahi(c.v0[e >> 1]) : alo(c.v0[e >> 1]);
  return __uint_as_float(e < 4 ? c.v0[e] : c.v1[e - 4]);
}

// Sampling geometry of one (pixel, group, tap): bilinear weights, validity and corner addresses.
struct Sample {
  float hh, hw, lh, lw, m;
  bool ok[4];
  long long pix[4];   // linear pixel index (b, h, w) of each corner
  int cb;             // first channel of this item's 8-channel chunk
};

template <bool XBF16, int ITEMS>
struct Gather {
  int px[ITEMS], ch[ITEMS], b[ITEMS], ho[ITEMS], wo[ITEMS];
  bool valid[ITEMS];
  Corner8<XBF16> cr[ITEMS][4];
  Sample sm[ITEMS];

  __device__ __forceinline__ void init(const DcnBwdParams& p, long long pix0, int tid) {
    const int nch = p.cpg >> 3;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      const int id = tid + i * DB_THREADS;
      ch[i] = id % nch;
      px[i] = id / nch;
      const long long gp = pix0 + px[i];
      valid[i] = gp < p.total_pix && px[i] < DB_PIX;
      const long long g2 = valid[i] ? gp : 0;
      wo[i] = (int)(g2 % p.Wo);
      ho[i] = (int)((g2 / p.Wo) % p.Ho);
      b[i] = (int)(g2 / ((long long)p.Wo * p.Ho));
    }
  }
  __device__ __forceinline__ void set_pixels(const DcnBwdParams& p, long long pix0, long long pix_end) {
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      const long long gp = pix0 + px[i];
      valid[i] = gp < pix_end;
      const long long g2 = valid[i] ? gp : 0;
      wo[i] = (int)(g2 % p.Wo);
      ho[i] = (int)((g2 / p.Wo) % p.Ho);
      b[i] = (int)(g2 / ((long long)p.Wo * p.Ho));
    }
  }
  __device__ __forceinline__ void issue(const DcnBwdParams& p, int s) {
    const int K = p.kh * p.kw;
    const int g = s / K, tap = s % K, ti = tap / p.kw, tj = tap % p.kw;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      const long long pin = (long long)ho[i] * p.Wo + wo[i];
      const float* op = p.offset + (long long)b[i] * p.off_bstride + ((long long)g * 2 * K + 2 * tap) * p.off_plane + pin;
      const float oh = valid[i] ? op[0] : 0.f;
      const float ow = valid[i] ? op[p.off_plane] : 0.f;
      const float m = valid[i] ? p.mask[(long long)b[i] * p.mask_bstride + ((long long)g * K + tap) * p.mask_plane + pin] : 0.f;
      const float h_im = (float)(ho[i] * p.sh - p.ph + ti * p.dh) + oh;
      const float w_im = (float)(wo[i] * p.sw - p.pw + tj * p.dw) + ow;
      const bool inside = valid[i] && h_im > -1.f && w_im > -1.f && h_im < (float)p.H && w_im < (float)p.W;
      const float hf = floorf(h_im), wf = floorf(w_im);
      const int h_low = (int)hf, w_low = (int)wf, h_high = h_low + 1, w_high = w_low + 1;
      Sample& q = sm[i];
      q.lh = h_im - hf; q.lw = w_im - wf; q.hh = 1.f - q.lh; q.hw = 1.f - q.lw; q.m = m;
      q.ok[0] = inside && h_low >= 0 && w_low >= 0;
      q.ok[1] = inside && h_low >= 0 && w_high <= p.W - 1;
      q.ok[2] = inside && h_high <= p.H - 1 && w_low >= 0;
      q.ok[3] = inside && h_high <= p.H - 1 && w_high <= p.W - 1;
      q.cb = g * p.cpg + ch[i] * 8;
      const long long row_lo = ((long long)b[i] * p.H + h_low) * p.W, row_hi = row_lo + p.W;
      q.pix[0] = row_lo + w_low;
      q.pix[1] = row_lo + w_high;
      q.pix[2] = row_hi + w_low;
      q.pix[3] = row_hi + w_high;
#pragma unroll
      for (int c = 0; c < 4; ++c) load8b<XBF16>(p.x, q.pix[c] * p.xpitch + p.xoff + q.cb, q.ok[c], cr[i][c]);
    }
  }
};

// ---------------------------------------------------------------------------------------------------
// data pass.  NTILE = cpg/32 column tiles; KSPLIT = 2 when only two MFMA tiles exist (cpg = 32): the
// contraction over Co is then split between wave pairs and the partials are summed at the read.
template <bool XBF16, int NTILE, int ITEMS>
__global__ __launch_bounds__(DB_THREADS) void dcn_bwd_data_kernel(const DcnBwdParams p) {
  constexpr int KSPLIT = NTILE == 1 ? 2 : 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* goT = reinterpret_cast<float*>(smem);               // [Co][DB_PITCH]
  float* cgT = goT + (size_t)p.Co * DB_PITCH;                // [KSPLIT][cpg][DB_PITCH]
  const int cpg = p.cpg, K = p.kh * p.kw, n_stages = p.dg * K;
  // grad_input scatter plan of the stage: per pixel the 4 corner pixels (-1: dropped) and corner weight x modulation
  long long* scat_pix = reinterpret_cast<long long*>(cgT + (size_t)KSPLIT * cpg * DB_PITCH + (((size_t)KSPLIT * cpg * DB_PITCH) & 1));
  float* scat_w = reinterpret_cast<float*>(scat_pix + DB_PIX * 4);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware tile order (dcn.hip): XCD j takes the j-th contiguous eighth of the pixel tiles, so the rows a tile gathers from
  // and scatters to stay in one L2
  unsigned tile = blockIdx.x;
  {
    const unsigned n = gridDim.x, q = n / 8, r = n % 8, xcd = tile % 8, k = tile / 8;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const long long pix0 = (long long)tile * DB_PIX;
  const long long hw_out = (long long)p.Ho * p.Wo;

  // gO tile -> LDS, transposed [co][pixel] (each plane row of 64 pixels is contiguous in HBM)
  for (int i = tid; i < p.Co * DB_PIX; i += DB_THREADS) {
    const int co = i / DB_PIX, px = i % DB_PIX;
    const long long gp = pix0 + px;
    float v = 0.f;
    if (gp < p.total_pix) v = p.gout[((gp / hw_out) * p.Co + co) * p.gout_plane + gp % hw_out];
    goT[co * DB_PITCH + px] = v;
  }

  Gather<XBF16, ITEMS> G;
  G.init(p, pix0, tid);
  const int mt = (NTILE == 1) ? (wave & 1) : (wave >> 1);
  const int nt = (NTILE == 1) ? 0 : (wave & 1);
  const int kpart = (NTILE == 1) ? (wave >> 1) : 0;
  const int k_len = p.Co / KSPLIT, k0 = kpart * k_len;
  const int khalf = lane >> 5, l31 = lane & 31;
  const int nch = cpg >> 3;
  __syncthreads();

  for (int s = 0; s < n_stages; ++s) {
    G.issue(p, s);  // corner loads fly during the MFMAs
    // ---- cg tile = gO[64 x Co] . Wb[Co x cpg]
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* a_src = goT + (size_t)k0 * DB_PITCH + mt * 32 + l31;
    const float* b_src = p.wtb + ((size_t)s * p.Co + k0) * cpg + nt * 32 + l31;
    // The B operand comes straight from global memory (L2): eight k-steps are requested together and the next eight while these
    // are contracted -- issued one by one in front of their MFMA, each of the Co / 2 steps exposed a full L2 round trip
    // (1.33 -> 0.41 ms at 1 x 128 x 128 x 256, 0.96 -> 0.39 ms at 1 x 256 x 256 x 128).
    constexpr int KB = 8;
    float bq[2][KB];
#pragma unroll
    for (int u = 0; u < KB; ++u) bq[0][u] = b_src[(size_t)(2 * u + khalf) * cpg];
    for (int ks0 = 0; ks0 < k_len / 2; ks0 += 2 * KB) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int base = ks0 + h * KB;
        if (base + KB < k_len / 2) {
#pragma unroll
          for (int u = 0; u < KB; ++u) bq[h ^ 1][u] = b_src[(size_t)(2 * (base + KB + u) + khalf) * cpg];
        }
        float aq[KB];
#pragma unroll
        for (int u = 0; u < KB; ++u) aq[u] = a_src[(2 * (base + u) + khalf) * DB_PITCH];
#pragma unroll
        for (int u = 0; u < KB; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[u], bq[h][u], acc, 0, 0, 0);
      }
    }
    float* dst = cgT + (size_t)kpart * cpg * DB_PITCH + (nt * 32 + l31) * DB_PITCH + mt * 32;
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[(r & 3) + 8 * (r >> 2) + 4 * khalf] = acc[r];
    __syncthreads();
    // ---- per (pixel, 8 channels): mask / offset gradients and the scatter into grad_input
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      const Sample& q = G.sm[i];
      const float w1 = q.hh * q.hw, w2 = q.hh * q.lw, w3 = q.lh * q.hw, w4 = q.lh * q.lw;
      float mval = 0.f, goh = 0.f, gow = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = G.ch[i] * 8 + e;
        float cg = cgT[c * DB_PITCH + G.px[i]];
        if (KSPLIT == 2) cg += cgT[(cpg + c) * DB_PITCH + G.px[i]];
        const float v1 = elem8<XBF16>(G.cr[i][0], e), v2 = elem8<XBF16>(G.cr[i][1], e);
        const float v3 = elem8<XBF16>(G.cr[i][2], e), v4 = elem8<XBF16>(G.cr[i][3], e);
        mval += cg * (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);
        const float top = cg * q.m;
        goh += (q.hw * (v3 - v1) + q.lw * (v4 - v2)) * top;   // d/dh: kernel.cu:543-553
        gow += (q.hh * (v2 - v1) + q.lh * (v4 - v3)) * top;   // d/dw: kernel.cu:554-564
      }
      if (p.gx && G.ch[i] == 0) {   // the scatter plan of this pixel (one lane per pixel writes it)
#pragma unroll
        for (int c = 0; c < 4; ++c) scat_pix[G.px[i] * 4 + c] = (G.valid[i] && q.ok[c]) ? q.pix[c] : -1;
        scat_w[G.px[i] * 4] = w1 * q.m; scat_w[G.px[i] * 4 + 1] = w2 * q.m; scat_w[G.px[i] * 4 + 2] = w3 * q.m; scat_w[G.px[i] * 4 + 3] = w4 * q.m;
      }
      {
      // reduce over the nch channel chunks of this pixel (adjacent lanes)
      for (int o = 1; o < nch; o <<= 1) {
        mval += __shfl_xor(mval, o, 64);
        goh += __shfl_xor(goh, o, 64);
        gow += __shfl_xor(gow, o, 64);
      }
      if (G.ch[i] == 0 && G.valid[i]) {
        const int g = s / K, tap = s % K;
        const long long pin = (long long)G.ho[i] * p.Wo + G.wo[i];
        float* go = p.goff + ((long long)G.b[i] * p.dg * 2 * K + (long long)g * 2 * K + 2 * tap) * hw_out + pin;
        go[0] = goh;
        go[hw_out] = gow;
        p.gmask[((long long)G.b[i] * p.dg * K + (long long)g * K + tap) * hw_out + pin] = mval;
      }
      }
    }
    if (p.gx) {
      // grad_input (kernel.cu:677-691), channel-major: lane = channel of the group, so one atomic instruction adds whole
      // 128-B / 256-B runs of one or two corner pixels.  (Issued from the gather layout -- a lane = 8 channels of a pixel, one
      // channel per instruction -- every instruction touched 16 pixels x 4 scattered dwords: 7.96 ms at 1 x 256 x 256 x 128 and
      // 4.57 ms at 1 x 128 x 128 x 256 for the whole backward with grad_input, against 0.91 / 0.96 ms without.)
      __syncthreads();                       // the scatter plan and every wave's cg tile are complete
      const int ppi = 64 / cpg;              // pixels per wave instruction (cpg = 32: 2, cpg = 64: 1)
      const int ch = lane % cpg, g = s / K;
      float* gxc = p.gx + (size_t)g * cpg + ch;
      for (int it = 0; it < DB_PIX / (4 * ppi); ++it) {
        const int px = (it * 4 + wave) * ppi + lane / cpg;
        float cg = cgT[ch * DB_PITCH + px];
        if (KSPLIT == 2) cg += cgT[(cpg + ch) * DB_PITCH + px];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const long long pc = scat_pix[px * 4 + c];
          if (pc >= 0) atomicAdd(gxc + pc * p.C, scat_w[px * 4 + c] * cg);
        }
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------
// weight pass.  grid = (stages, splits); MT = Co/32 row tiles, NTILE = cpg/32; tiles are dealt to waves.
template <bool XBF16, int MT, int NTILE, int ITEMS>
__global__ __launch_bounds__(DB_THREADS) void dcn_bwd_weight_kernel(const DcnBwdParams p) {
  constexpr int TILES = MT * NTILE, TPW = TILES / 4;  // tiles per wave
  static_assert(TILES % 4 == 0, "tile count is a multiple of the 4 waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* goT = reinterpret_cast<float*>(smem);             // [Co][DB_PITCH]  (A: co x pixel)
  float* colP = goT + (size_t)p.Co * DB_PITCH;             // [64][cpg + 1]   (B: pixel x c)
  const int cpg = p.cpg, K = p.kh * p.kw;
  const int s = blockIdx.x, split = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int khalf = lane >> 5, l31 = lane & 31;
  const long long hw_out = (long long)p.Ho * p.Wo;
  const long long per = ((p.total_pix + p.splits - 1) / p.splits + DB_PIX - 1) / DB_PIX * DB_PIX;
  const long long p_begin = (long long)split * per, p_end = min(p.total_pix, p_begin + per);

  f32x16 acc[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  Gather<XBF16, ITEMS> G;
  G.init(p, 0, tid);
  for (long long pix0 = p_begin; pix0 < p_end; pix0 += DB_PIX) {
    G.set_pixels(p, pix0, p_end);
    G.issue(p, s);
    __syncthreads();  // previous tile's MFMAs are done with goT / colP
    {  // gO tile -> LDS: a thread's pixel column is fixed (DB_THREADS % DB_PIX == 0), its rows are requested eight at a time
       // (one load per loop trip before: 1.28 -> 0.53 ms and 1.01 -> 0.42 ms for the two stage-3 shapes)
      const int px = tid % DB_PIX;
      const long long gp = pix0 + px;
      const bool ok = gp < p_end;
      const float* src = p.gout + ((ok ? gp / hw_out : 0) * p.Co) * p.gout_plane + (ok ? gp % hw_out : 0);
      constexpr int CSTEP = DB_THREADS / DB_PIX;     // output channels between a thread's consecutive rows
      for (int co0 = tid / DB_PIX; co0 < p.Co; co0 += 8 * CSTEP) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = ok ? src[(size_t)(co0 + u * CSTEP) * p.gout_plane] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) goT[(co0 + u * CSTEP) * DB_PITCH + px] = v[u];
      }
    }
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      const Sample& q = G.sm[i];
      const float w1 = q.hh * q.hw, w2 = q.hh * q.lw, w3 = q.lh * q.hw, w4 = q.lh * q.lw;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float val = w1 * elem8<XBF16>(G.cr[i][0], e) + w2 * elem8<XBF16>(G.cr[i][1], e) +
                          w3 * elem8<XBF16>(G.cr[i][2], e) + w4 * elem8<XBF16>(G.cr[i][3], e);
        colP[G.px[i] * (cpg + 1) + G.ch[i] * 8 + e] = G.valid[i] ? val * q.m : 0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int tile = wave * TPW + t, mt = tile / NTILE, nt = tile % NTILE;
      const float* a_src = goT + (size_t)(mt * 32 + l31) * DB_PITCH;
      const float* b_src = colP + nt * 32 + l31;
#pragma unroll 8
      for (int ks = 0; ks < DB_PIX / 2; ++ks) {
        const int kk = 2 * ks + khalf;
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_src[kk], b_src[kk * (cpg + 1)], acc[t], 0, 0, 0);
      }
    }
  }
  // partial gW[split][co][g*cpg + c][tap]
  const int g = s / K, tap = s % K;
  float* out = p.gw_partial + (size_t)split * p.Co * p.C * K;
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int tile = wave * TPW + t, mt = tile / NTILE, nt = tile % NTILE;
    const int c = g * cpg + nt * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
      out[((size_t)co * p.C + c) * K + tap] = acc[t][r];
    }
  }
}

__global__ void dcn_bwd_reduce_kernel(const float* __restrict__ partial, float* __restrict__ gw, long long n, int splits) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  int k = 0;
  for (; k + 7 < splits; k += 8) {   // eight loads in flight, added in order
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)(k + u) * n + i];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; k < splits; ++k) s += partial[(size_t)k * n + i];
  gw[i] += s;  // accumulates into the caller's (zeroed) buffer, as the reference's addmm_ does
}

__global__ __launch_bounds__(256) void dcn_bwd_bias_kernel(const float* __restrict__ gout, float* __restrict__ gb, int B, int Co,
                                                           long long hw_out, long long plane) {
  __shared__ float red[4];
  const int co = blockIdx.x;
  float s = 0.f;
  for (int b = 0; b < B; ++b) {
    const float* src = gout + ((size_t)b * Co + co) * plane;
    for (long long i = threadIdx.x; i < hw_out; i += 256) s += src[i];
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) gb[co] += (red[0] + red[1]) + (red[2] + red[3]);
}

// [Co][C][K] -> [stage][Co][cpg]
__global__ void dcn_bwd_pack_kernel(const float* __restrict__ w, float* __restrict__ wtb, int Co, int C, int K, int dg) {
  const long long total = (long long)Co * C * K;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int cpg = C / dg;
  const int c = (int)(i % cpg);
  long long t = i / cpg;
  const int co = (int)(t % Co);
  const int s = (int)(t / Co);
  const int g = s / K, tap = s % K;
  wtb[i] = w[((size_t)co * C + g * cpg + c) * K + tap];
}

constexpr int DB_SPLITS = 32;

}  // namespace

extern "C" size_t glare_mdcn_backward_workspace_bytes(int B, int C, int H, int W, int Co, int kh, int kw) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || Co <= 0 || kh <= 0 || kw <= 0) return 0;
  const size_t xs = (size_t)B * C * H * W, ws = (size_t)Co * C * kh * kw;
  return (2 * xs + ws * (1 + DB_SPLITS)) * sizeof(float) + 256;
}

// Drop-in for deform_conv_ext.modulated_deform_conv_backward (deform_conv_ext.cpp:127-147): reference
// layouts; grad_weight / grad_bias are accumulated into, the other gradients are overwritten.
extern "C" int glare_mdcn_backward_f32(const float* x, const float* offset, const float* mask, const float* weight,
                                       const float* grad_out, float* grad_input, float* grad_offset, float* grad_mask,
                                       float* grad_weight, float* grad_bias_or_null, int B, int C, int H, int W, int Co,
                                       int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int groups, int dg,
                                       void* workspace, size_t workspace_bytes, glare_stream_t stream_) {
  if (!x || !offset || !mask || !weight || !grad_out || !grad_offset || !grad_mask || !grad_weight) return GLARE_ERR_INVALID;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || Co <= 0 || kh <= 0 || kw <= 0 || sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0 ||
      dg <= 0 || groups <= 0 || C % dg)
    return GLARE_ERR_INVALID;
  const int cpg = C / dg;
  if (groups != 1 || (cpg != 32 && cpg != 64) || Co % 128 || Co > 256) {   // outside the MFMA kernels: general fp32 kernels, no workspace
    const int gs = glare_mdcn_generic_check(B, C, H, W, Co, kh, kw, sh, sw, dh, dw, groups, dg);
    if (gs != GLARE_OK) return gs;
    return glare_mdcn_generic_backward(x, offset, mask, weight, grad_out, grad_input, grad_offset, grad_mask, grad_weight,
                                       grad_bias_or_null, B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, groups, dg, (hipStream_t)stream_);
  }
  if (!workspace || workspace_bytes < glare_mdcn_backward_workspace_bytes(B, C, H, W, Co, kh, kw)) return GLARE_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  const int K = kh * kw;
  const size_t xs = (size_t)B * C * H * W, ws = (size_t)Co * C * K;
  float* x_nhwc = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
  float* gx_nhwc = x_nhwc + xs;
  float* wtb = gx_nhwc + xs;
  float* partial = wtb + ws;

  DcnBwdParams p;
  p.x = x_nhwc; p.offset = offset; p.mask = mask; p.gout = grad_out; p.wtb = wtb;
  p.gx = grad_input ? gx_nhwc : nullptr; p.goff = grad_offset; p.gmask = grad_mask; p.gw_partial = partial;
  p.B = B; p.C = C; p.H = H; p.W = W; p.Co = Co;
  p.Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1;
  p.Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1;
  if (p.Ho <= 0 || p.Wo <= 0) return GLARE_ERR_INVALID;
  p.kh = kh; p.kw = kw; p.sh = sh; p.sw = sw; p.ph = ph; p.pw = pw; p.dh = dh; p.dw = dw; p.dg = dg; p.cpg = cpg;
  p.xpitch = C; p.xoff = 0;
  const long long hw_out = (long long)p.Ho * p.Wo;
  p.off_plane = p.mask_plane = p.gout_plane = hw_out;
  p.off_bstride = (long long)dg * 2 * K * hw_out;
  p.mask_bstride = (long long)dg * K * hw_out;
  p.total_pix = (long long)B * hw_out;
  p.splits = DB_SPLITS;

  int rc = glare_nchw_to_nhwc(x, x_nhwc, B, C, (long long)H * W, C, 0, 0, stream_);
  if (rc != GLARE_OK) return rc;
  if (grad_input && hipMemsetAsync(gx_nhwc, 0, xs * sizeof(float), stream) != hipSuccess) return GLARE_ERR_LAUNCH;
  hipLaunchKernelGGL(dcn_bwd_pack_kernel, dim3((unsigned)((ws + 255) / 256)), dim3(256), 0, stream, weight, wtb, Co, C, K, dg);

  const unsigned blocks = (unsigned)((p.total_pix + DB_PIX - 1) / DB_PIX);
  const size_t lds_data = ((size_t)Co * DB_PITCH + (size_t)(cpg == 32 ? 2 : 1) * cpg * DB_PITCH + 2) * sizeof(float) +
                          (size_t)DB_PIX * 4 * (sizeof(long long) + sizeof(float));   // + the grad_input scatter plan
  const size_t lds_w = ((size_t)Co * DB_PITCH + (size_t)DB_PIX * (cpg + 1)) * sizeof(float);
  const dim3 wgrid(dg * K, DB_SPLITS);
#define DB_ATTR(k, bytes)                                                                                              \
  if ((bytes) > 64 * 1024 &&                                                                                           \
      hipFuncSetAttribute((const void*)(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)) != hipSuccess)   \
    return GLARE_ERR_LAUNCH;
  if (cpg == 32 && Co == 128) {
    DB_ATTR((dcn_bwd_data_kernel<false, 1, 1>), lds_data)
    hipLaunchKernelGGL((dcn_bwd_data_kernel<false, 1, 1>), dim3(blocks), dim3(DB_THREADS), lds_data, stream, p);
    hipLaunchKernelGGL((dcn_bwd_weight_kernel<false, 4, 1, 1>), wgrid, dim3(DB_THREADS), lds_w, stream, p);
  } else if (cpg == 64 && Co == 256) {
    DB_ATTR((dcn_bwd_data_kernel<false, 2, 2>), lds_data)
    DB_ATTR((dcn_bwd_weight_kernel<false, 8, 2, 2>), lds_w)
    hipLaunchKernelGGL((dcn_bwd_data_kernel<false, 2, 2>), dim3(blocks), dim3(DB_THREADS), lds_data, stream, p);
    hipLaunchKernelGGL((dcn_bwd_weight_kernel<false, 8, 2, 2>), wgrid, dim3(DB_THREADS), lds_w, stream, p);
  } else if (cpg == 32 && Co == 256) {
    DB_ATTR((dcn_bwd_data_kernel<false, 1, 1>), lds_data)
    DB_ATTR((dcn_bwd_weight_kernel<false, 8, 1, 1>), lds_w)
    hipLaunchKernelGGL((dcn_bwd_data_kernel<false, 1, 1>), dim3(blocks), dim3(DB_THREADS), lds_data, stream, p);
    hipLaunchKernelGGL((dcn_bwd_weight_kernel<false, 8, 1, 1>), wgrid, dim3(DB_THREADS), lds_w, stream, p);
  } else if (cpg == 64 && Co == 128) {
    DB_ATTR((dcn_bwd_data_kernel<false, 2, 2>), lds_data)
    hipLaunchKernelGGL((dcn_bwd_data_kernel<false, 2, 2>), dim3(blocks), dim3(DB_THREADS), lds_data, stream, p);
    hipLaunchKernelGGL((dcn_bwd_weight_kernel<false, 4, 2, 2>), wgrid, dim3(DB_THREADS), lds_w, stream, p);
  } else {
    return GLARE_ERR_UNSUPPORTED;
  }
#undef DB_ATTR
  hipLaunchKernelGGL(dcn_bwd_reduce_kernel, dim3((unsigned)((ws + 255) / 256)), dim3(256), 0, stream, partial, grad_weight,
                     (long long)ws, DB_SPLITS);
  if (grad_bias_or_null)
    hipLaunchKernelGGL(dcn_bwd_bias_kernel, dim3(Co), dim3(256), 0, stream, grad_out, grad_bias_or_null, B, Co, hw_out, hw_out);
  rc = glare_launch_status();
  if (rc != GLARE_OK) return rc;
  if (grad_input) return glare_nhwc_to_nchw(gx_nhwc, grad_input, B, C, (long long)H * W, C, 0, 0, stream_);
  return GLARE_OK;
}
