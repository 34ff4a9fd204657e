// Modulated deformable convolution (DCNv2), forward: bilinear gather fused with the weight
// contraction on the exact-fp32 MFMA -- no `columns` tensor.
//
// Replaces modulated_deform_conv_cuda_forward (reference: ops/dcn/src/deform_conv_cuda.cpp:490-569),
// i.e. per sample: at::zeros(columns) -> modulated_deformable_im2col_gpu_kernel
// (deform_conv_cuda_kernel.cu:571-633, bilinear :468-497) -> addmm_ -> bias.  The reference writes
// and re-reads a C*9*H*W fp32 `columns` buffer (1.2 GB per 400x600 image at the full-resolution
// warp) and gathers channel-planar (NCHW), one 4-B load per channel per corner.
//
// Here x is NHWC, so the cpg channels of a deformable group at one sampling corner are ONE contiguous
// run (64-256 B): 16-B loads, lanes along the channel axis.  A workgroup owns 64 output pixels x all
// output channels; for each (deformable group, tap) "stage" it samples a 64 x cpg tile into LDS
// (8-channel planes [c/8][pixel][8], so the MFMA A-fragment read is conflict-free) and contracts it with
// the [cpg x Co] weight slab.  The reference computes this in fp32 (DCNv2Pack casts everything to fp32,
// deformableDecoder_arch.py:143,550-551); the f32-input MFMA runs at 1/16 of the bf16 rate and made the
// kernel contraction-bound (8.9 ms per 8 images at full resolution), so the contraction uses the
// split-bf16 form: every fp32 operand v is carried as hi = bf16(v), lo = bf16(v - hi) and
//     a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi          (3 bf16 MFMAs, fp32 accumulate)
// which keeps 16 mantissa bits per operand (dropped term a_lo*b_lo <= 2^-16 relative): measured max
// error vs the fp32/double oracle ~1e-5 relative, inside the 1e-4 test tolerance, at 16/3 the speed.
// The gather of stage s+1 is issued before the MFMAs of stage s (registers hold the corners in
// flight); weights come straight from L2 as B-fragments (pre-split, fragment-ordered, 1 KB per load).
//   out[p, co] = bias[co] + sum_{g,k,c} W[co, g*cpg+c, k] * mask[g,k,p] * bilinear(x[., g*cpg+c], pos(p,g,k))
// Sampling semantics follow the reference exactly: a sample is 0 unless -1 < h < H and -1 < w < W,
// and each of the 4 corners is dropped individually when it lies outside the image.
#include <atomic>

#include <type_traits>

#include "common.h"
#include "dcn_generic.h"

namespace {

constexpr int DC_THREADS = 256;
constexpr int DC_PIX = 64;          // output pixels per workgroup

// the extra outputs of glare_mdcn_forward_nhwc_fused
struct MdcnFusedOut { float* out32; void* out16; float* sum_part; };

struct DcnParams {
  const void* x;          // NHWC, fp32 or bf16
  const float* offset;    // [B][dg*2K][off_plane]
  const float* mask;      // [B][dg*K][mask_plane]
  const float* wt;        // packed split-bf16 image, see dcn_pack_weight_kernel
  const float* bias;
  float* out;
  int B, C, H, W, Co, Ho, Wo;
  int kh, kw, sh, sw, ph, pw, dh, dw, dg, cpg;
  int xpitch, xoff;
  long long off_plane, mask_plane, off_bstride, mask_bstride;
  int mask_is_logit;
  int out_planar;         // 0: NHWC [p][opitch] at ooff; 1: NCHW planes of out_plane elements
  int opitch, ooff;
  long long out_plane;
  long long total_pix;
  unsigned x_bytes, off_bytes, mask_bytes, wt_bytes;   // buffer-descriptor extents (fast path only; each < 2^31)
  // round 6 (fast kernel only; both optional):
  a16_t* out16;           // the output as 16-bit NHWC [p][opitch] at ooff instead of fp32 `out` (rounded once from the fp32 accumulator + bias)
  float* sum_part;        // [B][ceil(Ho Wo / PIX)]: per pixel tile the sum of its fp32 outputs (before any rounding); the tiles are then cut PER
                          // IMAGE: mean(x_w) of `h + x_w mean(h) / mean(x_w)` (deformableDecoder_arch.py:567) without another pass over x_w
};

template <bool XBF16>
struct Corner {
  // 8 channels of one corner, kept as raw 16-B vectors while in flight
  u32x4 v0, v1;
};

template <bool XBF16>
__device__ __forceinline__ void load8(const void* base, long long elem_off, bool ok, Corner<XBF16>& c) {
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
__device__ __forceinline__ float elem(const Corner<XBF16>& c, int e) {
  if (XBF16) return (e & 1) ? ahi(c.v0[e >> 1]) : alo(c.v0[e >> 1]);   // x: the build's 16-bit activation format
  return __uint_as_float(e < 4 ? c.v0[e] : c.v1[e - 4]);
}

// NT: 32-wide output-channel tiles per wave (Co = 2 waves x NT x 32); ITEMS: (pixel, 8-channel) items
// per thread per stage = 64 * (cpg/8) / 256.
template <bool XBF16, int NT, int ITEMS>
__global__ __launch_bounds__(DC_THREADS) void dcn_fwd_kernel(const DcnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u32x4* colH = reinterpret_cast<u32x4*>(smem);  // [2 buffers][hi|lo][cpg/8][64 pixels] 16-B chunks
  const int cpg = p.cpg;
  const int K = p.kh * p.kw;
  const int n_stages = p.dg * K;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int nch = cpg >> 3;  // 8-channel chunks per group
  // XCD-aware tile order (see dcn_fwd_fast_kernel): XCD j takes the j-th contiguous eighth of the pixel tiles
  unsigned tile = blockIdx.x;
  {
    const unsigned n = gridDim.x, q = n / 8, r = n % 8, xcd = tile % 8, k = tile / 8;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const long long pix0 = (long long)tile * DC_PIX;

  // this thread's items: (pixel, chunk), pixel-major over chunks so 8-channel neighbours are lanes
  int it_px[ITEMS], it_ch[ITEMS];
  int it_b[ITEMS], it_ho[ITEMS], it_wo[ITEMS];
  bool it_ok[ITEMS];
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const int id = tid + i * DC_THREADS;
    it_ch[i] = id % nch;
    it_px[i] = id / nch;
    const long long gp = pix0 + it_px[i];
    it_ok[i] = gp < p.total_pix;
    const long long g2 = it_ok[i] ? gp : 0;
    it_wo[i] = (int)(g2 % p.Wo);
    it_ho[i] = (int)((g2 / p.Wo) % p.Ho);
    it_b[i] = (int)(g2 / ((long long)p.Wo * p.Ho));
  }

  f32x16 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  Corner<XBF16> cr[ITEMS][4];
  float cw[ITEMS][4], cm[ITEMS];

  auto gather_issue = [&](int s) {
    const int g = s / K, tap = s % K;
    const int ti = tap / p.kw, tj = tap % p.kw;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      const long long pin = (long long)it_ho[i] * p.Wo + it_wo[i];
      const float* op = p.offset + (long long)it_b[i] * p.off_bstride + ((long long)g * 2 * K + 2 * tap) * p.off_plane + pin;
      const float oh = it_ok[i] ? op[0] : 0.f;
      const float ow = it_ok[i] ? op[p.off_plane] : 0.f;
      float m = it_ok[i] ? p.mask[(long long)it_b[i] * p.mask_bstride + ((long long)g * K + tap) * p.mask_plane + pin] : 0.f;
      if (p.mask_is_logit) m = 1.0f / (1.0f + expf(-m));
      const float h_im = (float)(it_ho[i] * p.sh - p.ph + ti * p.dh) + oh;
      const float w_im = (float)(it_wo[i] * p.sw - p.pw + tj * p.dw) + ow;
      const bool inside = it_ok[i] && h_im > -1.f && w_im > -1.f && h_im < (float)p.H && w_im < (float)p.W;
      const float hf = floorf(h_im), wf = floorf(w_im);
      const int h_low = (int)hf, w_low = (int)wf, h_high = h_low + 1, w_high = w_low + 1;
      const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
      const bool ok1 = inside && h_low >= 0 && w_low >= 0;
      const bool ok2 = inside && h_low >= 0 && w_high <= p.W - 1;
      const bool ok3 = inside && h_high <= p.H - 1 && w_low >= 0;
      const bool ok4 = inside && h_high <= p.H - 1 && w_high <= p.W - 1;
      cw[i][0] = hh * hw; cw[i][1] = hh * lw; cw[i][2] = lh * hw; cw[i][3] = lh * lw;
      cm[i] = m;
      const long long cbase = (long long)p.xoff + g * cpg + it_ch[i] * 8;
      const long long row_lo = ((long long)it_b[i] * p.H + h_low) * p.W, row_hi = row_lo + p.W;
      load8<XBF16>(p.x, (row_lo + w_low) * p.xpitch + cbase, ok1, cr[i][0]);
      load8<XBF16>(p.x, (row_lo + w_high) * p.xpitch + cbase, ok2, cr[i][1]);
      load8<XBF16>(p.x, (row_hi + w_low) * p.xpitch + cbase, ok3, cr[i][2]);
      load8<XBF16>(p.x, (row_hi + w_high) * p.xpitch + cbase, ok4, cr[i][3]);
    }
  };
  auto gather_finish = [&](int buf) {
    u32x4* dst = colH + (size_t)buf * 2 * nch * DC_PIX;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      float val[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        // reference order: (w1*v1 + w2*v2 + w3*v3 + w4*v4) * mask   (kernel.cu:493-496,625)
        val[e] = (cw[i][0] * elem<XBF16>(cr[i][0], e) + cw[i][1] * elem<XBF16>(cr[i][1], e) +
                  cw[i][2] * elem<XBF16>(cr[i][2], e) + cw[i][3] * elem<XBF16>(cr[i][3], e)) * cm[i];
      }
      u32x4 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        hi[e] = pack_bf2(val[2 * e], val[2 * e + 1]);
        lo[e] = pack_bf2(val[2 * e] - bflo(hi[e]), val[2 * e + 1] - bfhi(hi[e]));
      }
      dst[it_ch[i] * DC_PIX + it_px[i]] = hi;
      dst[(nch + it_ch[i]) * DC_PIX + it_px[i]] = lo;
    }
  };

  gather_issue(0);
  gather_finish(0);
  const int arow = (lane & 31) + 32 * wm, khalf = lane >> 5;
  const u32x4* wq = reinterpret_cast<const u32x4*>(p.wt);  // [stage][cpg/8][hi | lo][Co][16 B]
  for (int s = 0; s < n_stages; ++s) {
    __syncthreads();                            // sample tile s visible; tile s-1 retired
    if (s + 1 < n_stages) gather_issue(s + 1);  // corners of the next stage fly during the MFMAs
    const u32x4* a_src = colH + (size_t)(s & 1) * 2 * nch * DC_PIX + arow;
    for (int ks = 0; ks < cpg / 16; ++ks) {
      const int kk = 2 * ks + khalf;
      const bf16x8 ah = __builtin_bit_cast(bf16x8, a_src[kk * DC_PIX]);
      const bf16x8 al = __builtin_bit_cast(bf16x8, a_src[(nch + kk) * DC_PIX]);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const u32x4* wp = wq + ((size_t)s * nch + kk) * 2 * p.Co + (wn * NT + j) * 32 + (lane & 31);
        const bf16x8 bh = __builtin_bit_cast(bf16x8, wp[0]);
        const bf16x8 bl = __builtin_bit_cast(bf16x8, wp[p.Co]);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[j], 0, 0, 0);
      }
    }
    if (s + 1 < n_stages) gather_finish((s + 1) & 1);
  }

  // epilogue: C/D layout col = lane&31 (co), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (pixel)
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int co = (wn * NT + j) * 32 + (lane & 31);
    const float bv = p.bias ? p.bias[co] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int prow = (r & 3) + 8 * (r >> 2) + 4 * khalf + 32 * wm;
      const long long gp = pix0 + prow;
      if (gp < p.total_pix) {
        const float v = acc[j][r] + bv;
        if (p.out_planar) {
          const long long hw = (long long)p.Ho * p.Wo;
          const long long b = gp / hw, pin = gp % hw;
          p.out[(b * p.Co + co) * p.out_plane + pin] = v;
        } else {
          p.out[gp * p.opitch + p.ooff + co] = v;
        }
      }
    }
  }
}


// ---- fast path: bf16 x, every tensor < 2 GB, kh*kw % 3 == 0 ----------------------------------------------------------
// Same algorithm and arithmetic as dcn_fwd_kernel; what changes is the instruction and memory-request count around it.
// Compiled-out ablations of the generic kernel at 8 x 420x620x128 (round 1): 1.35 ms with no loads at all, and each of
// the three load families (offset/mask triples, the 4 corners, the weight fragments) adds ~0.85 ms on top -- the cost is
// per vector-memory instruction and per exposed latency, not per byte.  Hence:
//   * every global access is a buffer load: 32-bit per-lane offsets computed once per item, the per-stage part (group,
//     tap, k-step) in the scalar offset; a dropped corner is an out-of-range offset (the descriptor returns zeros), so
//     the four corner loads are branch-free;
//   * the sampling plan -- corner byte offsets, bilinear weights, modulation -- is computed ONCE per (pixel, tap) by one
//     thread (coalesced 256-B loads of the offset / mask planes) and staged through LDS three taps ahead; the cpg/8 lanes
//     that gather one pixel's channels read it back (2 x 16 B + 4 B) instead of each redoing the coordinate arithmetic
//     and the sigmoid;
//   * weight fragments are prefetched into registers one k-step ahead;
//
// NO PACKED FP32 IN THIS KERNEL (dcn.hip is compiled with -fno-slp-vectorize, build.py; tests/test_build_invariants.py greps
// the ISA).  Written with v_pk_mul_f32 / v_pk_fma_f32 for the blend -- whose destinations the register allocator places on
// the B-fragment registers the MFMAs issued a few instructions earlier are still reading -- the kernel was NOT deterministic
// at the full-size shapes: 0.2 % (C = 128) to 12 % (C = 256) of the pixels, whole quarter-waves of the gather lanes,
// differed from the general kernel and from launch to launch, while every small-image parity test passed.  Plain (unpacked)
// VALU writes over the same registers at the same place are harmless, as they are in the general kernel.  What did NOT
// matter: LDS barriers, counted vs full s_waitcnt, buffer vs global loads, scalar-offset operands, out-of-range loads; loads
// of one wave do return in issue order (tools/probes/vmcnt_order_probe.hip).  The same blend in scalar fp32 is bit-stable
// over every shape of tools/determinism_check.py and just as fast (the kernel is not VALU-bound enough to notice).
//
// SINGLE = true (GLARE_MDCN_SINGLE_PASS): the blended sample and the filter are rounded ONCE to the library's 16-bit activation
// format and contracted by one MFMA per product -- the arithmetic of every other convolution on the path (16-bit operands, fp32
// accumulation) instead of the split form's 3 MFMAs; half the sample tile in LDS, half the fragment reads and weight loads.
// WN = waves along the output channels (2 or 4; 4 / WN along the pixels).  WN = 4 with MT = 2 (round 4, late): every wave covers all 64
// pixels and its own Co / 4 output channels, so no two waves of the workgroup fetch the same weight fragments -- with WN = 2 the two
// wave rows each load all of them, and the texture-path counters say that is what this kernel is bound by: TA_BUSY 79 % of the launch,
// 57 GB through the L1 per launch of which 37.5 GB weight fragments and 19 GB gathered corners (profiles/r04_pmc_dcn_ta.txt).
// Measured (fp16, one box, alternated): 2.90 -> 2.64 ms at C = 128, 2.23 -> 2.09 ms at C = 256; results bit-identical.
template <int NT, int NCH, int MT, bool SINGLE = false, int WN = 2>
__global__ __launch_bounds__(DC_THREADS) void dcn_fwd_fast_kernel(const DcnParams p) {
  constexpr int HL = SINGLE ? 1 : 2;                // 16-bit planes of the sample tile (hi | lo, or the one rounded value)
  constexpr int PIX = 32 * MT * (4 / WN);
  constexpr int ITEMS = PIX * NCH / DC_THREADS;
  constexpr int KSN = NCH / 2;                      // 16-channel MFMA k-steps per stage
  constexpr int CT = 3;                             // taps per staged sampling-plan chunk (K % 3 == 0 on this path)
  constexpr int NPASS = (CT * PIX + DC_THREADS - 1) / DC_THREADS;
  constexpr unsigned OOB = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u32x4* colH = reinterpret_cast<u32x4*>(smem);                    // [2 buffers][HL: hi|lo][NCH][PIX] 16-B chunks
  // sampling plan, [2 buffers] x { corner offsets u32x4 [CT][PIX] | corner weights f32x4 [CT][PIX] | mask f32 [CT][PIX] }
  constexpr int PLAN_BYTES = CT * PIX * 36;
  char* plan = smem + (size_t)2 * HL * NCH * PIX * 16;
  const int K = p.kh * p.kw;
  const int n_stages = p.dg * K, n_chunks = n_stages / CT;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  // XCD-aware order: workgroups are dispatched round-robin over the 8 XCDs, so XCD j takes the j-th contiguous eighth of the
  // pixel tiles -- vertically adjacent tiles (which gather from the same rows of x) then share one L2 instead of eight
  unsigned tile = blockIdx.x;
  {
    const unsigned n = gridDim.x, q = n / 8, r = n % 8, xcd = tile % 8, k = tile / 8;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  // sum_part (round 6): tiles are cut PER IMAGE (the last tile of an image is ragged) so that a tile's sum belongs to one image and an
  // image's tiles -- hence its mean -- are the same whatever batch it sits in; otherwise tiles run over the flattened B * Ho * Wo pixels
  const unsigned hw = (unsigned)p.Ho * p.Wo;
  unsigned pix0 = tile * (unsigned)PIX, total = (unsigned)p.total_pix;
  if (p.sum_part) {
    const unsigned tpi = (hw + PIX - 1) / PIX, b = tile / tpi;
    pix0 = b * hw + (tile - b * tpi) * (unsigned)PIX;
    total = (b + 1u) * hw;                                   // rows at or beyond it are not this tile's (plan: dropped; stores: skipped)
  }

  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t offr =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.offset), 0, (int)p.off_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t mr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.mask), 0, (int)p.mask_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wt), 0, (int)p.wt_bytes, 0x00020000);

  // ---- sampling plan: one thread per (tap of the chunk, pixel) turns (offset_h, offset_w, mask) into the four corner
  // byte offsets (out-of-range when the corner is dropped), the four bilinear weights and the modulation, ONCE per
  // pixel and tap; the NCH lanes that gather the channels of that pixel read the plan back from LDS ----
  const int spx = tid % PIX;
  unsigned s_ovo, s_mvo, s_xvo;
  float s_h0, s_w0;
  bool s_ok;
  {
    const unsigned gp = pix0 + spx;
    s_ok = gp < total;
    const unsigned g2 = s_ok ? gp : 0u;
    const unsigned wo = g2 % (unsigned)p.Wo, t = g2 / (unsigned)p.Wo;
    const unsigned ho = t % (unsigned)p.Ho, b = t / (unsigned)p.Ho;
    const unsigned pin = ho * p.Wo + wo;
    s_ovo = (b * (unsigned)p.off_bstride + pin) * 4u;
    s_mvo = (b * (unsigned)p.mask_bstride + pin) * 4u;
    s_xvo = ((b * (unsigned)p.H * p.W) * p.xpitch + p.xoff) * 2u;
    s_h0 = (float)((int)ho * p.sh - p.ph);
    s_w0 = (float)((int)wo * p.sw - p.pw);
  }
  const unsigned px_b = (unsigned)p.xpitch * 2u, row_b = (unsigned)p.W * px_b;
  float s_oh[NPASS], s_ow[NPASS], s_m[NPASS];
  auto plan_load = [&](int c) {
#pragma unroll
    for (int q = 0; q < NPASS; ++q) {
      const int tl = __builtin_amdgcn_readfirstlane((tid + q * DC_THREADS) / PIX);   // tap within the chunk, wave-uniform
      if (tl < CT) {
        const int s = c * CT + tl, g = s / K, tap = s - g * K;
        const unsigned so = (unsigned)((g * 2 * K + 2 * tap) * p.off_plane) * 4u;
        s_oh[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(offr, s_ovo, so, 0));
        s_ow[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(offr, s_ovo, so + (unsigned)p.off_plane * 4u, 0));
        s_m[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(mr, s_mvo, (unsigned)((g * K + tap) * p.mask_plane) * 4u, 0));
      }
    }
  };
  auto plan_write = [&](int c) {
    char* dst = plan + (size_t)(c & 1) * PLAN_BYTES;
#pragma unroll
    for (int q = 0; q < NPASS; ++q) {
      const int tl = __builtin_amdgcn_readfirstlane((tid + q * DC_THREADS) / PIX);
      if (tl < CT) {
        const int s = c * CT + tl, tap = s % K;
        const int ti = tap / p.kw, tj = tap - ti * p.kw;
        const float h_im = (s_h0 + (float)(ti * p.dh)) + s_oh[q];
        const float w_im = (s_w0 + (float)(tj * p.dw)) + s_ow[q];
        const bool inside = s_ok && h_im > -1.f && w_im > -1.f && h_im < (float)p.H && w_im < (float)p.W;
        const float hf = floorf(h_im), wf = floorf(w_im);
        const int h_low = (int)hf, w_low = (int)wf;
        const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
        const bool top = inside && h_low >= 0, bot = inside && h_low + 1 <= p.H - 1;
        const bool lft = w_low >= 0, rgt = w_low + 1 <= p.W - 1;
        const unsigned o1 = s_xvo + (unsigned)(h_low * p.W + w_low) * px_b;
        u32x4 vo;
        vo[0] = (top && lft) ? o1 : OOB;
        vo[1] = (top && rgt) ? o1 + px_b : OOB;
        vo[2] = (bot && lft) ? o1 + row_b : OOB;
        vo[3] = (bot && rgt) ? o1 + row_b + px_b : OOB;
        float m = s_m[q];
        if (p.mask_is_logit) m = 1.0f / (1.0f + expf(-m));
        const int e = tl * PIX + spx;
        reinterpret_cast<u32x4*>(dst)[e] = vo;
        reinterpret_cast<f32x4*>(dst + CT * PIX * 16)[e] = f32x4{hh * hw, hh * lw, lh * hw, lh * lw};
        reinterpret_cast<float*>(dst + CT * PIX * 32)[e] = m;
      }
    }
  };

  // ---- gather items: (pixel, 8-channel chunk), chunk fastest so the lanes of one pixel read one contiguous run ----
  int it_lds[ITEMS], it_px[ITEMS];
  unsigned it_cb[ITEMS];
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const int id = tid + i * DC_THREADS;
    const int ch = id % NCH, px = id / NCH;
    // tile layout [c/8][pixel] in 16-B chunks, pixel index XOR-swizzled by the chunk: the NCH lanes of one pixel write chunks that
    // are a multiple of 1 KB apart -- the same banks, an NCH-way conflict on every tile write (SQ_LDS_BANK_CONFLICT was 50-70 %
    // of the LDS-active cycles); with px ^ (ch * 16 / NCH) the 16 lanes of a write phase cover 16 different bank groups, and
    // the MFMA fragment read of chunk kk (32 consecutive rows) stays a permutation inside each 16-row block
    it_lds[i] = ch * PIX + (px ^ (ch * (16 / NCH)));
    it_px[i] = px;
    it_cb[i] = ch * 16u;     // out-of-range offsets stay out of range: 2^31 + 16*ch + the group offset < 2^32
  }

  f32x16 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][j][r] = 0.f;

  struct CornerSet {       // one stage's samples in flight: raw corners, bilinear weights, modulation
    u32x4 cr[ITEMS][4];
    f32x4 cw[ITEMS];
    float cm[ITEMS];
  };
  CornerSet cs0;
  auto gather_issue = [&](int s, CornerSet& cs) {
    u32x4 (&cr)[ITEMS][4] = cs.cr;
    f32x4 (&cw)[ITEMS] = cs.cw;
    float (&cm)[ITEMS] = cs.cm;
    const int c = s / CT, tl = s - c * CT;
    const unsigned sx = (unsigned)((s / K) * p.cpg) * 2u;
    const char* src = plan + (size_t)(c & 1) * PLAN_BYTES;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      const int e = tl * PIX + it_px[i];
      const u32x4 vo = reinterpret_cast<const u32x4*>(src)[e];
      cw[i] = reinterpret_cast<const f32x4*>(src + CT * PIX * 16)[e];
      cm[i] = reinterpret_cast<const float*>(src + CT * PIX * 32)[e];
      cr[i][0] = __builtin_amdgcn_raw_buffer_load_b128(xr, vo[0] + it_cb[i], sx, 0);
      cr[i][1] = __builtin_amdgcn_raw_buffer_load_b128(xr, vo[1] + it_cb[i], sx, 0);
      cr[i][2] = __builtin_amdgcn_raw_buffer_load_b128(xr, vo[2] + it_cb[i], sx, 0);
      cr[i][3] = __builtin_amdgcn_raw_buffer_load_b128(xr, vo[3] + it_cb[i], sx, 0);
    }
  };
  auto gather_finish = [&](int buf, const CornerSet& cs) {
    const u32x4 (&cr)[ITEMS][4] = cs.cr;
    const f32x4 (&cw)[ITEMS] = cs.cw;
    const float (&cm)[ITEMS] = cs.cm;
    u32x4* dst = colH + (size_t)buf * HL * NCH * PIX;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      const float m = cm[i];
      u32x4 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        // reference order: (w1*v1 + w2*v2 + w3*v3 + w4*v4) * mask   (kernel.cu:493-496,625), on (even, odd) channel pairs.
        // SCALAR fp32 on purpose, see the note on packed fp32 in the kernel's header comment.
        float v0 = alo(cr[i][0][e]) * cw[i][0], v1 = ahi(cr[i][0][e]) * cw[i][0];
        v0 = __builtin_fmaf(alo(cr[i][1][e]), cw[i][1], v0); v1 = __builtin_fmaf(ahi(cr[i][1][e]), cw[i][1], v1);
        v0 = __builtin_fmaf(alo(cr[i][2][e]), cw[i][2], v0); v1 = __builtin_fmaf(ahi(cr[i][2][e]), cw[i][2], v1);
        v0 = __builtin_fmaf(alo(cr[i][3][e]), cw[i][3], v0); v1 = __builtin_fmaf(ahi(cr[i][3][e]), cw[i][3], v1);
        v0 *= m; v1 *= m;
        asm volatile("" : "+v"(v0), "+v"(v1));   // keeps the pair out of the vectoriser's hands
        if constexpr (SINGLE) {
          hi[e] = pack_a2(v0, v1);
        } else {
          hi[e] = pack_bf2(v0, v1);
          float r0 = v0 - bflo(hi[e]), r1 = v1 - bfhi(hi[e]);
          asm volatile("" : "+v"(r0), "+v"(r1));
          lo[e] = pack_bf2(r0, r1);
        }
      }
      dst[it_lds[i]] = hi;
      if constexpr (!SINGLE) dst[NCH * PIX + it_lds[i]] = lo;
    }
  };

  // ---- weight fragments: [stage][c/8][hi | lo][Co][16 B], one (hi, lo) pair per (k-step, N tile): every load reads 2 x 512
  // contiguous bytes ----  (SINGLE: [stage][c/8][Co][16 B], one fragment per (k-step, N tile))
  const int khalf = lane >> 5;
  constexpr unsigned WP = SINGLE ? 1u : 2u;         // 16-B planes per (8 channels, output channel) of the packed filter
  const unsigned wvo = (unsigned)(khalf * WP * p.Co + wn * NT * 32 + (lane & 31)) * 16u;
  const unsigned wlo = (unsigned)p.Co * 16u;        // from a chunk's hi plane to its lo plane
  auto load_b = [&](int s, int ks, u32x4 (&dst)[NT][HL]) {
    const unsigned ws = (unsigned)((s * NCH + 2 * ks) * p.Co) * (16u * WP);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      dst[j][0] = __builtin_amdgcn_raw_buffer_load_b128(wr, wvo + j * 512, ws, 0);
      if constexpr (!SINGLE) dst[j][HL - 1] = __builtin_amdgcn_raw_buffer_load_b128(wr, wvo + j * 512, ws + wlo, 0);   // (the plane step rides in the scalar offset)
    }
  };
  auto mfma_step = [&](const u32x4* a_src, int ks, const u32x4 (&b)[NT][HL]) {
    const int kk = 2 * ks + khalf;
    if constexpr (SINGLE) {
      a16x8 a[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int sw = ((lane & 31) ^ (kk * (16 / NCH))) - (lane & 31);
        a[m] = __builtin_bit_cast(a16x8, a_src[kk * PIX + 32 * m + sw]);
      }
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int m = 0; m < MT; ++m)
          acc[m][j] = mfma_a16_32x32x16(a[m], __builtin_bit_cast(a16x8, b[j][0]), acc[m][j], 0, 0, 0);
    } else {
    bf16x8 ah[MT], al[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int sw = ((lane & 31) ^ (kk * (16 / NCH))) - (lane & 31);   // swizzled row of this lane, relative to a_src
      ah[m] = __builtin_bit_cast(bf16x8, a_src[kk * PIX + 32 * m + sw]);
      al[m] = __builtin_bit_cast(bf16x8, a_src[(NCH + kk) * PIX + 32 * m + sw]);
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const bf16x8 bh = __builtin_bit_cast(bf16x8, b[j][0]);
      const bf16x8 bl = __builtin_bit_cast(bf16x8, b[j][HL - 1]);
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[m], bh, acc[m][j], 0, 0, 0);
        acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bl, acc[m][j], 0, 0, 0);
        acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bh, acc[m][j], 0, 0, 0);
      }
    }
    }
  };

  u32x4 bA[NT][HL], bB[NT][HL];

  // one stage: barrier; the gather of stage s+1 is issued; the MFMAs of stage s (the weight fragments of the next
  // k-step always in flight); the samples of stage s+1 are blended into the other tile.  Measured and dropped: weight
  // fragments a whole stage ahead (+32 VGPRs: 3.82 ms vs 3.38 ms at 8 x 420x620x128 -- occupancy 2 instead of 3) and
  // the gather two stages ahead with a second corner set (3.81 ms, same reason): this kernel hides latency with waves.
  auto stage_body = [&](int s) {
    __syncthreads();                            // sample tile s (and any staged plan chunk) visible; tile s-1 retired
    const bool more = s + 1 < n_stages;
    const int c = s / CT, tl = s - c * CT;
    if (more) gather_issue(s + 1, cs0);
    if (tl == 0 && c + 1 < n_chunks) plan_load(c + 1);
    const u32x4* a_src = colH + (size_t)(s & 1) * HL * NCH * PIX + (lane & 31) + 32 * MT * wm;
    // KSN is even: the k-steps alternate between the two register sets and every stage starts on bA
#pragma unroll
    for (int ks = 0; ks < KSN; ks += 2) {
      load_b(s, ks + 1, bB);
      mfma_step(a_src, ks, bA);
      if (ks + 2 < KSN) load_b(s, ks + 2, bA);
      else if (more) load_b(s + 1, 0, bA);
      mfma_step(a_src, ks + 1, bB);
    }
    if (more) gather_finish((s + 1) & 1, cs0);
    if (tl == CT - 2 && c + 1 < n_chunks) plan_write(c + 1);
  };

  plan_load(0);
  plan_write(0);
  load_b(0, 0, bA);
  __syncthreads();
  gather_issue(0, cs0);
  gather_finish(0, cs0);
  for (int s = 0; s < n_stages; ++s) stage_body(s);

  // epilogue: C/D layout col = lane&31 (co), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (pixel)
  float s0 = 0.f;                                       // sum of this lane's outputs (sum_part)
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const unsigned gp_t = pix0 + 32 * (MT * wm + m) + 4 * khalf;   // first row of this lane in the tile
    const unsigned b_t = gp_t / hw, pin_t = gp_t % hw;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int co = (wn * NT + j) * 32 + (lane & 31);
      const float bv = p.bias ? p.bias[co] : 0.f;
      if (p.out16) {
        // 16-bit NHWC: lanes (co, co + 1) exchange one value per register pair so that every lane stores one packed 4-B word --
        // even lanes row 2t, odd lanes row 2t + 1 (the conv kernels' phase-1 trick, without the LDS slab: 64-B runs per row)
        const int odd = lane & 1;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const float a = acc[m][j][2 * t] + bv, c = acc[m][j][2 * t + 1] + bv;
          const unsigned ra = ((2 * t) & 3) + 8 * ((2 * t) >> 2), rc = ((2 * t + 1) & 3) + 8 * ((2 * t + 1) >> 2);
          s0 += gp_t + ra < total ? a : 0.f;
          s0 += gp_t + rc < total ? c : 0.f;
          const float send = odd ? a : c;
          const float recv = __shfl_xor(send, 1, 64);
          const unsigned gp = gp_t + (odd ? rc : ra);
          const uint32_t wv = odd ? pack_a2(recv, c) : pack_a2(a, recv);
          if (gp < total) *reinterpret_cast<uint32_t*>(p.out16 + (size_t)gp * p.opitch + p.ooff + (co & ~1)) = wv;
        }
        continue;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned dr = (r & 3) + 8 * (r >> 2);
        const unsigned gp = gp_t + dr;
        if (gp < total) {
          const float v = acc[m][j][r] + bv;
          s0 += v;
          if (p.out_planar) {
            unsigned b = b_t, pin = pin_t + dr;
            while (pin >= hw) { pin -= hw; ++b; }
            p.out[((size_t)b * p.Co + co) * p.out_plane + pin] = v;
          } else {
            p.out[(size_t)gp * p.opitch + p.ooff + co] = v;
          }
        }
      }
    }
  }
  if (p.sum_part) {      // fixed order: lane tree, then the waves one after the other -- deterministic
    s0 = wave_sum(s0);
    __syncthreads();     // every wave is done with the sample tiles: the LDS is free
    float* red = reinterpret_cast<float*>(smem);
    if (lane == 0) red[wave] = s0;
    __syncthreads();
    if (tid == 0) {
      float a = 0.f;
      for (int wv = 0; wv < DC_THREADS / 64; ++wv) a += red[wv];
      p.sum_part[tile] = a;
    }
  }
}

// The producer / consumer form of this kernel (consumer waves own the accumulators, producer waves the gather with 2-3 corner sets in
// flight; bit-identical results, 45 parity / determinism tests green) measured 4.0-4.2 ms instead of 2.9 (C = 128) and 2.9-3.1 instead
// of 2.25 (C = 256), the same for a gather depth of 2 and of 3: the stage time is not exposed gather LATENCY that more loads in flight
// could hide (round 4).

// [Co][C][kh][kw] fp32 (reference layout) -> split-bf16 B-fragment image
// [stage = g*K + tap][c/8][hi | lo][Co][8 bf16]  (same byte count as the fp32 filter).  Round 5: the hi and the lo fragments of a
// chunk are two PLANES of Co contiguous 16-B records, not interleaved 32-B records -- a B-fragment load of 32 consecutive output channels
// then reads 512 contiguous bytes (4 lines) instead of every other 16 B of 1 KB (8 lines, half of each unused).
__global__ void dcn_pack_weight_kernel(const float* __restrict__ w, bf16_t* __restrict__ wt, int Co, int C, int K, int dg) {
  const long long total = (long long)Co * C * K;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int cpg = C / dg, nch = cpg / 8;
  const int e = (int)(i % 8);
  long long t = i / 8;
  const int co = (int)(t % Co);
  t /= Co;
  const int kk = (int)(t % nch);
  const int s = (int)(t / nch);
  const int g = s / K, tap = s % K;
  const float v = w[((size_t)co * C + g * cpg + kk * 8 + e) * K + tap];
  const bf16_t hi = f2bf(v);
  const size_t base = ((((size_t)s * nch + kk) * 2) * Co + co) * 8;     // the hi plane of this (stage, chunk); its lo plane Co records further
  wt[base + e] = hi;
  wt[base + (size_t)Co * 8 + e] = f2bf(v - bf2f(hi));
}

// the single-pass image: [stage][c/8][Co][8 x a16], the filter rounded once to the library's activation format
__global__ void dcn_pack_weight_single_kernel(const float* __restrict__ w, a16_t* __restrict__ wt, int Co, int C, int K, int dg) {
  const long long total = (long long)Co * C * K;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int cpg = C / dg, nch = cpg / 8;
  const int e = (int)(i % 8);
  long long t = i / 8;
  const int co = (int)(t % Co);
  t /= Co;
  const int kk = (int)(t % nch);
  const int s = (int)(t / nch);
  const int g = s / K, tap = s % K;
  wt[(((size_t)s * nch + kk) * Co + co) * 8 + e] = f2a(w[((size_t)co * C + g * cpg + kk * 8 + e) * K + tap]);
}

template <bool XBF16>
int launch_dcn(const DcnParams& p, hipStream_t stream) {
  const int nt = p.Co / 64;           // 2 waves along Co
  const int items = DC_PIX * (p.cpg / 8) / DC_THREADS;
  const size_t lds = (size_t)2 * 2 * (p.cpg / 8) * DC_PIX * 16;
  const unsigned blocks = (unsigned)((p.total_pix + DC_PIX - 1) / DC_PIX);
#define DCN_CASE(NT_, IT_)                                                                            \
  if (nt == NT_ && items == IT_) {                                                                    \
    hipLaunchKernelGGL((dcn_fwd_kernel<XBF16, NT_, IT_>), dim3(blocks), dim3(DC_THREADS), lds, stream, p); \
    return glare_launch_status();                                                                     \
  }
  DCN_CASE(1, 1) DCN_CASE(2, 1) DCN_CASE(4, 1) DCN_CASE(1, 2) DCN_CASE(2, 2) DCN_CASE(4, 2)
#undef DCN_CASE
  return GLARE_ERR_UNSUPPORTED;
}

// MT = 1 everywhere: 128-pixel workgroups (MT = 2) halve the weight-fragment traffic but run at occupancy 2 and measured
// 3.35 ms vs 3.32 ms (C = 128) and 2.95 ms vs 2.42 ms (C = 256) at 8 images -- the kernel needs the waves.
int launch_dcn_fast(const DcnParams& p, bool single, hipStream_t stream) {
  const int nt = p.Co / 64, nch = p.cpg / 8;
  if (single && nt == 2 && nch == 4) {
    // 128-pixel workgroups (MT = 2) for the single-pass form at C = Co = 128: half the weight-fragment loads per pixel at 140
    // VGPRs (3 waves / SIMD): 2.13 -> 1.94 ms at 8 x 420 x 620.  Not at C = 256 (1.59 -> 2.15 ms: occupancy 1), and not for the
    // split form (measured in round 2, see above launch_dcn_fast).
    const int pix2 = 128;
    const size_t lds2 = (size_t)2 * nch * pix2 * 16 + (size_t)2 * 3 * pix2 * 36;
    const long long hw2 = (long long)p.Ho * p.Wo;
    const unsigned blocks2 = p.sum_part ? (unsigned)(p.B * ((hw2 + pix2 - 1) / pix2)) : (unsigned)((p.total_pix + pix2 - 1) / pix2);
    hipLaunchKernelGGL((dcn_fwd_fast_kernel<2, 4, 2, true>), dim3(blocks2), dim3(DC_THREADS), lds2, stream, p);
    return glare_launch_status();
  }
  const int pix = 64;
  const size_t lds = (size_t)2 * (single ? 1 : 2) * nch * pix * 16 + (size_t)2 * 3 * pix * 36;   // sample tiles + sampling plan
  const long long hw = (long long)p.Ho * p.Wo;
  const unsigned blocks = p.sum_part ? (unsigned)(p.B * ((hw + pix - 1) / pix)) : (unsigned)((p.total_pix + pix - 1) / pix);   // sum_part: tiles per image
  if (!single && nch % 2 == 0 && (p.Co == 128 || p.Co == 256)) {
#define DCN_WN4(NT_, NCH_)                                                                                                       \
    if (p.Co == 128 * NT_ && nch == NCH_) {                                                                                       \
      hipLaunchKernelGGL((dcn_fwd_fast_kernel<NT_, NCH_, 2, false, 4>), dim3(blocks), dim3(DC_THREADS), lds, stream, p);          \
      return glare_launch_status();                                                                                               \
    }
    DCN_WN4(1, 4) DCN_WN4(2, 4) DCN_WN4(1, 8) DCN_WN4(2, 8)
#undef DCN_WN4
  }
#define DCN_FAST(NT_, NCH_)                                                                                  \
  if (nt == NT_ && nch == NCH_) {                                                                            \
    if (single)                                                                                              \
      hipLaunchKernelGGL((dcn_fwd_fast_kernel<NT_, NCH_, 1, true>), dim3(blocks), dim3(DC_THREADS), lds, stream, p); \
    else                                                                                                     \
      hipLaunchKernelGGL((dcn_fwd_fast_kernel<NT_, NCH_, 1>), dim3(blocks), dim3(DC_THREADS), lds, stream, p); \
    return glare_launch_status();                                                                            \
  }
  DCN_FAST(1, 4) DCN_FAST(2, 4) DCN_FAST(4, 4) DCN_FAST(1, 8) DCN_FAST(2, 8) DCN_FAST(4, 8)
#undef DCN_FAST
  return GLARE_ERR_UNSUPPORTED;
}

int dcn_check(int B, int C, int H, int W, int Co, int kh, int kw, int sh, int sw, int dh, int dw, int groups, int dg) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || Co <= 0 || kh <= 0 || kw <= 0 || sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0 ||
      dg <= 0 || groups <= 0)
    return GLARE_ERR_INVALID;
  if (C % dg) return GLARE_ERR_INVALID;                     // shape_check, deform_conv_cuda.cpp:511-516
  if (groups != 1) return GLARE_ERR_UNSUPPORTED;            // the path uses groups = 1 (deformableDecoder_arch.py:283)
  const int cpg = C / dg;
  if (cpg != 32 && cpg != 64) return GLARE_ERR_UNSUPPORTED; // 64 px x cpg/8 items over 256 lanes
  if (Co % 64 || Co > 256) return GLARE_ERR_UNSUPPORTED;
  return GLARE_OK;
}

}  // namespace

extern "C" int glare_mdcn_pack_weight_f32(const float* weight_oihw, float* packed, int Co, int C, int kh, int kw, int dg,
                                          glare_stream_t stream) {
  if (!weight_oihw || !packed || Co <= 0 || C <= 0 || kh <= 0 || kw <= 0 || dg <= 0 || C % dg) return GLARE_ERR_INVALID;
  const long long total = (long long)Co * C * kh * kw;
  hipLaunchKernelGGL(dcn_pack_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     weight_oihw, reinterpret_cast<bf16_t*>(packed), Co, C, kh * kw, dg);
  return glare_launch_status();
}

extern "C" int glare_mdcn_pack_weight_single_f32(const float* weight_oihw, void* packed, int Co, int C, int kh, int kw, int dg,
                                                 glare_stream_t stream) {
  if (!weight_oihw || !packed || Co <= 0 || C <= 0 || kh <= 0 || kw <= 0 || dg <= 0 || C % dg || (C / dg) % 8) return GLARE_ERR_INVALID;
  const long long total = (long long)Co * C * kh * kw;
  hipLaunchKernelGGL(dcn_pack_weight_single_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     weight_oihw, reinterpret_cast<a16_t*>(packed), Co, C, kh * kw, dg);
  return glare_launch_status();
}

static int mdcn_forward_nhwc_impl(const void* x, int x_is_bf16, int x_pitch, int x_off, const float* offset,
                                  long long offset_plane, long long offset_batch_stride, const float* mask,
                                  long long mask_plane, long long mask_batch_stride, int mask_is_logit,
                                  const float* weight_packed, const float* bias, float* out, int out_planar,
                                  int out_pitch, int out_off, long long out_plane, int B, int C, int H, int W, int Co,
                                  int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int groups, int dg,
                                  int flags, const MdcnFusedOut* fo, glare_stream_t stream) {
  if (!x || !offset || !mask || !weight_packed || (!out && !fo)) return GLARE_ERR_INVALID;
  const int st = dcn_check(B, C, H, W, Co, kh, kw, sh, sw, dh, dw, groups, dg);
  if (st != GLARE_OK) return st;
  if ((x_pitch % 8) || (x_off % 8) || x_off + C > x_pitch) return GLARE_ERR_UNSUPPORTED;
  DcnParams p;
  p.x = x; p.offset = offset; p.mask = mask; p.wt = weight_packed; p.bias = bias; p.out = out;
  p.B = B; p.C = C; p.H = H; p.W = W; p.Co = Co;
  p.Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1;
  p.Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1;
  if (p.Ho <= 0 || p.Wo <= 0) return GLARE_ERR_INVALID;
  p.kh = kh; p.kw = kw; p.sh = sh; p.sw = sw; p.ph = ph; p.pw = pw; p.dh = dh; p.dw = dw; p.dg = dg; p.cpg = C / dg;
  p.xpitch = x_pitch; p.xoff = x_off;
  p.off_plane = offset_plane > 0 ? offset_plane : (long long)p.Ho * p.Wo;
  p.mask_plane = mask_plane > 0 ? mask_plane : (long long)p.Ho * p.Wo;
  p.off_bstride = offset_batch_stride > 0 ? offset_batch_stride : (long long)dg * 2 * kh * kw * p.off_plane;
  p.mask_bstride = mask_batch_stride > 0 ? mask_batch_stride : (long long)dg * kh * kw * p.mask_plane;
  p.mask_is_logit = mask_is_logit;
  p.out_planar = out_planar; p.opitch = out_pitch; p.ooff = out_off;
  p.out_plane = out_plane > 0 ? out_plane : (long long)p.Ho * p.Wo;
  if (!out_planar && out_off + Co > out_pitch) return GLARE_ERR_INVALID;
  p.total_pix = (long long)B * p.Ho * p.Wo;
  p.out16 = nullptr; p.sum_part = nullptr;
  if (fo) {                                                        // glare_mdcn_forward_nhwc_fused
    if (out_planar) return GLARE_ERR_UNSUPPORTED;
    p.out = fo->out32; p.out16 = (a16_t*)fo->out16; p.sum_part = fo->sum_part;
    if (p.out16 ? ((out_pitch % 2) || (out_off % 2)) : !p.out) return GLARE_ERR_INVALID;
  }
  // fast path: bf16 x and every extent addressable by a 31-bit buffer offset (the out-of-range sentinel is 2^31)
  const long long LIM = 0x7fffffffLL;
  const long long x_bytes = (long long)B * H * W * x_pitch * 2;
  const long long off_bytes = ((long long)(B - 1) * p.off_bstride + (long long)dg * 2 * kh * kw * p.off_plane) * 4;
  const long long mask_bytes = ((long long)(B - 1) * p.mask_bstride + (long long)dg * kh * kw * p.mask_plane) * 4;
  const long long wt_bytes = (long long)Co * C * kh * kw * 4;
  const bool generic_only = (flags & GLARE_MDCN_GENERAL_KERNEL) != 0, single = (flags & GLARE_MDCN_SINGLE_PASS) != 0;
  if (single && generic_only) return GLARE_ERR_INVALID;
  const int taps = kh * kw;
  if (x_is_bf16 && !generic_only && taps % 3 == 0 && x_bytes < LIM && off_bytes < LIM && mask_bytes < LIM && wt_bytes < LIM &&
      p.total_pix < LIM - 256) {
    p.x_bytes = (unsigned)x_bytes; p.off_bytes = (unsigned)off_bytes; p.mask_bytes = (unsigned)mask_bytes;
    p.wt_bytes = (unsigned)(single ? wt_bytes / 2 : wt_bytes);
    return launch_dcn_fast(p, single, (hipStream_t)stream);
  }
  if (single || p.out16 || p.sum_part) return GLARE_ERR_UNSUPPORTED;   // these forms exist on the fast path only (16-bit x, 3 | kh * kw, < 2 GB tensors)
  p.x_bytes = p.off_bytes = p.mask_bytes = p.wt_bytes = 0;
  return x_is_bf16 ? launch_dcn<true>(p, (hipStream_t)stream) : launch_dcn<false>(p, (hipStream_t)stream);
}

extern "C" int glare_mdcn_forward_nhwc(const void* x, int x_is_bf16, int x_pitch, int x_off, const float* offset,
                                       long long offset_plane, long long offset_batch_stride, const float* mask,
                                       long long mask_plane, long long mask_batch_stride, int mask_is_logit,
                                       const float* weight_packed, const float* bias, float* out, int out_planar,
                                       int out_pitch, int out_off, long long out_plane, int B, int C, int H, int W, int Co,
                                       int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int groups, int dg,
                                       int flags, glare_stream_t stream) {
  if (!out || (flags & ~(GLARE_MDCN_GENERAL_KERNEL | GLARE_MDCN_SINGLE_PASS))) return GLARE_ERR_INVALID;
  return mdcn_forward_nhwc_impl(x, x_is_bf16, x_pitch, x_off, offset, offset_plane, offset_batch_stride, mask, mask_plane, mask_batch_stride,
                                mask_is_logit, weight_packed, bias, out, out_planar, out_pitch, out_off, out_plane, B, C, H, W, Co, kh, kw, sh,
                                sw, ph, pw, dh, dw, groups, dg, flags, nullptr, stream);
}

extern "C" int glare_mdcn_tile_pixels(int C, int Co, int dg, int flags) {
  if (C <= 0 || Co <= 0 || dg <= 0 || C % dg) return GLARE_ERR_INVALID;
  return ((flags & GLARE_MDCN_SINGLE_PASS) && Co / 64 == 2 && (C / dg) / 8 == 4) ? 128 : 64;     // launch_dcn_fast's tile sizes
}

// The pipeline's form of the call above (round 6): the output as 16-bit NHWC (out16, else fp32 out32) and / or the
// per-tile sums of the fp32 outputs (tile_sums [B][ceil(Ho Wo / glare_mdcn_tile_pixels())]: the tiles are then cut per image).  Fast kernel
// only (16-bit x, 3 | kh kw, tensors < 2 GB): GLARE_ERR_UNSUPPORTED otherwise -- the caller falls back to the plain call.
extern "C" int glare_mdcn_forward_nhwc_fused(const void* x, int x_pitch, int x_off, const float* offset, long long offset_plane,
                                             long long offset_batch_stride, const float* mask, long long mask_plane,
                                             long long mask_batch_stride, int mask_is_logit, const float* weight_packed, const float* bias,
                                             float* out32_or_null, void* out16_or_null, int out_pitch, int out_off, float* tile_sums_or_null,
                                             int B, int C, int H, int W, int Co, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                                             int groups, int dg, int flags, glare_stream_t stream) {
  if ((flags & ~GLARE_MDCN_SINGLE_PASS) || (!out16_or_null && !tile_sums_or_null) || (!out16_or_null && !out32_or_null)) return GLARE_ERR_INVALID;
  const MdcnFusedOut fo = {out16_or_null ? nullptr : out32_or_null, out16_or_null, tile_sums_or_null};
  return mdcn_forward_nhwc_impl(x, 1, x_pitch, x_off, offset, offset_plane, offset_batch_stride, mask, mask_plane, mask_batch_stride,
                                mask_is_logit, weight_packed, bias, nullptr, 0, out_pitch, out_off, 0, B, C, H, W, Co, kh, kw, sh, sw, ph, pw,
                                dh, dw, groups, dg, flags, &fo, stream);
}

extern "C" size_t glare_mdcn_workspace_bytes(int B, int C, int H, int W, int Co, int kh, int kw) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || Co <= 0 || kh <= 0 || kw <= 0) return 0;
  return ((size_t)B * C * H * W + (size_t)Co * C * kh * kw) * sizeof(float) + 256;
}

// Drop-in for deform_conv_ext.modulated_deform_conv_forward (deform_conv_ext.cpp:107-124): reference
// layouts (NCHW fp32 everywhere), caller-allocated output, callee-owned scratch replaced by `workspace`.
extern "C" int glare_mdcn_forward_f32(const float* x, const float* offset, const float* mask, const float* weight,
                                      const float* bias_or_null, float* out, int B, int C, int H, int W, int Co, int kh,
                                      int kw, int sh, int sw, int ph, int pw, int dh, int dw, int groups, int dg,
                                      void* workspace, size_t workspace_bytes, glare_stream_t stream) {
  if (!x || !offset || !mask || !weight || !out) return GLARE_ERR_INVALID;
  const int st = dcn_check(B, C, H, W, Co, kh, kw, sh, sw, dh, dw, groups, dg);
  if (st == GLARE_ERR_UNSUPPORTED) {   // outside the MFMA kernels' configurations: the general fp32 kernel (dcn_generic.hip), no workspace
    const int gs = glare_mdcn_generic_check(B, C, H, W, Co, kh, kw, sh, sw, dh, dw, groups, dg);
    if (gs != GLARE_OK) return gs;
    return glare_mdcn_generic_forward(x, offset, mask, weight, bias_or_null, out, B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, groups,
                                      dg, (hipStream_t)stream);
  }
  if (st != GLARE_OK) return st;
  if (!workspace || workspace_bytes < glare_mdcn_workspace_bytes(B, C, H, W, Co, kh, kw)) return GLARE_ERR_WORKSPACE;
  float* x_nhwc = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
  float* wt = x_nhwc + (size_t)B * C * H * W;
  int rc = glare_nchw_to_nhwc(x, x_nhwc, B, C, (long long)H * W, C, 0, 0, stream);
  if (rc != GLARE_OK) return rc;
  rc = glare_mdcn_pack_weight_f32(weight, wt, Co, C, kh, kw, dg, stream);
  if (rc != GLARE_OK) return rc;
  return glare_mdcn_forward_nhwc(x_nhwc, 0, C, 0, offset, 0, 0, mask, 0, 0, 0, wt, bias_or_null, out, 1, 0, 0, 0, B, C, H, W, Co,
                                 kh, kw, sh, sw, ph, pw, dh, dw, groups, dg, 0, stream);
}
