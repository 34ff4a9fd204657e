// 1x1 convolution as a WEIGHT-STATIONARY GEMM (round 2).
//
// Replaces the 1x1 nn.Conv2d's of the path with 128 <= Cin <= 512 and Cout a multiple of 128: AttnBlock's folded query / output
// projections (encoder_decoder.py:146-165, 22 launches per 8-image step) and the ResnetBlock nin_shortcuts (:104-115).
//   out[p, co] = act(sum_ci x[p, ci] * w[co, ci] + bias[co]) (+ residual[p, co])
// The implicit-GEMM conv kernel (conv_igemm.hip, KS = 1) ran these at 0.17-0.18 of the bf16 peak: with K = Cin = 512 a 256 x 128
// tile is only 16 stages of 16 MFMAs per wave, each with its barrier and 6 LDS-DMA pieces per wave, plus a prologue and an
// LDS-staged epilogue per tile.  The GEMM is short in K and long in M (130 200 pixels x 512 x 512 at the quarter resolution): it is
// bound by HBM (x read once, out written once, residual read once: 400 MB = 0.07 ms at 5.7 TB/s), not by the matrix pipe (0.03 ms).
// So the weights stay put and the pixels stream:
//   * one persistent workgroup per CU (256 in all); its 128-cout slice of W (128 x Cin bf16 <= 128 KB) is DMA'd into LDS ONCE;
//   * each of the 4 waves (one per SIMD) walks its own 32-pixel row blocks: A fragments straight from global memory to
//     registers (32 rows x 32 B per instruction; the four k-steps that share a 128-B line follow each other), B fragments
//     from the resident LDS tile (XOR-swizzled, conflict-free ds_read_b128), 4 independent accumulators (32 px x 128 co);
//   * no barrier and no DMA in the steady state: the waves never wait for each other;
//   * the 128 / Cout... co-tiles of one pixel range run on CUs of the SAME XCD (block -> (xcd, slot)), so x is fetched from
//     HBM once per range and the other co-tiles hit L2;
//   * epilogue per row block through a wave-private 8 KB LDS slab: bias, residual (prefetched at the start of the block),
//     activation, 16-B stores, and the GroupNorm partial sums of the rounded output (one part per (image, row block)).
// Measured (MI355X, 8 x 105 x 155 pixels, 512 -> 512): 0.110-0.117 ms against 0.143-0.150 for the implicit-GEMM kernel (0.147 vs
// 0.189-0.206 with the residual); MFMA + LDS core alone 0.083 ms -- one 1-KB B fragment from LDS per MFMA is half the LDS
// bandwidth; a two-row-block variant that halves it needs 128 more registers and spilled (tried, dropped).
// Round 5 built that variant properly (64 pixels per wave pass, every B fragment feeding two MFMAs, A fragments in a ring of 8 k-steps
// refilled behind their MFMAs, units of a pass as an unrolled inner loop so that the accumulators stay in AGPRs, one epilogue copy: 0 B of
// scratch, bit-identical outputs, 14 GPU tests): 512 -> 512 @q 0.109 ms against 0.109 (+ residual 0.143 against 0.138), 512 -> 1024 0.212
// against 0.227 -- and 61.1 images/s either way.  The LDS fragment traffic is NOT what bounds this kernel: it is the L1 / texture path --
// an A-fragment load instruction reads 16 B from each of 64 different 128-B lines (a lane's row is fixed by the MFMA layout), and the four
// co-tiles of a pixel range each pull all of x through their CU's L1: 533 MB of 16-B accesses per launch.  Removed again.
#include <type_traits>

#include "common.h"

namespace {

struct C1Params {
  const a16_t* x;
  const a16_t* w;       // [Cout][Cin] bf16
  const float* bias;
  const a16_t* res;
  const a16_t* res_lo;   // hi / lo form (conv_igemm.hip's HILO epilogue): remainder halves of the residual and of the output
  a16_t* out;
  a16_t* out_lo;
  const a16_t* x_lo;     // SPLIT (fp32-class form): remainder halves of the activation and of the filter
  const a16_t* w_lo;
  float* gn_part;        // [B][rbi][Cout/4][2] or null
  int B, N;              // images, pixels per image
  int Cin, Cout;
  int xpitch, xoff, opitch, ooff, rpitch, roff;
  int act, nct, rbi;     // co-tiles (Cout / 128), row blocks per image
  long long w_istride;   // 0: one filter for all images; else elements between the filters of consecutive images (a pixel range
  int b_istride;         // then never straddles two images: (8 * 32 / nct) % B == 0); likewise for the bias
};

template <int ACT>   // compile-time inside the epilogue (a runtime switch per element costs ~3 scalar branches per element)
__device__ __forceinline__ float act1(float v) {
  if constexpr (ACT == GLARE_ACT_RELU) return fmaxf(v, 0.f);
  else if constexpr (ACT == GLARE_ACT_SIGMOID) return sigmoidf_(v);
  else if constexpr (ACT == GLARE_ACT_SWISH) return swishf_(v);
  else return v;
}
template <typename F>
__device__ __forceinline__ void with_act1(int act, F&& f) {
  switch (act) {
    case GLARE_ACT_RELU: f(std::integral_constant<int, GLARE_ACT_RELU>{}); break;
    case GLARE_ACT_SIGMOID: f(std::integral_constant<int, GLARE_ACT_SIGMOID>{}); break;
    case GLARE_ACT_SWISH: f(std::integral_constant<int, GLARE_ACT_SWISH>{}); break;
    default: f(std::integral_constant<int, GLARE_ACT_NONE>{}); break;
  }
}

// SPLIT (round 4): the fp32-class form of glare_conv_desc.k_wrap on this kernel -- x and w are hi / lo pairs and
//   out = x_hi . w_hi + x_lo . w_hi + x_hi . w_lo     (fp32 accumulation; the dropped x_lo . w_lo is 2^-22)
// on a 64-cout tile: the resident LDS image holds rows 0-63 = w_hi and rows 64-127 = w_lo of the SAME 64 output channels, so the
// four accumulators of a row block are (x_hi . w_hi | x_hi . w_lo) and the x_lo fragments feed the first two only: 6 MFMAs per k-step
// for 64 channels = 3 per product, both halves of x read once per co-tile.  The epilogue is the hi / lo one (fp32 through the slab,
// nothing rounded before the residual add); out_lo == nullptr writes the 16-bit value alone.
template <int KSTEPS, bool HILO = false, bool SPLIT = false>   // Cin / 16
__global__ __launch_bounds__(256, 1) void conv1x1_ws_kernel(const C1Params p) {
  static_assert(!SPLIT || HILO, "the split form uses the hi / lo epilogue");
  constexpr int CT = SPLIT ? 64 : 128;           // output channels per tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ROWB = KSTEPS * 32;              // bytes of one weight row
  constexpr int CPR = ROWB / 16;                 // 16-B chunks per weight row (16 / 32 / 64)
  constexpr int RPP = 1024 / ROWB;               // weight rows per 1-KB DMA piece (1, 2 or 4... 8 for Cin = 128)
  char* const slab_base = smem + 128 * ROWB;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // block -> (xcd, slot): the co-tiles of one pixel range sit on one XCD
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;                 // slot 0..31
  const int ct = slot % p.nct, rng_in_xcd = slot / p.nct, rpx = 32 / p.nct;
  const int range = xcd * rpx + rng_in_xcd, n_ranges = 8 * rpx;
  const int total_rb = p.B * p.rbi;
  int rb_lo = (int)((long long)total_rb * range / n_ranges), rb_hi = (int)((long long)total_rb * (range + 1) / n_ranges);
  int w_img = 0;
  if (p.w_istride != 0) {   // per-image filters: ranges are cut per image
    const int rpi = n_ranges / p.B, sub = range % rpi;
    w_img = range / rpi;
    rb_lo = w_img * p.rbi + (int)((long long)p.rbi * sub / rpi);
    rb_hi = w_img * p.rbi + (int)((long long)p.rbi * (sub + 1) / rpi);
  }

  // ---- weights of this co-tile into LDS, once: row r, chunk c at r * CPR + (c ^ (r & 15)) (source-side swizzle)
  {
    const a16_t* wt = p.w + (size_t)w_img * p.w_istride + (size_t)ct * CT * p.Cin;
    [[maybe_unused]] const a16_t* wl = SPLIT ? p.w_lo + (size_t)ct * CT * p.Cin : nullptr;
    const int r_in = lane / CPR, c_ph = lane % CPR;
#pragma unroll 4
    for (int piece = wave; piece < 128 / RPP; piece += 4) {
      const int r = piece * RPP + r_in;
      const int c_src = c_ph ^ (r & 15);
      const a16_t* wrow = (SPLIT && r >= 64) ? wl + (size_t)(r - 64) * p.Cin : wt + (size_t)r * p.Cin;   // a piece never straddles the halves
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wrow + c_src * 8),
                                       (__attribute__((address_space(3))) void*)(smem + piece * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  const int px = lane & 31, khalf = lane >> 5;
  // The contraction order is free as long as A and B agree: "k-step" v = 4 g + i pairs lane (px, khalf) with the 8 channels
  // 64 g + 32 khalf + 8 i .. +8, so that a lane's four A loads of a group are 64 CONTIGUOUS bytes and the two half-waves cover
  // whole 128-B lines (with the natural order 16 v + 8 khalf every load instruction touched a quarter of 32 lines, four times).
  // B fragment (j, v): row 32 j + px, 16-B chunk 8 g + 4 khalf + i  ->  boff[v & 7] + (v >> 3) * 256 + j * 32 * ROWB
  int boff[8];
#pragma unroll
  for (int bb = 0; bb < 8; ++bb) boff[bb] = px * ROWB + ((((bb >> 2) * 8 + 4 * khalf + (bb & 3)) ^ (px & 15)) * 16);
  char* const slab = slab_base + wave * 8192;

  const int co0 = ct * CT;
  float bias_v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bias_v[j] = (p.bias && (!SPLIT || j < 2)) ? p.bias[(size_t)w_img * p.b_istride + co0 + 32 * j + px] : 0.f;

  // A fragments are fetched one UNIT (16 k-steps = 16 KB per wave) ahead of the MFMAs that consume them: with one wave per SIMD
  // nothing else hides the memory latency (8 k-steps ahead left the kernel latency-bound at 1.3 TB/s), and two whole row blocks of
  // fragments (256 registers) spill -- a scratch reload in this loop waits vmcnt(0), i.e. for every prefetch in flight.
  constexpr int UNIT = SPLIT ? 8 : (KSTEPS < 16 ? KSTEPS : 16), UPB = KSTEPS / UNIT;          // units per row block (SPLIT: two fragment sets)
  auto unit_ptr = [&](int q) {                                                   // unit q of this wave: row block + k offset
    const int rb_ = rb_lo + wave + 4 * (q / UPB), u_ = q % UPB;
    const int b_ = rb_ / p.rbi, r0_ = (rb_ - b_ * p.rbi) * 32;
    const int nrows_ = min(32, p.N - r0_);
    return ((size_t)b_ * p.N + r0_ + min(px, nrows_ - 1)) * p.xpitch + p.xoff + khalf * 32 + u_ * UNIT * 16;     // element offset
  };
  const int n_rb = rb_hi > rb_lo + wave ? (rb_hi - rb_lo - wave + 3) / 4 : 0, n_units = n_rb * UPB;
  a16x8 af[2][UNIT];
  [[maybe_unused]] a16x8 al[2][SPLIT ? UNIT : 1];
  auto fetch = [&](auto parc, int q) {
    constexpr int P = decltype(parc)::value;
    const size_t aoff = unit_ptr(q);
    const a16_t* ap = p.x + aoff;
    if constexpr (SPLIT) {
      const a16_t* lp = p.x_lo + aoff;
#pragma unroll
      for (int e = 0; e < UNIT; ++e) al[P][e] = *reinterpret_cast<const a16x8*>(lp + (e >> 2) * 64 + (e & 3) * 8);
    }
#pragma unroll
    for (int e = 0; e < UNIT; ++e) {
      af[P][e] = *reinterpret_cast<const a16x8*>(ap + (e >> 2) * 64 + (e & 3) * 8);
    }
  };
  if (n_units > 0) fetch(std::integral_constant<int, 0>{}, 0);
  f32x16 acc[4];
  u32x4 resv[8];
  [[maybe_unused]] u32x4 resl[8];
  for (int q = 0; q < n_units; ++q) {
    const int rb = rb_lo + wave + 4 * (q / UPB), u = q % UPB;
    const int b = rb / p.rbi, r0 = (rb - b * p.rbi) * 32;
    const int nrows = min(32, p.N - r0);
    const size_t pix0 = (size_t)b * p.N + r0;
    if (u == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
      if (p.res) {   // the residual rows of this block, fetched NOW: they land under the MFMAs instead of stalling the epilogue
#pragma unroll
        for (int it = 0; it < (SPLIT ? 4 : 8); ++it) {   // SPLIT: one 64-channel half
          // hi / lo form: the epilogue runs in two 64-channel halves, item it = 4 * half + t is row (lane >> 3) + 8 t, chunk 8 half + (lane & 7)
          const int m = HILO ? min((lane >> 3) + 8 * (it & 3), nrows - 1) : min((lane >> 4) + 4 * it, nrows - 1);
          const int chn = HILO ? 8 * (it >> 2) + (lane & 7) : (lane & 15);
          resv[it] = *reinterpret_cast<const u32x4*>(p.res + (pix0 + m) * p.rpitch + p.roff + co0 + chn * 8);
          if constexpr (HILO)
            resl[it] = p.res_lo ? *reinterpret_cast<const u32x4*>(p.res_lo + (pix0 + m) * p.rpitch + p.roff + co0 + chn * 8) : u32x4{0u, 0u, 0u, 0u};
        }
      }
    }
    auto body = [&](auto parc) {
      constexpr int P = decltype(parc)::value;
      if (q + 1 < n_units) fetch(std::integral_constant<int, P ^ 1>{}, q + 1);
#pragma unroll
      for (int e = 0; e < UNIT; ++e) {
        const int ks = u * UNIT + e;            // u is 0 when UPB == 1; otherwise boff[] only depends on ks & 7 = e & 7
        a16x8 bf[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          bf[j] = *reinterpret_cast<const a16x8*>(smem + boff[e & 7] + (ks >> 3) * 256 + j * 32 * ROWB);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = mfma_a16_32x32x16(af[P][e], bf[j], acc[j], 0, 0, 0);
        if constexpr (SPLIT) {   // x_lo . w_hi: the first two accumulators only (rows 0-63 of the image are w_hi)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[j] = mfma_a16_32x32x16(al[P][e], bf[j], acc[j], 0, 0, 0);
        }
      }
    };
    if ((q & 1) == 0) body(std::integral_constant<int, 0>{});
    else body(std::integral_constant<int, 1>{});
    if (u != UPB - 1) continue;

    // ---- epilogue: C/D layout col = lane & 31 (co within the 32-tile), row = (r & 3) + 8 (r >> 2) + 4 khalf (pixel).
    // Phase 1: + bias (+ act when there is no residual) -> bf16; neighbouring lanes (co, co + 1) exchange one value so that each
    // lane writes one packed 4-B word per register pair; slab row m = 256 B, 16-B chunk c at (c ^ (m & 15)).
    float gs0 = 0.f, gq0 = 0.f, gs1 = 0.f, gq1 = 0.f;
    if constexpr (HILO) {
      // hi / lo epilogue (see conv_igemm.hip): fp32 through the 8 KB slab, 64 output channels at a time (row = 64 floats, 16-B
      // chunk c at c ^ (m & 15)); v = acc + bias + residual_hi + residual_lo, hi = round16(v), lo = round16(v - hi)
      with_act1(p.act, [&](auto actc) {
      constexpr int ACT = decltype(actc)::value;
#pragma unroll
      for (int hf = 0; hf < (SPLIT ? 1 : 2); ++hf) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int j = 2 * hf + jj;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = (r & 3) + 8 * (r >> 2) + 4 * khalf;
            const int col = 32 * jj + px;                                      // 0..63 within the half
            float v0 = acc[j][r] + bias_v[j];
            if constexpr (SPLIT) v0 = (acc[jj][r] + acc[2 + jj][r]) + bias_v[jj];   // (x_hi.w_hi + x_lo.w_hi) + x_hi.w_lo
            *reinterpret_cast<float*>(slab + m * 256 + (((col >> 2) ^ (m & 15)) * 16) + (col & 3) * 4) = v0;
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int m = (lane >> 3) + 8 * t, c8 = lane & 7;
          if (m < nrows) {
            const f32x4 f0 = *reinterpret_cast<const f32x4*>(slab + m * 256 + (((2 * c8) ^ (m & 15)) * 16));
            const f32x4 f1 = *reinterpret_cast<const f32x4*>(slab + m * 256 + (((2 * c8 + 1) ^ (m & 15)) * 16));
            float v[8] = {f0[0], f0[1], f0[2], f0[3], f1[0], f1[1], f1[2], f1[3]};
            if (p.res) {
              const u32x4 rv = resv[4 * hf + t], rl = resl[4 * hf + t];   // (a 16-bit residual: res_lo == nullptr arrives as zeros)
#pragma unroll
              for (int e = 0; e < 4; ++e) { v[2 * e] += alo(rv[e]) + alo(rl[e]); v[2 * e + 1] += ahi(rv[e]) + ahi(rl[e]); }
            }
            u32x4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float a = act1<ACT>(v[2 * e]), c = act1<ACT>(v[2 * e + 1]);
              hi[e] = pack_a2(a, c);
              lo[e] = pack_a2(a - alo(hi[e]), c - ahi(hi[e]));
              if (e < 2) { gs0 += a + c; gq0 += a * a + c * c; } else { gs1 += a + c; gq1 += a * a + c * c; }
            }
            const size_t o = (pix0 + m) * p.opitch + p.ooff + co0 + (8 * hf + c8) * 8;
            *reinterpret_cast<u32x4*>(p.out + o) = hi;
            if (!SPLIT || p.out_lo) *reinterpret_cast<u32x4*>(p.out_lo + o) = lo;
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (p.gn_part) {   // lanes 8 apart hold the same channel chunk of this half
#pragma unroll
          for (int o = 8; o < 64; o <<= 1) {
            gs0 += __shfl_xor(gs0, o, 64); gq0 += __shfl_xor(gq0, o, 64);
            gs1 += __shfl_xor(gs1, o, 64); gq1 += __shfl_xor(gq1, o, 64);
          }
          if (lane < 8) {
            float* dst = p.gn_part + (((size_t)b * p.rbi + (rb - b * p.rbi)) * (p.Cout / 4) + (co0 + (8 * hf + lane) * 8) / 4) * 2;
            dst[0] = gs0; dst[1] = gq0; dst[2] = gs1; dst[3] = gq1;
          }
          gs0 = gq0 = gs1 = gq1 = 0.f;
        }
      }
      });
      continue;
    }
    with_act1(p.act, [&](auto actc) {
    constexpr int ACT = decltype(actc)::value;
    const bool act_early = p.res == nullptr;
    const int odd = lane & 1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        float a = acc[j][2 * t] + bias_v[j], c = acc[j][2 * t + 1] + bias_v[j];
        if (act_early) { a = act1<ACT>(a); c = act1<ACT>(c); }
        const float send = odd ? a : c;
        const float recv = __shfl_xor(send, 1, 64);
        const int r = 2 * t + odd;
        const int m = (r & 3) + 8 * (r >> 2) + 4 * khalf;
        const uint32_t wv = odd ? pack_a2(recv, c) : pack_a2(a, recv);
        const int col = 32 * j + (px & ~1);                               // even channel of the pair, 0..127
        *reinterpret_cast<uint32_t*>(slab + m * 256 + (((col >> 3) ^ (m & 15)) * 16) + (col & 7) * 2) = wv;
      }
    }
    // Phase 2: 16-B rows out: lane handles chunk ch = lane & 15 of rows (lane >> 4) + 4 it
    const int ch = lane & 15;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int m = (lane >> 4) + 4 * it;
      if (m < nrows) {
        u32x4 v = *reinterpret_cast<const u32x4*>(slab + m * 256 + ((ch ^ (m & 15)) * 16));
        const size_t pix = pix0 + m;
        const int co = co0 + ch * 8;
        if (p.res) {
          const u32x4 rv = resv[it];
#pragma unroll
          for (int e = 0; e < 4; ++e)
            v[e] = pack_a2(act1<ACT>(alo(v[e]) + alo(rv[e])), act1<ACT>(ahi(v[e]) + ahi(rv[e])));
        }
        *reinterpret_cast<u32x4*>(p.out + pix * p.opitch + p.ooff + co) = v;
        if (p.gn_part) {
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const float x0 = alo(v[e]), x1 = ahi(v[e]), y0 = alo(v[2 + e]), y1 = ahi(v[2 + e]);
            gs0 += x0 + x1; gq0 += x0 * x0 + x1 * x1;
            gs1 += y0 + y1; gq1 += y0 * y0 + y1 * y1;
          }
        }
      }
    }
    });
    if (p.gn_part) {   // lanes 16 apart hold the same channel chunk
#pragma unroll
      for (int o = 16; o < 64; o <<= 1) {
        gs0 += __shfl_xor(gs0, o, 64); gq0 += __shfl_xor(gq0, o, 64);
        gs1 += __shfl_xor(gs1, o, 64); gq1 += __shfl_xor(gq1, o, 64);
      }
      if (lane < 16) {
        float* dst = p.gn_part + (((size_t)b * p.rbi + (rb - b * p.rbi)) * (p.Cout / 4) + (co0 + lane * 8) / 4) * 2;
        dst[0] = gs0; dst[1] = gq0; dst[2] = gs1; dst[3] = gq1;
      }
    }
  }
}

// OIHW fp32 [Cout][Cin][1][1] -> bf16 [Cout][Cin]
__global__ void c1_pack_kernel(const float* __restrict__ w, a16_t* __restrict__ out, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = f2a(w[i]);
}

// [B][rbi][Cout/4][2] -> [B][1][32][2] (the statistics block gn_apply consumes)
__global__ __launch_bounds__(256) void c1_gn_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, int nparts, int Cout) {
  __shared__ float red[2][4];
  const int b = blockIdx.x / 32, g = blockIdx.x % 32;
  const int upg = Cout / 128, U = Cout / 4;
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

}  // namespace

extern "C" int glare_conv1x1_ws_supported(int Cin, int Cout) {
  if (Cin != 128 && Cin != 256 && Cin != 512) return 0;
  if (Cout <= 0 || Cout % 128) return 0;
  const int nct = Cout / 128;
  return (nct <= 32 && 32 % nct == 0) ? 1 : 0;
}

extern "C" int glare_conv1x1_ws_pack_weight(const float* w_oihw, int cout, int cin, void* w_bf16, glare_stream_t stream) {
  if (!w_oihw || !w_bf16 || cout <= 0 || cin <= 0) return GLARE_ERR_INVALID;
  const long long n = (long long)cout * cin;
  hipLaunchKernelGGL(c1_pack_kernel, dim3((unsigned)cdivll(n, 256)), dim3(256), 0, (hipStream_t)stream, w_oihw, (a16_t*)w_bf16, n);
  return glare_launch_status();
}

extern "C" long long glare_conv1x1_ws_gn_partial_elems(int B, long long pixels_per_image, int Cout) {
  if (B <= 0 || pixels_per_image <= 0 || Cout <= 0) return GLARE_ERR_INVALID;
  return (long long)B * cdivll(pixels_per_image, 32) * (Cout / 4) * 2;
}

extern "C" int glare_conv1x1_ws_gn_reduce(const float* gn_partial, float* stats_out, int B, long long pixels_per_image, int Cout,
                                          glare_stream_t stream) {
  if (!gn_partial || !stats_out || B <= 0 || Cout % 128) return GLARE_ERR_INVALID;
  hipLaunchKernelGGL(c1_gn_reduce_kernel, dim3(B * 32), dim3(256), 0, (hipStream_t)stream, gn_partial, stats_out,
                     (int)cdivll(pixels_per_image, 32), Cout);
  return glare_launch_status();
}

static int c1_launch(const void* x, int x_pitch, int x_off, const void* w_bf16, long long w_istride, const float* bias, int b_istride,
                     const void* residual, int res_pitch, int res_off, void* out, int out_pitch, int out_off, int B,
                     long long pixels_per_image, int Cin, int Cout, int act, float* gn_partial, glare_stream_t stream,
                     const void* residual_lo = nullptr, void* out_lo = nullptr) {
  if (!x || !w_bf16 || !out || B <= 0 || pixels_per_image <= 0) return GLARE_ERR_INVALID;
  if (!glare_conv1x1_ws_supported(Cin, Cout)) return GLARE_ERR_UNSUPPORTED;
  if (w_istride != 0 && ((8 * 32 / (Cout / 128)) % B != 0 || w_istride % 8 != 0 || (bias && b_istride <= 0))) return GLARE_ERR_UNSUPPORTED;
  if ((x_pitch % 8) || (x_off % 8) || (out_pitch % 8) || (out_off % 8) || x_off + Cin > x_pitch || out_off + Cout > out_pitch)
    return GLARE_ERR_UNSUPPORTED;
  if (residual && ((res_pitch % 8) || (res_off % 8) || res_off + Cout > res_pitch)) return GLARE_ERR_UNSUPPORTED;
  if (pixels_per_image > 0x7fffffffLL || (long long)B * cdivll(pixels_per_image, 32) > 0x7fffffffLL) return GLARE_ERR_INVALID;
  C1Params p;
  p.x = (const a16_t*)x; p.w = (const a16_t*)w_bf16; p.bias = bias; p.res = (const a16_t*)residual; p.out = (a16_t*)out;
  p.gn_part = gn_partial;
  p.res_lo = (const a16_t*)residual_lo; p.out_lo = (a16_t*)out_lo;
  p.x_lo = nullptr; p.w_lo = nullptr;
  if (residual_lo && !(residual && out_lo)) return GLARE_ERR_INVALID;
  p.B = B; p.N = (int)pixels_per_image; p.Cin = Cin; p.Cout = Cout;
  p.xpitch = x_pitch; p.xoff = x_off; p.opitch = out_pitch; p.ooff = out_off; p.rpitch = res_pitch; p.roff = res_off;
  p.act = act; p.nct = Cout / 128; p.rbi = (int)cdivll(pixels_per_image, 32);
  p.w_istride = w_istride; p.b_istride = w_istride != 0 ? b_istride : 0;
  const size_t lds = (size_t)128 * Cin * 2 + 4 * 8192;
  hipStream_t s = (hipStream_t)stream;
#define C1_CASE(KS_, HL_)                                                                                                  \
  if (Cin == 16 * KS_ && (out_lo != nullptr) == HL_) {                                                                     \
    if (hipFuncSetAttribute((const void*)conv1x1_ws_kernel<KS_, HL_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
      return GLARE_ERR_LAUNCH;                                                                                             \
    hipLaunchKernelGGL((conv1x1_ws_kernel<KS_, HL_>), dim3(256), dim3(256), lds, s, p);                                    \
    return glare_launch_status();                                                                                          \
  }
  C1_CASE(8, false) C1_CASE(16, false) C1_CASE(32, false) C1_CASE(8, true) C1_CASE(16, true) C1_CASE(32, true)
#undef C1_CASE
  return GLARE_ERR_UNSUPPORTED;
}

extern "C" int glare_conv1x1_ws_bf16(const void* x, int x_pitch, int x_off, const void* w_bf16, const float* bias, const void* residual,
                                     int res_pitch, int res_off, void* out, int out_pitch, int out_off, int B, long long pixels_per_image,
                                     int Cin, int Cout, int act, float* gn_partial, glare_stream_t stream) {
  return c1_launch(x, x_pitch, x_off, w_bf16, 0, bias, 0, residual, res_pitch, res_off, out, out_pitch, out_off, B, pixels_per_image, Cin,
                   Cout, act, gn_partial, stream);
}

// The same with ONE FILTER PER IMAGE: w_bf16 = [B][Cout][Cin] bf16 (w_image_stride elements apart), bias = [B][bias_image_stride].
// GroupNorm without an activation in front of a 1x1 conv is a per-(image, channel) affine map of the conv's input, i.e. a per-image
// filter (glare_attn_fold_groupnorm_f32): the normalised tensor is never written.  Needs (8 * 32 / (Cout / 128)) % B == 0.
extern "C" int glare_conv1x1_ws_image_bf16(const void* x, int x_pitch, int x_off, const void* w_bf16, long long w_image_stride,
                                           const float* bias, int bias_image_stride, const void* residual, int res_pitch, int res_off,
                                           void* out, int out_pitch, int out_off, int B, long long pixels_per_image, int Cin, int Cout,
                                           int act, float* gn_partial, glare_stream_t stream) {
  if (w_image_stride <= 0) return GLARE_ERR_INVALID;
  return c1_launch(x, x_pitch, x_off, w_bf16, w_image_stride, bias, bias_image_stride, residual, res_pitch, res_off, out, out_pitch, out_off,
                   B, pixels_per_image, Cin, Cout, act, gn_partial, stream);
}

// The fp32-class form (SPLIT, see conv1x1_ws_kernel): activation and filter as hi / lo pairs of 16-bit tensors, 64-cout tiles.
extern "C" int glare_conv1x1_ws_split_supported(int Cin, int Cout) {
  if (Cin != 128 && Cin != 256 && Cin != 512) return 0;
  if (Cout <= 0 || Cout % 64) return 0;
  const int nct = Cout / 64;
  return (nct <= 32 && 32 % nct == 0) ? 1 : 0;
}

extern "C" int glare_conv1x1_ws_split_bf16(const void* x_hi, const void* x_lo, int x_pitch, int x_off, const void* w_hi, const void* w_lo,
                                           const float* bias, const void* residual, const void* residual_lo, int res_pitch, int res_off,
                                           void* out, void* out_lo, int out_pitch, int out_off, int B, long long pixels_per_image, int Cin,
                                           int Cout, int act, float* gn_partial, glare_stream_t stream) {
  if (!x_hi || !x_lo || !w_hi || !w_lo || !out || B <= 0 || pixels_per_image <= 0) return GLARE_ERR_INVALID;
  if (!glare_conv1x1_ws_split_supported(Cin, Cout)) return GLARE_ERR_UNSUPPORTED;
  if ((x_pitch % 8) || (x_off % 8) || (out_pitch % 8) || (out_off % 8) || x_off + Cin > x_pitch || out_off + Cout > out_pitch)
    return GLARE_ERR_UNSUPPORTED;
  if (residual && ((res_pitch % 8) || (res_off % 8) || res_off + Cout > res_pitch)) return GLARE_ERR_UNSUPPORTED;
  if (residual_lo && !residual) return GLARE_ERR_INVALID;
  if (gn_partial && Cout % 128) return GLARE_ERR_UNSUPPORTED;      // the statistics block is per 128-channel... group layout of gn_reduce
  if (pixels_per_image > 0x7fffffffLL || (long long)B * cdivll(pixels_per_image, 32) > 0x7fffffffLL) return GLARE_ERR_INVALID;
  C1Params p;
  p.x = (const a16_t*)x_hi; p.x_lo = (const a16_t*)x_lo; p.w = (const a16_t*)w_hi; p.w_lo = (const a16_t*)w_lo; p.bias = bias;
  p.res = (const a16_t*)residual; p.res_lo = (const a16_t*)residual_lo; p.out = (a16_t*)out; p.out_lo = (a16_t*)out_lo;
  p.gn_part = gn_partial;
  p.B = B; p.N = (int)pixels_per_image; p.Cin = Cin; p.Cout = Cout;
  p.xpitch = x_pitch; p.xoff = x_off; p.opitch = out_pitch; p.ooff = out_off; p.rpitch = res_pitch; p.roff = res_off;
  p.act = act; p.nct = Cout / 64; p.rbi = (int)cdivll(pixels_per_image, 32);
  p.w_istride = 0; p.b_istride = 0;
  const size_t lds = (size_t)128 * Cin * 2 + 4 * 8192;
  hipStream_t s = (hipStream_t)stream;
#define C1S_CASE(KS_)                                                                                                              \
  if (Cin == 16 * KS_) {                                                                                                           \
    if (hipFuncSetAttribute((const void*)conv1x1_ws_kernel<KS_, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
      return GLARE_ERR_LAUNCH;                                                                                                     \
    hipLaunchKernelGGL((conv1x1_ws_kernel<KS_, true, true>), dim3(256), dim3(256), lds, s, p);                                     \
    return glare_launch_status();                                                                                                  \
  }
  C1S_CASE(8) C1S_CASE(16) C1S_CASE(32)
#undef C1S_CASE
  return GLARE_ERR_UNSUPPORTED;
}

// hi / lo form of both entry points above (w_image_stride = 0: one filter for all images): the output leaves as hi = round16(v) in
// `out` and lo = round16(v - hi) in `out_lo` (same pitch / offset) of v = acc + bias + residual + residual_lo, nothing rounded before
// the residual add; the fused GroupNorm statistics are those of v.  The residual stream of the conditional encoder (nin_shortcut,
// AttnBlock's output projection) in the fp16 precision -- see glare_conv_desc.out_lo.
extern "C" int glare_conv1x1_ws_hilo_bf16(const void* x, int x_pitch, int x_off, const void* w_bf16, long long w_image_stride,
                                          const float* bias, int bias_image_stride, const void* residual, const void* residual_lo,
                                          int res_pitch, int res_off, void* out, void* out_lo, int out_pitch, int out_off, int B,
                                          long long pixels_per_image, int Cin, int Cout, int act, float* gn_partial,
                                          glare_stream_t stream) {
  if (!out_lo || w_image_stride < 0) return GLARE_ERR_INVALID;
  return c1_launch(x, x_pitch, x_off, w_bf16, w_image_stride, bias, bias_image_stride, residual, res_pitch, res_off, out, out_pitch, out_off,
                   B, pixels_per_image, Cin, Cout, act, gn_partial, stream, residual_lo, out_lo);
}
