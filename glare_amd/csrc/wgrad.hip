// Weight gradient of a 3x3 / stride-1 / pad-1 convolution straight from the NHWC activations (cuDNN wgrad in the reference:
// `loss.backward()` of LLFlow_model.py:231-236 / VQLLFLOWD_model.py:226-229 over every Conv2d of encoder_decoder.py /
// deformableDecoder_arch.py):
//
//     dW[co][ci][ty][tx] = sum_{b,y,x} g[b][y][x][co] * x[b][y+ty-1][x+tx-1][ci]        db[co] = sum g[b][y][x][co]
//
// The contraction runs over PIXELS, the slow axis of both NHWC operands, so both MFMA operands are K-major in memory.  gfx950
// reads such tiles directly: the [pixel][channel] rows land in LDS by LDS-DMA and `ds_read_b64_tr_b16` hands each lane its
// 4 pixels x 1 channel column -- no transposed copy of x or g in HBM (the planar transposes this kernel replaces wrote 4x the
// activation).  One workgroup owns a (64 ci) x (64 co) block of ALL NINE taps, so an x row is fetched once and used by the three
// tap rows it belongs to, a g row once for nine taps: the nine taps differ only by the LDS address of the A fragment.
//
//   grid      : split s = (image, column strip of 16*nks pixels, range of rows) x tile (ci block, co block); the splits of one
//               tile leave fp32 partials [S][9*Ci + 1][Co] which glare_reduce_parts_f32 sums in a fixed order (deterministic)
//   LDS       : ring of 4 x rows [16*nks + 2 (+pad) pixels][64 ci] and 2 g rows [16*nks pixels][64 co], 128 B per pixel, the
//               16-B chunk index XOR-ed with 4*bit1(pixel) (on the DMA source side) so that a transpose read is conflict-free
//   per row   : vmcnt(0) + one barrier, DMA of the row after next, then nks k-steps x 3 tap rows x 3 MFMA 32x32x16 per wave
//               (wave = 32 ci x 32 co x 9 taps = 144 accumulators), fragments read one tap row ahead with counted lgkmcnt
//   borders   : padded / out-of-strip / out-of-range-channel elements carry a DMA offset beyond the buffer descriptor's range,
//               which the hardware turns into zeros
#include "common.h"

extern "C" int glare_reduce_parts_grouped_f32(const float* parts, int n_groups, int n_parts, long long n, float scale, float* out,
                                              int accumulate, glare_stream_t stream);

namespace {

constexpr int CB = 64;             // channel block of a workgroup on either side (2 x 2 waves of 32)
constexpr int PIXB = CB * 2;       // bytes per pixel of a row slot
constexpr int NX = 4, NG = 2;      // ring slots
constexpr int MAX_NKS = 6;
constexpr unsigned OOB = 0x80000000u;

struct WgradParams {
  const a16_t* x;
  const a16_t* g;
  float* parts;
  int B, H, W, xpitch, xoff, Ci, gpitch, Co;
  long long x_gstride, g_gstride;      // elements between the operands of consecutive groups (independent filters, see the entry point)
  int nks, nstrips, ysplits, rps;      // k-steps per strip row, strips per image row, row ranges per strip, rows per range
  int tiles_ci, tiles_co, n_blocks;
  int xslot, gslot;                    // bytes of one ring slot
  int nxi, ngi;                        // DMA instructions per x / g row
};

template <int N>
__device__ __forceinline__ void lgkm_wait() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(N) : "memory"); }
__device__ __forceinline__ void tr_read(u32x2& dst, int addr) { asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(dst) : "v"(addr)); }
__device__ __forceinline__ void pin(u32x2& a) { asm volatile("" : "+v"(a)); }   // nothing that reads `a` moves above this point

__device__ __forceinline__ a16x8 frag(const u32x2& lo, const u32x2& hi) {
  return __builtin_bit_cast(a16x8, u32x4{lo[0], lo[1], hi[0], hi[1]});
}

template <int KS>   // 3: the nine taps of a pad-1 filter; 1: a 1x1 filter (no halo, one tap)
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradParams p) {
  constexpr int TAPS = KS * KS, HALO = KS / 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wci = wave & 1, wco = wave >> 1;

  // consecutive logical ids (the tiles of one split, which read the same rows) on the same XCD
  int bid = blockIdx.x;
  {
    const int n = p.n_blocks, q = n / 8, r = n % 8, xcd = bid % 8, kk = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + kk;
  }
  const int ntiles = p.tiles_ci * p.tiles_co;
  const int t = bid % ntiles, s = bid / ntiles;
  const int tci = t % p.tiles_ci, tco = t / p.tiles_ci;
  const int ys = s % p.ysplits, strip = (s / p.ysplits) % p.nstrips, bg = s / (p.ysplits * p.nstrips);   // bg = group * B + image
  const int grp = bg / p.B, b = bg % p.B;
  const int ws = 16 * p.nks, x0 = strip * ws;
  const int y0 = ys * p.rps, y1 = min(p.H, y0 + p.rps);
  const int ci0 = tci * CB, co0 = tco * CB;

  // ---- DMA geometry: one instruction = 8 pixels x 128 B; lane -> (pixel 8 j + lane / 8, LDS chunk lane % 8), which receives
  // source chunk (lane % 8) ^ 4 * bit1(pixel)
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<a16_t*>(p.x + grp * p.x_gstride + (size_t)b * p.H * p.W * p.xpitch), 0, (int)((long long)p.H * p.W * p.xpitch * 2),
      0x00020000);
  const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<a16_t*>(p.g + grp * p.g_gstride + (size_t)b * p.H * p.W * p.gpitch), 0, (int)((long long)p.H * p.W * p.gpitch * 2),
      0x00020000);
  unsigned xvo[4], gvo[3];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int px = 8 * (wave + 4 * i) + (lane >> 3), c = (lane & 7) ^ (((px >> 1) & 1) << 2);
    const int xg = x0 - HALO + px, ch = ci0 + 8 * c;
    xvo[i] = (xg >= 0 && xg < p.W && px < ws + 2 * HALO && ch < p.Ci) ? (unsigned)(xg * p.xpitch + p.xoff + ch) * 2u : OOB;
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int px = 8 * (wave + 4 * i) + (lane >> 3), c = (lane & 7) ^ (((px >> 1) & 1) << 2);
    const int xg = x0 + px, ch = co0 + 8 * c;
    gvo[i] = (xg < p.W && px < ws && ch < p.Co) ? (unsigned)(xg * p.gpitch + ch) * 2u : OOB;
  }
  const int xrow_bytes = p.W * p.xpitch * 2, grow_bytes = p.W * p.gpitch * 2;
  auto issue_x = [&](int y) {   // row y of the image (zeros outside it) -> slot (y + HALO) & 3
    const bool rv = y >= 0 && y < p.H;
    char* dst = smem + ((y + HALO) & (NX - 1)) * p.xslot;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = wave + 4 * i;
      if (j < p.nxi)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (__attribute__((address_space(3))) void*)(dst + j * 1024), 16, rv ? xvo[i] : OOB,
                                                 rv ? y * xrow_bytes : 0, 0, 0);
    }
  };
  auto issue_g = [&](int y) {   // row y < H -> slot y & 1
    char* dst = smem + NX * p.xslot + (y & (NG - 1)) * p.gslot;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int j = wave + 4 * i;
      if (j < p.ngi)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(gr, (__attribute__((address_space(3))) void*)(dst + j * 1024), 16, gvo[i], y * grow_bytes,
                                                 0, 0);
    }
  };

  // ---- fragment addresses.  Lane i of a 16-lane group supplies the address of 4 consecutive channels of pixel 8 hi + 4 h + (i >> 2)
  // of the k-step and receives [those 4 pixels][channel 16 g4 + i] of the wave's 32-channel block (h = 0, 1: the two halves of
  // the lane's 8 k values).  Tap column tx shifts the pixel by tx (slot pixel 0 is image column x0 - 1).
  const int i16 = lane & 15, g4 = (lane >> 4) & 1, hi = lane >> 5;
  int abase[KS][2], bbase[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int px = 8 * hi + 4 * h + (i16 >> 2);
#pragma unroll
    for (int tx = 0; tx < KS; ++tx) {
      const int q = px + tx, chunk = (wci * 4 + 2 * g4 + ((i16 & 3) >> 1)) ^ (((q >> 1) & 1) << 2);
      abase[tx][h] = q * PIXB + chunk * 16 + (i16 & 1) * 8;
    }
    const int chunk = (wco * 4 + 2 * g4 + ((i16 & 3) >> 1)) ^ (((px >> 1) & 1) << 2);
    bbase[h] = NX * p.xslot + px * PIXB + chunk * 16 + (i16 & 1) * 8;
  }

  f32x16 acc[TAPS], accb;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    accb[r] = 0.f;
#pragma unroll
    for (int tp = 0; tp < TAPS; ++tp) acc[tp][r] = 0.f;
  }
  const bool want_bias = tci == 0 && wci == 0;   // db comes from the ci-block-0 workgroups: one extra MFMA per k-step with A = ones
  const a16x8 ones = __builtin_bit_cast(a16x8, u32x4{A16_ONE * 0x10001u, A16_ONE * 0x10001u, A16_ONE * 0x10001u, A16_ONE * 0x10001u});

  for (int dy = -HALO; dy <= HALO; ++dy) issue_x(y0 + dy);
  issue_g(y0);
  for (int y = y0; y < y1; ++y) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // rows y - HALO .. y + HALO and g row y are in LDS; every wave is done with the rows of step y - 1
    if (y + 1 < y1) {
      issue_x(y + 1 + HALO);
      issue_g(y + 1);
    }
    int xs[KS];
#pragma unroll
    for (int ty = 0; ty < KS; ++ty) xs[ty] = ((y + ty) & (NX - 1)) * p.xslot;   // image row y - HALO + ty
    const int gs = (y & (NG - 1)) * p.gslot;

    u32x2 af[KS][KS][2], bn[2];   // A fragments of tap row ty (KS tap columns x two halves); the next k-step's B halves
    a16x8 bcur;
    auto load_a = [&](int ks, int ty) {
#pragma unroll
      for (int tx = 0; tx < KS; ++tx)
#pragma unroll
        for (int h = 0; h < 2; ++h) tr_read(af[ty][tx][h], abase[tx][h] + xs[ty] + ks * (16 * PIXB));
    };
    auto load_b = [&](int ks) {
#pragma unroll
      for (int h = 0; h < 2; ++h) tr_read(bn[h], bbase[h] + gs + ks * (16 * PIXB));
    };
    auto pin_a = [&](int ty) {
#pragma unroll
      for (int tx = 0; tx < KS; ++tx)
#pragma unroll
        for (int h = 0; h < 2; ++h) pin(af[ty][tx][h]);
    };
    auto mma = [&](int ty) {
#pragma unroll
      for (int tx = 0; tx < KS; ++tx)
        acc[ty * KS + tx] = mfma_a16_32x32x16(frag(af[ty][tx][0], af[ty][tx][1]), bcur, acc[ty * KS + tx], 0, 0, 0);
    };
    if constexpr (KS == 3) {
      load_a(0, 0);
      load_b(0);
      for (int ks = 0; ks < p.nks; ++ks) {
        load_a(ks, 1);                       // 6 reads behind the 8 of (ks, tap row 0) + B
        lgkm_wait<6>();
        pin_a(0); pin(bn[0]); pin(bn[1]);
        bcur = frag(bn[0], bn[1]);
        mma(0);
        if (want_bias) accb = mfma_a16_32x32x16(ones, bcur, accb, 0, 0, 0);
        load_a(ks, 2);
        lgkm_wait<6>();
        pin_a(1);
        mma(1);
        if (ks + 1 < p.nks) {
          load_a(ks + 1, 0);
          load_b(ks + 1);
          lgkm_wait<8>();
        } else {
          lgkm_wait<0>();
        }
        pin_a(2);
        mma(2);
      }
    } else {   // one tap: two fragment sets, the reads of a k-step fly while the previous one multiplies
      u32x2 f0[4], f1[4];   // {A lo, A hi, B lo, B hi}
      auto rd = [&](u32x2(&f)[4], int ks) {
        tr_read(f[0], abase[0][0] + xs[0] + ks * (16 * PIXB));
        tr_read(f[1], abase[0][1] + xs[0] + ks * (16 * PIXB));
        tr_read(f[2], bbase[0] + gs + ks * (16 * PIXB));
        tr_read(f[3], bbase[1] + gs + ks * (16 * PIXB));
      };
      auto mm = [&](u32x2(&f)[4]) {
        pin(f[0]); pin(f[1]); pin(f[2]); pin(f[3]);
        const a16x8 bq = frag(f[2], f[3]);
        acc[0] = mfma_a16_32x32x16(frag(f[0], f[1]), bq, acc[0], 0, 0, 0);
        if (want_bias) accb = mfma_a16_32x32x16(ones, bq, accb, 0, 0, 0);
      };
      rd(f0, 0);
      for (int ks = 0; ks < p.nks; ks += 2) {
        if (ks + 1 < p.nks) { rd(f1, ks + 1); lgkm_wait<4>(); } else { lgkm_wait<0>(); }
        mm(f0);
        if (ks + 1 < p.nks) {
          if (ks + 2 < p.nks) { rd(f0, ks + 2); lgkm_wait<4>(); } else { lgkm_wait<0>(); }
          mm(f1);
        }
      }
    }
  }

  // ---- partial of this split: parts[s][tap * Ci + ci][co], row TAPS * Ci = the bias gradient
  float* out = p.parts + (size_t)s * (TAPS * p.Ci + 1) * p.Co;
  const int co = co0 + wco * 32 + (lane & 31);
  if (co < p.Co) {
#pragma unroll
    for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = ci0 + wci * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (ci < p.Ci) out[(size_t)(tp * p.Ci + ci) * p.Co + co] = acc[tp][r];
      }
    if (want_bias && hi == 0) out[(size_t)TAPS * p.Ci * p.Co + co] = accb[0];
  }
}

struct Plan {
  int nks, nstrips, ysplits, rps, S;   // S = splits per group
};

// strips: the narrowest zero padding of the row, then the fewest strips; row ranges: about 512 workgroups over all tiles and
// groups, at least 4 rows each (every range of a 3x3 filter re-reads 2 halo rows)
Plan make_plan(int G, int B, int H, int W, int Ci, int Co) {
  Plan pl;
  int best_pad = 1 << 30;
  pl.nks = 1; pl.nstrips = 1;
  for (int ns = cdiv(W, 16 * MAX_NKS); ns <= cdiv(W, 16); ++ns) {
    const int nks = cdiv(cdiv(W, ns), 16);
    if (nks > MAX_NKS) continue;
    const int pad = ns * nks * 16 - W;
    if (pad < best_pad) { best_pad = pad; pl.nks = nks; pl.nstrips = ns; }
  }
  const long long tiles = (long long)cdiv(Ci, CB) * cdiv(Co, CB) * G * B * pl.nstrips;
  int ysplits = (int)((512 + tiles - 1) / tiles);
  ysplits = ysplits < 1 ? 1 : ysplits;
  if (ysplits > cdiv(H, 4)) ysplits = cdiv(H, 4);
  pl.rps = cdiv(H, ysplits);
  pl.ysplits = cdiv(H, pl.rps);
  pl.S = B * pl.nstrips * pl.ysplits;
  return pl;
}

// The fixed-order sum of the fp32 partials [S][TAPS * Ci + 1][Co] written straight into the filter's own layout (round 6): dW OIHW
// [Co][ci_total][KS][KS] at input-channel offset ci_off (a conv over torch.cat((x, x2), 1): one launch per source), the bias row into db.
// A gradient that is laid out like its parameter is taken over by autograd's AccumulateGrad as it is; the [row][Co] form reached it as
// a permuted view and was cloned by one copy_ launch per parameter (tools/probes/small_ops.py).
__global__ __launch_bounds__(256) void wgrad_reduce_oihw_kernel(const float* __restrict__ parts, int S, int taps, int Ci, int Co, int ci_total,
                                                                 int ci_off, float* __restrict__ dW, float* __restrict__ db) {
  const long long n = (long long)(taps * Ci + 1) * Co;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int k = 0; k < S; ++k) s += parts[(long long)k * n + i];       // ascending order: the same bits as reduce_parts_kernel
  const int r = (int)(i / Co), co = (int)(i - (long long)r * Co);
  if (r < taps * Ci) {
    const int tap = r / Ci, ci = r - tap * Ci;
    dW[((size_t)co * ci_total + ci_off + ci) * taps + tap] = s;
  } else if (db) {
    db[co] = s;
  }
}

template <int KS>
int wgrad_launch(const void* x, int xpitch, int xoff, long long x_gstride, const void* g, int gpitch, long long g_gstride, float* dWt,
                 int G, int B, int H, int W, int Ci, int Co, void* workspace, size_t workspace_bytes, glare_stream_t stream,
                 float* dW_oihw = nullptr, int ci_total = 0, int ci_off = 0, float* db = nullptr) {
  constexpr int TAPS = KS * KS;
  const bool oihw = dW_oihw != nullptr;
  if (oihw && (G != 1 || ci_off < 0 || ci_off + Ci > ci_total)) return GLARE_ERR_INVALID;
  if (!x || !g || (!dWt && !oihw) || G <= 0 || B <= 0 || H <= 0 || W <= 0 || Ci <= 0 || Co <= 0) return GLARE_ERR_INVALID;
  if (Ci % 8 || Co % 8 || xoff % 8 || xpitch % 8 || gpitch % 8 || xoff + Ci > xpitch || Co > gpitch || x_gstride % 8 || g_gstride % 8)
    return GLARE_ERR_INVALID;
  if (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(g)) & 15) != 0) return GLARE_ERR_INVALID;
  if ((long long)H * W * xpitch * 2 >= (1ll << 31) || (long long)H * W * gpitch * 2 >= (1ll << 31)) return GLARE_ERR_UNSUPPORTED;
  const Plan pl = make_plan(G, B, H, W, Ci, Co);
  const size_t n_out = (size_t)(TAPS * Ci + 1) * Co;
  const size_t need = (pl.S > 1 || oihw) ? (size_t)G * pl.S * n_out * sizeof(float) : 0;     // oihw: the partials always go through the workspace
  if (need && (!workspace || workspace_bytes < need)) return GLARE_ERR_WORKSPACE;
  WgradParams p;
  p.x = static_cast<const a16_t*>(x);
  p.g = static_cast<const a16_t*>(g);
  p.parts = (pl.S > 1 || oihw) ? static_cast<float*>(workspace) : dWt;
  p.B = B; p.H = H; p.W = W; p.xpitch = xpitch; p.xoff = xoff; p.Ci = Ci; p.gpitch = gpitch; p.Co = Co;
  p.x_gstride = x_gstride; p.g_gstride = g_gstride;
  p.nks = pl.nks; p.nstrips = pl.nstrips; p.ysplits = pl.ysplits; p.rps = pl.rps;
  p.tiles_ci = cdiv(Ci, CB); p.tiles_co = cdiv(Co, CB);
  const long long nb = (long long)G * pl.S * p.tiles_ci * p.tiles_co;
  if (nb > 0x7fffffffLL) return GLARE_ERR_UNSUPPORTED;
  p.n_blocks = (int)nb;
  p.nxi = cdiv(16 * pl.nks + 2 * (KS / 2), 8); p.ngi = 2 * pl.nks;
  p.xslot = p.nxi * 1024; p.gslot = p.ngi * 1024;
  const size_t lds = (size_t)NX * p.xslot + (size_t)NG * p.gslot;
  static const hipError_t attr =
      hipFuncSetAttribute((const void*)wgrad_kernel<KS>, hipFuncAttributeMaxDynamicSharedMemorySize, NX * 13 * 1024 + NG * 12 * 1024);
  if (attr != hipSuccess) return GLARE_ERR_LAUNCH;
  hipLaunchKernelGGL(wgrad_kernel<KS>, dim3((unsigned)nb), dim3(256), lds, static_cast<hipStream_t>(stream), p);
  if (oihw) {
    hipLaunchKernelGGL(wgrad_reduce_oihw_kernel, dim3((unsigned)cdivll((long long)n_out, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       (const float*)p.parts, pl.S, TAPS, Ci, Co, ci_total, ci_off, dW_oihw, db);
    return glare_launch_status();
  }
  if (pl.S > 1) return glare_reduce_parts_grouped_f32(p.parts, G, pl.S, (long long)n_out, 1.0f, dWt, 0, stream);
  return glare_launch_status();
}

}  // namespace

extern "C" size_t glare_conv_wgrad_workspace_bytes(int ksize, int groups, int B, int H, int W, int Ci, int Co) {
  if ((ksize != 1 && ksize != 3) || groups <= 0 || B <= 0 || H <= 0 || W <= 0 || Ci <= 0 || Co <= 0) return 0;
  const Plan pl = make_plan(groups, B, H, W, Ci, Co);
  return pl.S > 1 ? (size_t)groups * pl.S * ((size_t)ksize * ksize * Ci + 1) * Co * sizeof(float) : 0;
}

// Weight + bias gradients of `groups` independent ksize x ksize (3: pad 1; 1), stride-1 convolutions of the same shape:
// dWt[grp][(ty*ksize + tx) * Ci + ci][co], row ksize^2 * Ci = the bias gradient; fp32 [groups][ksize^2 * Ci + 1][Co].
// Group grp reads x + grp * x_gstride (bf16 NHWC [B][H][W][xpitch], Ci channels at xoff) and g + grp * g_gstride (bf16 NHWC
// [B][H][W][gpitch], Co channels at 0): a stride of B*H*W*pitch walks step-major tensors, a stride of Ci walks channel blocks of
// one tensor.  Ci, Co, xoff, pitches and strides multiples of 8.
extern "C" int glare_conv_wgrad_bf16(int ksize, const void* x, int xpitch, int xoff, long long x_gstride, const void* g, int gpitch,
                                     long long g_gstride, float* dWt, int groups, int B, int H, int W, int Ci, int Co, void* workspace,
                                     size_t workspace_bytes, glare_stream_t stream) {
  if (ksize == 3)
    return wgrad_launch<3>(x, xpitch, xoff, x_gstride, g, gpitch, g_gstride, dWt, groups, B, H, W, Ci, Co, workspace, workspace_bytes, stream);
  if (ksize == 1)
    return wgrad_launch<1>(x, xpitch, xoff, x_gstride, g, gpitch, g_gstride, dWt, groups, B, H, W, Ci, Co, workspace, workspace_bytes, stream);
  return GLARE_ERR_UNSUPPORTED;
}

// The same for ONE conv with the result in the filter's own layout: dW_oihw fp32 [Co][ci_total][ksize][ksize], this launch filling the
// input channels [ci_off, ci_off + Ci) (a conv over a channel concatenation: one launch per source); db fp32 [Co] or NULL.  The workspace
// is always used (glare_conv_wgrad_oihw_workspace_bytes).
extern "C" size_t glare_conv_wgrad_oihw_workspace_bytes(int ksize, int B, int H, int W, int Ci, int Co) {
  if ((ksize != 1 && ksize != 3) || B <= 0 || H <= 0 || W <= 0 || Ci <= 0 || Co <= 0) return 0;
  const Plan pl = make_plan(1, B, H, W, Ci, Co);
  return (size_t)pl.S * ((size_t)ksize * ksize * Ci + 1) * Co * sizeof(float);
}

extern "C" int glare_conv_wgrad_oihw_bf16(int ksize, const void* x, int xpitch, int xoff, const void* g, int gpitch, float* dW_oihw, int ci_total,
                                          int ci_off, float* db_or_null, int B, int H, int W, int Ci, int Co, void* workspace,
                                          size_t workspace_bytes, glare_stream_t stream) {
  if (!dW_oihw) return GLARE_ERR_INVALID;
  if (ksize == 3)
    return wgrad_launch<3>(x, xpitch, xoff, 0, g, gpitch, 0, nullptr, 1, B, H, W, Ci, Co, workspace, workspace_bytes, stream, dW_oihw, ci_total, ci_off,
                           db_or_null);
  if (ksize == 1)
    return wgrad_launch<1>(x, xpitch, xoff, 0, g, gpitch, 0, nullptr, 1, B, H, W, Ci, Co, workspace, workspace_bytes, stream, dW_oihw, ci_total, ci_off,
                           db_or_null);
  return GLARE_ERR_UNSUPPORTED;
}
