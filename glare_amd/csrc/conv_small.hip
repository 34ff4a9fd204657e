// Direct convolution for the thin layers whose INPUT has at most 4 channels.
//
// Replaces nn.Conv2d for conv_in 3->128 (Encoder, encoder_decoder.py:355), conv_in 3->512
// (Decoder :467, MultiScaleDecoder2 deformableDecoder_arch.py:436), cond_conv 3->64 + sigmoid and
// color_conv 3->3 (ConditionEncoder.py:41-43), quant/post_quant 1x1 3->3 (VQModel_arch.py:46-47).
// These are HBM-bound (27 MACs per output element): no MFMA, fp32 input and arithmetic, each lane
// produces 8 consecutive output channels of one pixel (one 16-B store), weights sit in LDS as
// [tap][ci][Cout] so the 8 weights of a lane are one ds_read_b128 pair.
// Input is addressed with explicit strides, so both the harness' NCHW image and the NHWC fp32
// latent (tokens x 3) are read in place.
#include "common.h"

namespace {

constexpr int CS_THREADS = 256;

template <int ACT>
__device__ __forceinline__ float act_apply(float v) {
  if (ACT == GLARE_ACT_SIGMOID) return sigmoidf_(v);
  if (ACT == GLARE_ACT_RELU) return fmaxf(v, 0.f);
  if (ACT == GLARE_ACT_SWISH) return swishf_(v);
  return v;
}

// ACT and the store form are compile-time: as a run-time `if` chain per output element they were ~430 branches in the ISA and the
// kernel ran at 0.78 TB/s of its 4+ TB/s HBM bound (0.71 ms for conv_in 3 -> 128 at 8 x 420 x 620); border pixels are clamped
// loads + selects instead of branches.
// (Loading all KS*KS*Cin input rows of a lane up front -- one memory latency per item instead of nine -- was tried: hipcc then
// preloads every weight as well, 256 registers or hundreds of spills under a register bound; 1.3 - 8.9 ms instead of 0.53.)
template <int KS, int ACT>
__global__ __launch_bounds__(CS_THREADS) void conv_small_kernel(
    const float* __restrict__ x, long long sb, long long sc, long long sy, long long sx, const float* __restrict__ w,
    const float* __restrict__ bias, void* __restrict__ out, int B, int H, int W, int Cin, int Cout, int out_pitch,
    int out_off, int out_f32, a16_t* __restrict__ out_lo) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* wl = reinterpret_cast<float*>(smem);  // [KS*KS][Cin][CoutP], then the bias [CoutP]
  const int CoutP = (Cout + 7) & ~7;
  const int taps = KS * KS;
  for (int i = threadIdx.x; i < taps * Cin * CoutP; i += CS_THREADS) {
    const int co = i % CoutP, ci = (i / CoutP) % Cin, t = i / (CoutP * Cin);
    wl[i] = co < Cout ? w[((size_t)co * Cin + ci) * taps + t] : 0.f;
  }
  float* bl = wl + taps * Cin * CoutP;
  for (int i = threadIdx.x; i < CoutP; i += CS_THREADS) bl[i] = (bias && i < Cout) ? bias[i] : 0.f;
  __syncthreads();
  // Each lane produces 8 consecutive output channels of PX consecutive pixels of a row: the 8 weights of a (tap, ci) are read
  // from LDS once and reused for PX pixels (the kernel was LDS-read bound at one pixel per lane: 2 ds_read_b128 per 8 FMAs).
  constexpr int PX = 4;
  const int groups = CoutP / 8, WQ = (W + PX - 1) / PX;
  const long long total = (long long)B * H * WQ * groups;
  const bool vec_store = !out_f32 && ((out_pitch | out_off) % 8) == 0;
  for (long long idx = (long long)blockIdx.x * CS_THREADS + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * CS_THREADS) {
    const int g = idx % groups;
    long long t2 = idx / groups;
    const int xq = t2 % WQ;
    t2 /= WQ;
    const int yh = t2 % H;
    const int b = t2 / H;
    const int x0 = xq * PX;
    float acc[PX][8];
    {
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(bl + g * 8), b1 = *reinterpret_cast<const f32x4*>(bl + g * 8 + 4);
#pragma unroll
      for (int q = 0; q < PX; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[q][e] = b0[e]; acc[q][4 + e] = b1[e]; }
    }
    // column validity and clamped column offsets of the PX + KS - 1 inputs of a row segment (the same for every row and channel)
    long long xo[PX + KS - 1];
    bool xok[PX + KS - 1];
#pragma unroll
    for (int k = 0; k < PX + KS - 1; ++k) {
      const int ix = x0 + k - KS / 2;
      xok[k] = ix >= 0 && ix < W;
      xo[k] = (long long)min(max(ix, 0), W - 1) * sx;
    }
    auto fma_row = [&](int ty, int ci, const float (&v)[PX + KS - 1]) {
#pragma unroll
      for (int tx = 0; tx < KS; ++tx) {
        const f32x4* wp = reinterpret_cast<const f32x4*>(wl + ((size_t)(ty * KS + tx) * Cin + ci) * CoutP + g * 8);
        const f32x4 w0 = wp[0], w1 = wp[1];
#pragma unroll
        for (int q = 0; q < PX; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc[q][e] = fmaf(v[q + tx], w0[e], acc[q][e]);
            acc[q][4 + e] = fmaf(v[q + tx], w1[e], acc[q][4 + e]);
          }
      }
    };
#pragma unroll
    for (int ty = 0; ty < KS; ++ty) {
      const int iy = yh + ty - KS / 2;
      const bool yok = iy >= 0 && iy < H;
      const float* xrow = x + b * sb + (long long)min(max(iy, 0), H - 1) * sy;
      for (int ci = 0; ci < Cin; ++ci) {
        float v[PX + KS - 1];
        const float* xp = xrow + ci * sc;
#pragma unroll
        for (int k = 0; k < PX + KS - 1; ++k) {
          const float t = xp[xo[k]];
          v[k] = (yok && xok[k]) ? t : 0.f;
        }
        fma_row(ty, ci, v);
      }
    }
#pragma unroll
    for (int q = 0; q < PX; ++q) {
      const int xw = x0 + q;
      if (xw >= W) break;
      const size_t opix = ((size_t)b * H + yh) * W + xw;
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[q][e] = act_apply<ACT>(acc[q][e]);
      if (vec_store && g * 8 + 8 <= Cout) {
        a16_t* o = reinterpret_cast<a16_t*>(out) + opix * out_pitch + out_off + g * 8;
        const u32x4 hi = u32x4{pack_a2(acc[q][0], acc[q][1]), pack_a2(acc[q][2], acc[q][3]), pack_a2(acc[q][4], acc[q][5]),
                               pack_a2(acc[q][6], acc[q][7])};
        *reinterpret_cast<u32x4*>(o) = hi;
        if (out_lo) {   // hi / lo pair (glare_conv_desc.out_lo): the remainder of the 16-bit rounding, same pitch / offset
          u32x4 lo;
#pragma unroll
          for (int e = 0; e < 4; ++e) lo[e] = pack_a2(acc[q][2 * e] - alo(hi[e]), acc[q][2 * e + 1] - ahi(hi[e]));
          *reinterpret_cast<u32x4*>(out_lo + opix * out_pitch + out_off + g * 8) = lo;
        }
      } else if (out_f32) {
        float* o = reinterpret_cast<float*>(out) + opix * out_pitch + out_off + g * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (g * 8 + e < Cout) o[e] = acc[q][e];
      } else {
        a16_t* o = reinterpret_cast<a16_t*>(out) + opix * out_pitch + out_off + g * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (g * 8 + e < Cout) o[e] = f2a(acc[q][e]);
      }
    }
  }
}


// Cin == 3, 3x3 (conv_in of the three encoders / decoders, cond_conv, color_conv): the input patches of a block's pixel quads go through
// LDS.  The generic kernel above fetches a lane's 54 inputs in nine dependent global-load rounds (and the 8 ... 64 lanes that share a
// quad fetch them 8 ... 64 times); here the block's 256 / groups quads are staged once per iteration -- [quad][ty][ci][6] fp32, masked
// and clamped at staging time, the NEXT iteration's patch requested before this one's FMAs and written behind them into the other
// buffer, one barrier per iteration -- and the lanes read them back as broadcast ds_read_b64.  The FMA order per output element is the
// generic kernel's (bias; then ty, ci, tx): the results are bit-identical.
template <int ACT>
__global__ __launch_bounds__(CS_THREADS) void conv_small3_kernel(
    const float* __restrict__ x, long long sb, long long sc, long long sy, long long sx, const float* __restrict__ w,
    const float* __restrict__ bias, void* __restrict__ out, int B, int H, int W, int Cout, int out_pitch, int out_off, int out_f32,
    a16_t* __restrict__ out_lo) {
  constexpr int KS = 3, Cin = 3, PX = 4, PW = PX + KS - 1, PATCH = KS * Cin * PW;   // 54 fp32 per quad
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* wl = reinterpret_cast<float*>(smem);  // [9][3][CoutP], the bias [CoutP], then the two patch buffers
  const int CoutP = (Cout + 7) & ~7;
  for (int i = threadIdx.x; i < 9 * Cin * CoutP; i += CS_THREADS) {
    const int co = i % CoutP, ci = (i / CoutP) % Cin, t = i / (CoutP * Cin);
    wl[i] = co < Cout ? w[((size_t)co * Cin + ci) * 9 + t] : 0.f;
  }
  float* bl = wl + 9 * Cin * CoutP;
  for (int i = threadIdx.x; i < CoutP; i += CS_THREADS) bl[i] = (bias && i < Cout) ? bias[i] : 0.f;
  const int groups = CoutP / 8, NQ = CS_THREADS / groups, WQ = (W + PX - 1) / PX;   // groups divides 256 (launcher)
  float* patch = bl + CoutP;                     // [2][NQ][PATCH]
  const long long totalQ = (long long)B * H * WQ;
  const long long n_iter = (totalQ + NQ - 1) / NQ;
  constexpr int MAXE = 7;                        // staged elements per thread: NQ * 54 / 256 <= 6.75 (groups >= 8)
  const int n_el = NQ * PATCH;
  float stage[MAXE];
  auto stage_load = [&](long long it) {
#pragma unroll
    for (int u = 0; u < MAXE; ++u) {
      const int e = threadIdx.x + u * CS_THREADS;
      stage[u] = 0.f;
      if (e < n_el) {
        const int ql = e / PATCH, r = e % PATCH;
        const int ty = r / (Cin * PW), ci = (r / PW) % Cin, k = r % PW;
        const long long Q = it * NQ + ql;
        if (Q < totalQ) {
          const int xq = (int)(Q % WQ);
          const long long t2 = Q / WQ;
          const int yh = (int)(t2 % H), b = (int)(t2 / H);
          const int iy = yh + ty - 1, ix = xq * PX + k - 1;
          if (iy >= 0 && iy < H && ix >= 0 && ix < W) stage[u] = x[b * sb + ci * sc + iy * sy + ix * sx];
        }
      }
    }
  };
  auto stage_store = [&](int buf) {
#pragma unroll
    for (int u = 0; u < MAXE; ++u) {
      const int e = threadIdx.x + u * CS_THREADS;
      if (e < n_el) patch[buf * n_el + e] = stage[u];
    }
  };
  const int g = threadIdx.x % groups, ql = threadIdx.x / groups;
  const bool vec_store = !out_f32 && ((out_pitch | out_off) % 8) == 0;
  long long it = blockIdx.x;
  if (it < n_iter) stage_load(it);
  if (it < n_iter) stage_store(0);
  __syncthreads();                               // weights, bias and the first patch
  for (int buf = 0; it < n_iter; it += gridDim.x, buf ^= 1) {
    const long long nxt = it + gridDim.x;
    if (nxt < n_iter) stage_load(nxt);           // in flight under the FMAs
    const long long Q = it * NQ + ql;
    if (Q < totalQ) {
      const int xq = (int)(Q % WQ);
      const long long t2 = Q / WQ;
      const int yh = (int)(t2 % H), b = (int)(t2 / H);
      const int x0 = xq * PX;
      float acc[PX][8];
      {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(bl + g * 8), b1 = *reinterpret_cast<const f32x4*>(bl + g * 8 + 4);
#pragma unroll
        for (int q = 0; q < PX; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) { acc[q][e] = b0[e]; acc[q][4 + e] = b1[e]; }
      }
      const float* pq = patch + buf * n_el + ql * PATCH;
#pragma unroll
      for (int ty = 0; ty < KS; ++ty) {
#pragma nounroll      // (unrolled, hipcc reads all 54 weight vectors ahead: 256 registers, one wave per SIMD)
        for (int ci = 0; ci < Cin; ++ci) {
          float v[PW];
          const f32x2* pv = reinterpret_cast<const f32x2*>(pq + (ty * Cin + ci) * PW);
#pragma unroll
          for (int k = 0; k < PW / 2; ++k) { const f32x2 t = pv[k]; v[2 * k] = t[0]; v[2 * k + 1] = t[1]; }
#pragma unroll
          for (int tx = 0; tx < KS; ++tx) {
            const f32x4* wp = reinterpret_cast<const f32x4*>(wl + ((size_t)(ty * KS + tx) * Cin + ci) * CoutP + g * 8);
            const f32x4 w0 = wp[0], w1 = wp[1];
#pragma unroll
            for (int q = 0; q < PX; ++q)
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                acc[q][e] = fmaf(v[q + tx], w0[e], acc[q][e]);
                acc[q][4 + e] = fmaf(v[q + tx], w1[e], acc[q][4 + e]);
              }
          }
        }
      }
#pragma unroll
      for (int q = 0; q < PX; ++q) {
        const int xw = x0 + q;
        if (xw >= W) break;
        const size_t opix = ((size_t)b * H + yh) * W + xw;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[q][e] = act_apply<ACT>(acc[q][e]);
        if (vec_store && g * 8 + 8 <= Cout) {
          a16_t* o = reinterpret_cast<a16_t*>(out) + opix * out_pitch + out_off + g * 8;
          const u32x4 hi = u32x4{pack_a2(acc[q][0], acc[q][1]), pack_a2(acc[q][2], acc[q][3]), pack_a2(acc[q][4], acc[q][5]),
                                 pack_a2(acc[q][6], acc[q][7])};
          *reinterpret_cast<u32x4*>(o) = hi;
          if (out_lo) {
            u32x4 lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) lo[e] = pack_a2(acc[q][2 * e] - alo(hi[e]), acc[q][2 * e + 1] - ahi(hi[e]));
            *reinterpret_cast<u32x4*>(out_lo + opix * out_pitch + out_off + g * 8) = lo;
          }
        } else if (out_f32) {
          float* o = reinterpret_cast<float*>(out) + opix * out_pitch + out_off + g * 8;
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (g * 8 + e < Cout) o[e] = acc[q][e];
        } else {
          a16_t* o = reinterpret_cast<a16_t*>(out) + opix * out_pitch + out_off + g * 8;
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (g * 8 + e < Cout) o[e] = f2a(acc[q][e]);
        }
      }
    }
    if (nxt < n_iter) stage_store(buf ^ 1);
    __syncthreads();                             // the next patch is visible; everybody is done with this one
  }
}

template <int KS>
void launch_small(int act, unsigned blocks, size_t lds, hipStream_t st, const float* x, long long sb, long long sc, long long sy,
                  long long sx, const float* w, const float* bias, void* out, int B, int H, int W, int Cin, int Cout, int out_pitch,
                  int out_off, int out_f32, a16_t* out_lo) {
#define CS_LAUNCH(A)                                                                                                          \
  hipLaunchKernelGGL((conv_small_kernel<KS, A>), dim3(blocks), dim3(CS_THREADS), lds, st, x, sb, sc, sy, sx, w, bias, out, B, H, W, \
                     Cin, Cout, out_pitch, out_off, out_f32, out_lo)
  switch (act) {
    case GLARE_ACT_SIGMOID: CS_LAUNCH(GLARE_ACT_SIGMOID); break;
    case GLARE_ACT_RELU: CS_LAUNCH(GLARE_ACT_RELU); break;
    case GLARE_ACT_SWISH: CS_LAUNCH(GLARE_ACT_SWISH); break;
    default: CS_LAUNCH(GLARE_ACT_NONE);
  }
#undef CS_LAUNCH
}

}  // namespace

static int smallcin_launch(const float* x, long long stride_b, long long stride_c, long long stride_y, long long stride_x,
                           const float* w_oihw, const float* bias, void* out, int B, int H, int W, int Cin, int Cout, int ksize,
                           int out_pitch, int out_off, int act, int out_is_f32, void* out_lo, glare_stream_t stream) {
  if (!x || !w_oihw || !out || B <= 0 || H <= 0 || W <= 0 || Cout <= 0) return GLARE_ERR_INVALID;
  if (out_lo && (out_is_f32 || (out_pitch % 8) || (out_off % 8) || (Cout % 8))) return GLARE_ERR_UNSUPPORTED;
  if (Cin < 1 || Cin > 4 || (ksize != 1 && ksize != 3)) return GLARE_ERR_UNSUPPORTED;
  if (out_off + Cout > out_pitch) return GLARE_ERR_INVALID;
  const int CoutP = (Cout + 7) & ~7;
  const size_t lds = ((size_t)ksize * ksize * Cin + 1) * CoutP * sizeof(float);   // weights + bias
  if (lds > 64 * 1024) return GLARE_ERR_UNSUPPORTED;
  const long long total = (long long)B * H * ((W + 3) / 4) * (CoutP / 8);   // 4 pixels per lane
  long long blocks = (total + CS_THREADS - 1) / CS_THREADS;
  if (blocks > 256 * 16) blocks = 256 * 16;
  const int groups = CoutP / 8;
  if (ksize == 3 && Cin == 3 && groups >= 8 && groups <= 64 && (groups & (groups - 1)) == 0) {   // the LDS-staged form (conv_small3_kernel)
    const int NQ = CS_THREADS / groups;
    const size_t lds3 = lds + (size_t)2 * NQ * 54 * sizeof(float);
    const long long n_iter = ((long long)B * H * ((W + 3) / 4) + NQ - 1) / NQ;
    const unsigned blocks3 = (unsigned)(n_iter < 256 * 8 ? n_iter : 256 * 8);
    if (lds3 <= 64 * 1024) {
#define CS3_LAUNCH(A)                                                                                                              \
  hipLaunchKernelGGL((conv_small3_kernel<A>), dim3(blocks3), dim3(CS_THREADS), lds3, (hipStream_t)stream, x, stride_b, stride_c, stride_y, \
                     stride_x, w_oihw, bias, out, B, H, W, Cout, out_pitch, out_off, out_is_f32, (a16_t*)out_lo)
      switch (act) {
        case GLARE_ACT_SIGMOID: CS3_LAUNCH(GLARE_ACT_SIGMOID); break;
        case GLARE_ACT_RELU: CS3_LAUNCH(GLARE_ACT_RELU); break;
        case GLARE_ACT_SWISH: CS3_LAUNCH(GLARE_ACT_SWISH); break;
        default: CS3_LAUNCH(GLARE_ACT_NONE);
      }
#undef CS3_LAUNCH
      return glare_launch_status();
    }
  }
  if (ksize == 3)
    launch_small<3>(act, (unsigned)blocks, lds, (hipStream_t)stream, x, stride_b, stride_c, stride_y, stride_x, w_oihw, bias, out, B, H, W,
                    Cin, Cout, out_pitch, out_off, out_is_f32, (a16_t*)out_lo);
  else
    launch_small<1>(act, (unsigned)blocks, lds, (hipStream_t)stream, x, stride_b, stride_c, stride_y, stride_x, w_oihw, bias, out, B, H, W,
                    Cin, Cout, out_pitch, out_off, out_is_f32, (a16_t*)out_lo);
  return glare_launch_status();
}

extern "C" int glare_conv2d_smallcin_f32(const float* x, long long stride_b, long long stride_c, long long stride_y,
                                         long long stride_x, const float* w_oihw, const float* bias, void* out, int B,
                                         int H, int W, int Cin, int Cout, int ksize, int out_pitch, int out_off, int act,
                                         int out_is_f32, glare_stream_t stream) {
  return smallcin_launch(x, stride_b, stride_c, stride_y, stride_x, w_oihw, bias, out, B, H, W, Cin, Cout, ksize, out_pitch, out_off, act,
                         out_is_f32, nullptr, stream);
}

// The same with the output as a hi / lo pair (16-bit NHWC, Cout / pitch / offset multiples of 8; glare_conv_desc.out_lo): conv_in
// of the conditional encoder opens the 22-bit residual stream straight from its fp32 accumulators.
extern "C" int glare_conv2d_smallcin_hilo_f32(const float* x, long long stride_b, long long stride_c, long long stride_y,
                                              long long stride_x, const float* w_oihw, const float* bias, void* out_hi, void* out_lo,
                                              int B, int H, int W, int Cin, int Cout, int ksize, int out_pitch, int out_off, int act,
                                              glare_stream_t stream) {
  if (!out_lo) return GLARE_ERR_INVALID;
  return smallcin_launch(x, stride_b, stride_c, stride_y, stride_x, w_oihw, bias, out_hi, B, H, W, Cin, Cout, ksize, out_pitch, out_off, act,
                         0, out_lo, stream);
}
