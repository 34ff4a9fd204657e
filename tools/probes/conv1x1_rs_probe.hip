// NOT PART OF THE PRODUCT BUILD: the register-stationary 1x1 kernel of round 6, measured slower than csrc/conv1x1.hip (profiles/r06_conv1x1_rs.txt).
// Kept as the record of that experiment; it compiled as glare_amd/csrc/conv1x1_rs.hip against common.h / glare_hip.h of the round-6 tree.
// 1x1 convolution 512 -> Cout as a REGISTER-stationary GEMM with the activation rows through LDS-DMA (round 6).
//
// Replaces the same nn.Conv2d's as conv1x1.hip at the shape that dominates the step: AttnBlock's folded query / output projections
// (encoder_decoder.py:146-165; per-image filters from glare_attn_fold_groupnorm_f32), 512 -> 512 at the quarter resolution, 16 launches
// per 8-image step.  The weight-stationary kernel keeps its filter slice in LDS and reads the activation as MFMA A fragments straight
// from global memory: every load instruction takes 16 B from each of 64 different 128-B lines (a lane's row is fixed by the MFMA layout)
// and the Cout / 128 co-tiles of a pixel range each pull all of x through their CU's L1 -- bound by the texture path at 0.27 of HBM /
// 0.23 of MFMA.  Here the roles are exchanged:
//   * the product is computed TRANSPOSED, C^T[co][px] = W[co][k] . x^T[k][px]: the FILTER is the A operand and lives in REGISTERS -- a wave
//     keeps its 64 output channels x 512 input channels as 64 A fragments (256 VGPRs; one wave per SIMD owns the whole register file,
//     as in the attention kernel), loaded once per workgroup;
//   * the activation rows go global -> LDS by LDS-DMA, one 1-KB pixel row per instruction (fully coalesced), a ring of four 32-pixel
//     blocks with three requested ahead; the 16-B chunks of a row are XOR-swizzled by the row on the SOURCE side so that the B-fragment reads (32 rows x 16 B at one
//     channel offset) are conflict-free ds_read_b128;
//   * a workgroup = 4 waves x 64 = 256 output channels of one pixel block: x crosses L2 -> LDS Cout / 256 times (twice for 512) instead of
//     Cout / 128 times through per-lane gathers, and every B fragment feeds two MFMAs;
//   * epilogue through a wave-private LDS slab (the C^T layout has a pixel per lane): bias, residual (requested at the start of the
//     block), activation, 16-B stores, and the GroupNorm partial sums of the rounded output in conv1x1.hip's format
//     ([B][row block][Cout / 4][2]: the same c1_gn_reduce_kernel turns them into statistics).
#include "common.h"

namespace {

struct RsParams {
  const a16_t* x;
  const a16_t* w;        // [Cout][512], or per image [B][Cout][512] (w_istride elements apart)
  const float* bias;     // [Cout] or per image (b_istride apart), or null
  const a16_t* res;
  a16_t* out;
  float* gn_part;        // [B][rbi][Cout/4][2] or null
  int B, N, Cout;
  int xpitch, xoff, opitch, ooff, rpitch, roff;
  int act, ncg, rbi;     // co groups of 256 channels, row blocks (32 px) per image
  long long w_istride;
  int b_istride;
};

constexpr int RS_K = 512, RS_KS = RS_K / 16;        // input channels, k-steps
constexpr int RS_TILE = 32 * RS_K * 2;              // one 32-pixel block of x in LDS: 32 KB
constexpr int RS_SLAB = 32 * (128 + 16);            // per wave: 32 px x 64 co 16-bit, rows padded to 144 B (bank spread)
constexpr int RS_NBUF = 4;                          // ring of x blocks: three requested ahead of the one being contracted
constexpr int RS_LDS = RS_NBUF * RS_TILE + 4 * RS_SLAB;

__global__ __launch_bounds__(256, 1) void conv1x1_rs_kernel(const RsParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // block -> (xcd, slot): the co groups of one pixel range sit on one XCD (the second reads x out of that XCD's L2)
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;                 // slot 0..31
  const int cg = slot % p.ncg, rng_in_xcd = slot / p.ncg, rpx = 32 / p.ncg;
  const int range = xcd * rpx + rng_in_xcd, n_ranges = 8 * rpx;
  const int total_rb = p.B * p.rbi;
  int rb_lo = (int)((long long)total_rb * range / n_ranges), rb_hi = (int)((long long)total_rb * (range + 1) / n_ranges);
  int w_img = 0;
  if (p.w_istride != 0) {   // per-image filters: ranges are cut per image (n_ranges % B == 0)
    const int rpi = n_ranges / p.B, sub = range % rpi;
    w_img = range / rpi;
    rb_lo = w_img * p.rbi + (int)((long long)p.rbi * sub / rpi);
    rb_hi = w_img * p.rbi + (int)((long long)p.rbi * (sub + 1) / rpi);
  }
  const int px = lane & 31, khalf = lane >> 5;
  const int co_w = cg * 256 + wave * 64;          // this wave's first output channel

  // ---- the wave's filter slice into registers: A fragment (jt, ks) = rows co_w + 32 jt + px, channels 16 ks + 8 khalf .. + 8
  a16x8 A[2][RS_KS];
  {
    const a16_t* wb = p.w + (size_t)w_img * p.w_istride + (size_t)(co_w + px) * RS_K + 8 * khalf;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int ks = 0; ks < RS_KS; ++ks) A[jt][ks] = *reinterpret_cast<const a16x8*>(wb + (size_t)jt * 32 * RS_K + 16 * ks);
  }
  // bias of the lane's accumulator rows: co = co_w + 32 jt + 8 (r >> 2) + 4 khalf + (r & 3)
  float bias_v[2][16];
#pragma unroll
  for (int jt = 0; jt < 2; ++jt)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      bias_v[jt][r] = p.bias ? p.bias[(size_t)w_img * p.b_istride + co_w + 32 * jt + 8 * (r >> 2) + 4 * khalf + (r & 3)] : 0.f;

  // ---- x block -> LDS: piece = one pixel row (1 KB); lane L fetches the 16-B chunk (L ^ (row & 15)) of it: position L of the LDS row
  auto issue_block = [&](int rb, int buf) {
    const int b = rb / p.rbi, r0 = (rb - b * p.rbi) * 32;
    const int nrows = min(32, p.N - r0);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = wave + 4 * i;
      const int rr = min(row, nrows - 1);                         // ragged last block of an image: the rows beyond it are never stored
      const a16_t* src = p.x + ((size_t)b * p.N + r0 + rr) * p.xpitch + p.xoff + ((lane ^ (row & 15)) * 8);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(smem + buf * RS_TILE + row * 1024), 16, 0, 0);
    }
  };
  char* const slab = smem + RS_NBUF * RS_TILE + wave * RS_SLAB;
  const ActSel asel = act_sel(p.act);

  // A block's contraction is ~1 us (64 MFMAs per wave); a 32-KB block takes longer than that to arrive, so THREE blocks are requested ahead
  // (ring of four 32-KB buffers).  The wait for block rb is a COUNTED one -- the pieces of the blocks requested after it (8 per wave and
  // block) stay in flight; loads return in issue order -- and the residual rows are requested BEFORE the next block's pieces, so that the
  // compiler's own wait in front of their use does not cover those pieces (measured with one block ahead and vmcnt(0): 0.125 ms, slower
  // than the kernel this one replaces).
#pragma unroll
  for (int i = 0; i < RS_NBUF - 1; ++i)
    if (rb_lo + i < rb_hi) issue_block(rb_lo + i, i);
  for (int rb = rb_lo; rb < rb_hi; ++rb) {
    const int buf = (rb - rb_lo) % RS_NBUF;
    const int b = rb / p.rbi, r0 = (rb - b * p.rbi) * 32;
    const int nrows = min(32, p.N - r0);
    const size_t pix0 = (size_t)b * p.N + r0;
    const int ahead = min(RS_NBUF - 2, rb_hi - 1 - rb);    // blocks requested behind this one and still in flight: 2, at the end 1, 0
    if (ahead == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // (not __syncthreads(): with LDS-DMAs in flight hipcc puts s_waitcnt vmcnt(0) in front of it -- the whole ring would drain)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // everybody's pieces of block rb have landed, everybody is done with block rb - 1
    // residual rows of this block (phase 2 of the epilogue consumes them): lane -> row (lane >> 3) + 8 it, chunk lane & 7
    u32x4 resv[4];
    if (p.res) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int m = min((lane >> 3) + 8 * it, nrows - 1);
        resv[it] = *reinterpret_cast<const u32x4*>(p.res + (pix0 + m) * p.rpitch + p.roff + co_w + (lane & 7) * 8);
      }
    }
    if (rb + RS_NBUF - 1 < rb_hi) issue_block(rb + RS_NBUF - 1, (rb - rb_lo + RS_NBUF - 1) % RS_NBUF);   // into the buffer block rb - 1 was read from
    f32x16 acc[2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[jt][r] = bias_v[jt][r];
    const char* xt = smem + buf * RS_TILE + px * 1024;
    // B fragments in a ring, RS_D k-steps ahead of their MFMAs, the order pinned: left to itself hipcc puts every pair of ds_read_b128 right
    // in front of its MFMAs with s_waitcnt lgkmcnt(0) between them -- one wave per SIMD then eats the full LDS latency 16 times per block
    // (measured: 0.128 ms, slower than the kernel this one replaces)
    constexpr int RS_D = 8;
    auto bfrag = [&](int ks) { return *reinterpret_cast<const a16x8*>(xt + (((2 * ks + khalf) ^ (px & 15)) * 16)); };
    a16x8 bf[RS_D];
#pragma unroll
    for (int i = 0; i < RS_D; ++i) bf[i] = bfrag(i);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < RS_KS; ++ks) {
      acc[0] = mfma_a16_32x32x16(A[0][ks], bf[ks % RS_D], acc[0], 0, 0, 0);
      acc[1] = mfma_a16_32x32x16(A[1][ks], bf[ks % RS_D], acc[1], 0, 0, 0);
      if (ks + RS_D < RS_KS) bf[ks % RS_D] = bfrag(ks + RS_D);
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- epilogue.  Phase 1: the lane's pixel row of the slab: 4 consecutive channels per (jt, q) = one 8-B store at channel
    // 32 jt + 8 q + 4 khalf; without a residual the activation is applied here.
    const bool act_early = p.res == nullptr;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = acc[jt][4 * q + e];
          if (act_early) v[e] = act_any(v[e], asel);
        }
        const u32x2 w2 = {pack_a2(v[0], v[1]), pack_a2(v[2], v[3])};
        *reinterpret_cast<u32x2*>(slab + px * 144 + (32 * jt + 8 * q + 4 * khalf) * 2) = w2;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // wave-private slab: no barrier
    // Phase 2: 16-B rows out: lane -> row (lane >> 3) + 8 it, 8-channel chunk lane & 7
    float gs0 = 0.f, gq0 = 0.f, gs1 = 0.f, gq1 = 0.f;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int m = (lane >> 3) + 8 * it, c8 = lane & 7;
      if (m < nrows) {
        u32x4 v = *reinterpret_cast<const u32x4*>(slab + m * 144 + c8 * 16);
        if (p.res) {
          const u32x4 rv = resv[it];
#pragma unroll
          for (int e = 0; e < 4; ++e)
            v[e] = pack_a2(act_any(alo(v[e]) + alo(rv[e]), asel), act_any(ahi(v[e]) + ahi(rv[e]), asel));
        }
        *reinterpret_cast<u32x4*>(p.out + (pix0 + m) * p.opitch + p.ooff + co_w + c8 * 8) = v;
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
    if (p.gn_part) {   // lanes 8 apart hold the same channel chunk
#pragma unroll
      for (int o = 8; o < 64; o <<= 1) {
        gs0 += __shfl_xor(gs0, o, 64); gq0 += __shfl_xor(gq0, o, 64);
        gs1 += __shfl_xor(gs1, o, 64); gq1 += __shfl_xor(gq1, o, 64);
      }
      if (lane < 8) {
        float* dst = p.gn_part + (((size_t)b * p.rbi + (rb - b * p.rbi)) * (p.Cout / 4) + (co_w + lane * 8) / 4) * 2;
        dst[0] = gs0; dst[1] = gq0; dst[2] = gs1; dst[3] = gq1;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // the slab is read out before the next block's phase 1 overwrites it
  }
}

}  // namespace

extern "C" int glare_conv1x1_rs_supported(int Cin, int Cout, int B, int per_image) {
  if (Cin != RS_K || Cout <= 0 || Cout % 256) return 0;
  const int ncg = Cout / 256;
  if (ncg > 32 || 32 % ncg) return 0;
  if (per_image && (B <= 0 || (8 * 32 / ncg) % B != 0)) return 0;
  return 1;
}

// out = act(x . w^T + bias (+ residual)) for Cin = 512, Cout a multiple of 256 (csrc/conv1x1_rs.hip); arguments as
// glare_conv1x1_ws_image_bf16 (w_image_stride = 0: one filter for all images).  The GroupNorm partials have conv1x1.hip's format
// (glare_conv1x1_ws_gn_partial_elems / glare_conv1x1_ws_gn_reduce).
extern "C" int glare_conv1x1_rs_bf16(const void* x, int x_pitch, int x_off, const void* w_bf16, long long w_image_stride, const float* bias,
                                     int bias_image_stride, const void* residual, int res_pitch, int res_off, void* out, int out_pitch,
                                     int out_off, int B, long long pixels_per_image, int Cin, int Cout, int act, float* gn_partial,
                                     glare_stream_t stream) {
  if (!x || !w_bf16 || !out || B <= 0 || pixels_per_image <= 0 || w_image_stride < 0) return GLARE_ERR_INVALID;
  if (!glare_conv1x1_rs_supported(Cin, Cout, B, w_image_stride != 0)) return GLARE_ERR_UNSUPPORTED;
  if (w_image_stride != 0 && (w_image_stride % 8 != 0 || (bias && bias_image_stride <= 0))) return GLARE_ERR_UNSUPPORTED;
  if ((x_pitch % 8) || (x_off % 8) || (out_pitch % 8) || (out_off % 8) || x_off + Cin > x_pitch || out_off + Cout > out_pitch)
    return GLARE_ERR_UNSUPPORTED;
  if (residual && ((res_pitch % 8) || (res_off % 8) || res_off + Cout > res_pitch)) return GLARE_ERR_UNSUPPORTED;
  if (((uintptr_t)x | (uintptr_t)w_bf16 | (uintptr_t)out | (uintptr_t)residual) & 15) return GLARE_ERR_INVALID;
  if (pixels_per_image > 0x7fffffffLL || (long long)B * cdivll(pixels_per_image, 32) > 0x7fffffffLL) return GLARE_ERR_INVALID;
  RsParams p;
  p.x = (const a16_t*)x; p.w = (const a16_t*)w_bf16; p.bias = bias; p.res = (const a16_t*)residual; p.out = (a16_t*)out;
  p.gn_part = gn_partial;
  p.B = B; p.N = (int)pixels_per_image; p.Cout = Cout;
  p.xpitch = x_pitch; p.xoff = x_off; p.opitch = out_pitch; p.ooff = out_off; p.rpitch = res_pitch; p.roff = res_off;
  p.act = act; p.ncg = Cout / 256; p.rbi = (int)cdivll(pixels_per_image, 32);
  p.w_istride = w_image_stride; p.b_istride = w_image_stride != 0 ? bias_image_stride : 0;
  if (hipFuncSetAttribute((const void*)conv1x1_rs_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, RS_LDS) != hipSuccess)
    return GLARE_ERR_LAUNCH;
  hipLaunchKernelGGL(conv1x1_rs_kernel, dim3(256), dim3(256), RS_LDS, (hipStream_t)stream, p);
  return glare_launch_status();
}
