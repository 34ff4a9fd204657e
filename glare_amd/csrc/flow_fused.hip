// One coupling step of the conditional flow's reverse (sampling) pass as ONE kernel (round 6).
//
// Reference: FlowStep.reverse_flow (FlowStep.py:100-119) -> CondAffineSeparatedAndCond.forward(reverse=True)
// (FlowAffineCouplingsAblation.py:83-110): h = fAffine(cat[z[:, :1], ft]) with fAffine = Conv2d 3x3 65 -> 64, ReLU,
// Conv2d 1x1 64 -> 64, ReLU, Conv2dZeros 3x3 64 -> 4 (:143-151), then the two affine updates, invconv^-1 and actnorm^-1.
// Rounds 1-5 ran a step as four launches (flow.hip: flow_h1 -> 1x1 MFMA conv -> 3x3 MFMA conv -> flow_tail) of 5-56 us each
// on a strictly sequential chain of 24 steps: ~120 us per step for ~10 us of arithmetic, with h1 / h2 / h4 round-tripping HBM.
// Here a workgroup owns a 16 x 40 pixel tile of the latent and recomputes the one-pixel halo of h2 (18 x 42 = 756 pixels in 24
// blocks of 32, three per wave); the whole net runs on MFMA in the fp32-class form (every operand a hi / lo pair of 16-bit values, three
// products per contraction, fp32 accumulation) with the activations CHAINED THROUGH REGISTERS:
//   * every product is computed TRANSPOSED, C^T[channel][pixel] = W[channel][k] . X^T[k][pixel]: the weights are the A operand, the
//     32 pixels of a block the B operand's columns.  The C layout of v_mfma_f32_32x32x16 (lane = column, rows (r & 3) + 8 (r >> 2) +
//     4 (lane >> 5)) then IS a valid B-operand layout of the next product -- a lane keeps its own pixel, and the eight accumulator
//     registers r = 8u .. 8u + 7 are the eight k values of k-step u (the contraction order is free as long as A agrees: the host
//     packs the filters in that channel order, glare_amd.ops.flow_fused_image).  h1 and h2 never leave the registers;
//   * h1 = relu(ftA + conv3x3(z0 -> 64)): the 9-tap, 1-channel conv is one more K = 16 product (B = the lane's 9 neighbours of z0
//     out of a 20 x 44 LDS patch), the accumulator is INITIALISED with the z-independent part ftA (fp32, batched for all steps
//     before the loop); the 1x1 conv's accumulator is initialised with its bias;
//   * the 3x3 conv 64 -> 4 is a "1x1" product with (tap, cout) = 36 output rows per h2 pixel, S[tap * 4 + co][pixel] in LDS, and a
//     9-term shift-add per interior pixel afterwards -- 2.8x fewer MFMAs than nine K = 64 products with 4 of 32 columns used;
//   * the tail (both affine updates, M z + t) runs on the interior pixel's thread.  z is read from `z_in` and written to `z_out`
//     (the halo of a neighbouring tile reads channel 0 of pixels this tile writes: the caller ping-pongs two buffers).
// 54 MFMAs per 32 pixels; LDS: 36 KB filter image + 112 KB S + the z0 patch = 152 KB, one 8-wave workgroup per CU.  One launch per step:
// 19 us of device time at 8 x 105 x 155 against ~72 us for the four launches (tools/kbench.py flow).
#include "common.h"

namespace {

// Tile geometry as a template: TH x TW interior pixels per workgroup of WAVES waves.  The h1 / h2 region is (TH + 2) x (TW + 2) pixels in
// blocks of 32, S holds 36 rows of it, the z0 patch is (TH + 4) x (TW + 4).
template <int TH_, int TW_, int WAVES_>
struct FsGeom {
  static constexpr int TH = TH_, TW = TW_, WAVES = WAVES_, THREADS = 64 * WAVES_;
  static constexpr int RH = TH + 2, RW = TW + 2, RP = RH * RW;
  static constexpr int NB = (RP + 31) / 32;
  static constexpr int SP = NB * 32 + 8;          // S row pitch in floats (4 * SP % 64 == 32: the two half-waves of a store hit disjoint banks)
  static constexpr int ZH = TH + 4, ZW = TW + 4;
};
// filter image (ops.flow_fused_image): A fragments, 1 KB each = [half][row 0..31][8 a16]
constexpr int FS_OFF_WZ_HI = 0, FS_OFF_WZ_LO = 2048;    // [jt] : 2 fragments each
constexpr int FS_OFF_W2_HI = 4096, FS_OFF_W2_LO = 12288;   // [jt][ks] : 8 fragments each
constexpr int FS_OFF_W4_HI = 20480, FS_OFF_W4_LO = 28672;  // [jt][ks] : 8 fragments each
constexpr int FS_OFF_B2 = 36864;                        // fp32 [jt][half][16]
constexpr int FS_OFF_B4 = FS_OFF_B2 + 256;              // fp32 [4]
constexpr int FS_IMG_BYTES = FS_OFF_B4 + 16;            // 37136
constexpr int FS_LDS_S = ((FS_IMG_BYTES + 255) / 256) * 256;
template <typename G> constexpr int fs_lds_z() { return FS_LDS_S + 36 * G::SP * 4; }
template <typename G> constexpr int fs_lds_bytes() { return fs_lds_z<G>() + G::ZH * G::ZW * 4; }

struct FsParams {
  const float* z_in;
  float* z_out;
  const float* ftA;
  const char* image;
  const float* hF;
  int a_pitch, a_off, f_pitch, f_off;
  int B, H, W, tiles_x, tiles_y;
  float M[9], t[3], eps;
};

__device__ __forceinline__ float fs_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }   // as flow.hip's sigmoid_acc

// eight fp32 values -> the hi / lo pair of 8-element a16 fragments (hi = round16(v), lo = round16(v - hi))
__device__ __forceinline__ void fs_split8(const float* v, a16x8& hi, a16x8& lo) {
  u32x4 h, l;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    h[e] = pack_a2(v[2 * e], v[2 * e + 1]);
    l[e] = pack_a2(v[2 * e] - alo(h[e]), v[2 * e + 1] - ahi(h[e]));
  }
  hi = __builtin_bit_cast(a16x8, h);
  lo = __builtin_bit_cast(a16x8, l);
}

// acc += W . X in the fp32-class form: (w_hi . x_hi + w_lo . x_hi) + w_hi . x_lo
__device__ __forceinline__ f32x16 fs_mfma3(const a16x8& ah, const a16x8& al, const a16x8& xh, const a16x8& xl, f32x16 acc) {
  acc = mfma_a16_32x32x16(ah, xh, acc, 0, 0, 0);
  acc = mfma_a16_32x32x16(al, xh, acc, 0, 0, 0);
  acc = mfma_a16_32x32x16(ah, xl, acc, 0, 0, 0);
  return acc;
}

template <typename G>
__global__ __launch_bounds__(G::THREADS, 1) void flow_step_fused_kernel(const FsParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* const S = reinterpret_cast<float*>(smem + FS_LDS_S);
  float* const zp = reinterpret_cast<float*>(smem + fs_lds_z<G>());
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile = blockIdx.x % (p.tiles_x * p.tiles_y), b = blockIdx.x / (p.tiles_x * p.tiles_y);
  const int y0 = (tile / p.tiles_x) * G::TH, x0 = (tile % p.tiles_x) * G::TW;
  const size_t img0 = (size_t)b * p.H * p.W;

  // ---- filter image and the z0 patch (channel 0 of z, zero outside the image) into LDS
  for (int i = tid; i < FS_IMG_BYTES / 16; i += G::THREADS)
    *reinterpret_cast<u32x4*>(smem + i * 16) = *reinterpret_cast<const u32x4*>(p.image + i * 16);
  for (int i = tid; i < G::ZH * G::ZW; i += G::THREADS) {
    const int zy = y0 - 2 + i / G::ZW, zx = x0 - 2 + i % G::ZW;
    float v = 0.f;
    if (zy >= 0 && zy < p.H && zx >= 0 && zx < p.W) v = p.z_in[(img0 + (size_t)zy * p.W + zx) * 3];
    zp[i] = v;
  }
  __syncthreads();

  auto frag = [&](int off) { return *reinterpret_cast<const a16x8*>(smem + off + lane * 16); };
  const int half = lane >> 5;
  for (int blk = wave; blk < G::NB; blk += G::WAVES) {
    const int pr = blk * 32 + (lane & 31);            // region pixel of this lane (>= RP: padding of the last block)
    const int pc = min(pr, G::RP - 1);
    const int ry = pc / G::RW, rx = pc - ry * G::RW;
    const int gy = y0 - 1 + ry, gx = x0 - 1 + rx;
    const bool inimg = pr < G::RP && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
    const int cy = min(max(gy, 0), p.H - 1), cx = min(max(gx, 0), p.W - 1);
    const float* arow = p.ftA + (img0 + (size_t)cy * p.W + cx) * p.a_pitch + p.a_off + 4 * half;

    // h1^T = ftA + wz . Z : the accumulators start as the z-independent part (rows (r & 3) + 8 (r >> 2) + 4 half of tile jt).
    // (Requesting these rows one block ahead, and the tail's operands before the image load, measured 5 % SLOWER: the kernel is
    // bound by its vector-ALU work -- the hi / lo splits -- next to the MFMAs, not by these latencies.)
    f32x16 acc[2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(arow + 32 * jt + 8 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[jt][4 * q + e] = v[e];
      }
    // Z fragment: k = tap 8 half + i (taps >= 9 are zero); patch pixel of tap t is (ry + t / 3, rx + t % 3)
    a16x8 zh, zl;
    {
      float zt[8];
      const float* zc = zp + ry * G::ZW + rx;
      if (half == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) zt[i] = zc[(i / 3) * G::ZW + (i % 3)];
      } else {
        zt[0] = zc[2 * G::ZW + 2];
#pragma unroll
        for (int i = 1; i < 8; ++i) zt[i] = 0.f;
      }
      fs_split8(zt, zh, zl);
    }
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
      acc[jt] = fs_mfma3(frag(FS_OFF_WZ_HI + jt * 1024), frag(FS_OFF_WZ_LO + jt * 1024), zh, zl, acc[jt]);
    }

    // relu, split: B fragments of the 1x1 product, k-step (j, u) = accumulator registers 8u .. 8u + 7 of tile j
    a16x8 xh[4], xl[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = fmaxf(acc[ks >> 1][8 * (ks & 1) + i], 0.f);
      fs_split8(v, xh[ks], xl[ks]);
    }
    // h2^T = relu(W2 . h1^T + b2), zero outside the image (the 3x3 conv that follows pads h2 with zeros)
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
      const float* bb = reinterpret_cast<const float*>(smem + FS_OFF_B2) + (jt * 2 + half) * 16;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(bb + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[jt][4 * q + e] = v[e];
      }
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)           // the two tiles' chains interleaved: consecutive MFMAs never wait for each other's result
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) {
        acc[jt] = fs_mfma3(frag(FS_OFF_W2_HI + (jt * 4 + ks) * 1024), frag(FS_OFF_W2_LO + (jt * 4 + ks) * 1024), xh[ks], xl[ks], acc[jt]);
      }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = fmaxf(acc[ks >> 1][8 * (ks & 1) + i], 0.f);
      fs_split8(v, xh[ks], xl[ks]);
      if (!inimg) {                                   // (the select on the 8 packed registers of a k-step, not on its 8 values)
        xh[ks] = __builtin_bit_cast(a16x8, u32x4{0u, 0u, 0u, 0u});
        xl[ks] = xh[ks];
      }
    }
    // S^T[tap * 4 + co][pixel] = W4r . h2^T  (36 rows: tile 0 whole, rows 32 .. 35 of tile 1)
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[jt][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) {
        acc[jt] = fs_mfma3(frag(FS_OFF_W4_HI + (jt * 4 + ks) * 1024), frag(FS_OFF_W4_LO + (jt * 4 + ks) * 1024), xh[ks], xl[ks], acc[jt]);
      }
#pragma unroll
    for (int r = 0; r < 16; ++r) S[((r & 3) + 8 * (r >> 2) + 4 * half) * G::SP + pr] = acc[0][r];
    if (half == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) S[(32 + r) * G::SP + pr] = acc[1][r];
    }
  }
  __syncthreads();

  // ---- interior pixels: 9-tap shift-add of S, then the step's tail (flow.hip flow_tail_kernel, same op order)
  const float* b4 = reinterpret_cast<const float*>(smem + FS_OFF_B4);
  for (int q = tid; q < G::TH * G::TW; q += G::THREADS) {
    const int iy = q / G::TW, ix = q - iy * G::TW;
    const int gy = y0 + iy, gx = x0 + ix;
    if (gy >= p.H || gx >= p.W) continue;
    float h[4] = {b4[0], b4[1], b4[2], b4[3]};
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const float* s = S + (size_t)(4 * t) * G::SP + (iy + t / 3) * G::RW + ix + t % 3;
#pragma unroll
      for (int co = 0; co < 4; ++co) h[co] += s[co * G::SP];
    }
    const size_t pix = img0 + (size_t)gy * p.W + gx;
    float z0 = zp[(iy + 2) * G::ZW + ix + 2], z1 = p.z_in[pix * 3 + 1], z2 = p.z_in[pix * 3 + 2];
    z1 = z1 / (fs_sigmoid(h[1] + 2.f) + p.eps) - h[0];
    z2 = z2 / (fs_sigmoid(h[3] + 2.f) + p.eps) - h[2];
    const float* f = p.hF + pix * p.f_pitch + p.f_off;
    const f32x4 f0 = *reinterpret_cast<const f32x4*>(f);
    const f32x2 f1 = *reinterpret_cast<const f32x2*>(f + 4);
    z0 = z0 / (fs_sigmoid(f0[1] + 2.f) + p.eps) - f0[0];
    z1 = z1 / (fs_sigmoid(f0[3] + 2.f) + p.eps) - f0[2];
    z2 = z2 / (fs_sigmoid(f1[1] + 2.f) + p.eps) - f1[0];
    p.z_out[pix * 3] = fmaf(p.M[0], z0, fmaf(p.M[1], z1, fmaf(p.M[2], z2, p.t[0])));
    p.z_out[pix * 3 + 1] = fmaf(p.M[3], z0, fmaf(p.M[4], z1, fmaf(p.M[5], z2, p.t[1])));
    p.z_out[pix * 3 + 2] = fmaf(p.M[6], z0, fmaf(p.M[7], z1, fmaf(p.M[8], z2, p.t[2])));
  }
}

template <typename G>
int fs_launch(FsParams& p, hipStream_t stream) {
  p.tiles_x = cdiv(p.W, G::TW); p.tiles_y = cdiv(p.H, G::TH);
  const long long blocks = (long long)p.B * p.tiles_x * p.tiles_y;
  if (blocks > 0x7fffffffLL) return GLARE_ERR_INVALID;
  constexpr int lds = fs_lds_bytes<G>();
  static_assert(lds <= 160 * 1024, "LDS");
  if (hipFuncSetAttribute((const void*)flow_step_fused_kernel<G>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
    return GLARE_ERR_LAUNCH;
  hipLaunchKernelGGL((flow_step_fused_kernel<G>), dim3((unsigned)blocks), dim3(G::THREADS), lds, stream, p);
  return glare_launch_status();
}

}  // namespace

extern "C" long long glare_flow_step_fused_image_bytes(void) { return FS_IMG_BYTES; }

extern "C" int glare_flow_step_fused_bf16(const float* z_in, float* z_out, const float* ftA, int ftA_pitch, int ftA_off, const void* image,
                                          const float* hF, int hF_pitch, int hF_off, int B, int H, int W, const float* M_3x3_host,
                                          const float* t_3_host, float eps, glare_stream_t stream) {
  if (!z_in || !z_out || z_in == z_out || !ftA || !image || !hF || !M_3x3_host || !t_3_host || B <= 0 || H <= 0 || W <= 0) return GLARE_ERR_INVALID;
  if ((ftA_pitch % 4) || (ftA_off % 4) || ftA_off + 64 > ftA_pitch) return GLARE_ERR_UNSUPPORTED;
  if ((hF_pitch % 4) || (hF_off % 4) || hF_off + 6 > hF_pitch) return GLARE_ERR_UNSUPPORTED;
  if (((uintptr_t)image & 15) || ((uintptr_t)ftA & 15) || ((uintptr_t)hF & 15)) return GLARE_ERR_INVALID;
  FsParams p;
  p.z_in = z_in; p.z_out = z_out; p.ftA = ftA; p.image = (const char*)image; p.hF = hF;
  p.a_pitch = ftA_pitch; p.a_off = ftA_off; p.f_pitch = hF_pitch; p.f_off = hF_off;
  p.B = B; p.H = H; p.W = W;
  for (int i = 0; i < 9; ++i) p.M[i] = M_3x3_host[i];
  for (int i = 0; i < 3; ++i) p.t[i] = t_3_host[i];
  p.eps = eps;
  // 16 x 40 interior pixels per 8-wave workgroup: at the path's latent (105 x 155) that is 7 x 4 tiles per image, 224 workgroups for a batch
  // of 8 -- one round of the 256 CUs, three blocks per wave.  (Measured at 8 x 105 x 155, device time per step: 8 x 32 tiles / 4 waves
  // 36 us -- 560 workgroups of 91 KB LDS = three rounds --, this geometry 19 us; the A fragments kept in registers instead of re-read
  // from LDS per block: no gain, 20.5 us.)
  return fs_launch<FsGeom<16, 40, 8>>(p, (hipStream_t)stream);
}
