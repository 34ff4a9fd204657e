// Codebook retrieval: nearest-neighbour search of every latent token in the learned
// codebook, fused with the gather of the selected entries.
//
// Replaces VectorQuantizer2.forward's distance matrix + argmin + embedding lookup
// (reference: code/models/modules/quantize.py:276-285), which materialises an
// [N x 8192] fp32 matrix (533 MB per 400x600 image).  Here the codebook (8192 x 3 fp32 =
// 96 KB) plus its squared norms lives in one CU's LDS (128 KB of the 160 KB), every lane
// owns TPT tokens in registers and walks the codebook through broadcast ds_read_b128.
// HBM traffic is the algorithmic minimum: 12 B in + 12 B z_q + 8 B index per token.
//
// Bit-exactness contract (SURVEY.md section 7 "hard parts" item 2, oracle/vq_ref.c):
//   zz = (z0*z0 + z1*z1) + z2*z2          (no fma)
//   ee = (e0*e0 + e1*e1) + e2*e2          (no fma)
//   dot = fma(z2, e2, fma(z1, e1, z0*e0))
//   d   = (zz + ee) - 2*dot               first minimum wins
// This file is compiled with -ffp-contract=off and spells every rounding explicitly.
#include "common.h"

namespace {

constexpr int VQ_THREADS = 1024;   // 16 waves: 4 per SIMD hide the broadcast-read latency of the scan (one wave per SIMD ran 100
                                   // cycles per code for ~40 cycles of arithmetic)
constexpr int VQ_TPT = 2;          // tokens per lane
constexpr int VQ_CHUNK = 8192;     // codes resident in LDS at once (x 16 B = 128 KB)

// A workgroup = `slots` token lanes (a multiple of 64, TPT tokens each) x `parts` = VQ_THREADS / slots code ranges: thread
// (part, slot) scans the part-th share of every resident chunk for its tokens, first minimum wins inside its (ascending) sequence,
// and the parts are merged by (distance, index) -- the smaller index wins a tie, which is the sequential scan's "first minimum
// wins" whatever the order of the merge.  Few tokens (the training crops: 4 096 - 12 800) take few slots and many parts, so the
// launch still fills the chip: the scan of one workgroup was 394 us whether it served 130 200 tokens or 4 096.
template <int TPT>
__global__ __launch_bounds__(VQ_THREADS, 1) void vq_nearest_kernel(
    const float* __restrict__ z, const float* __restrict__ codebook, long long n_tokens,
    int n_codes, long long* __restrict__ idx_out, float* __restrict__ zq_out, int slots) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* cb = reinterpret_cast<f32x4*>(smem);

  const int parts = VQ_THREADS / slots;
  const int part = threadIdx.x / slots, slot = threadIdx.x % slots;   // part is wave-uniform (slots % 64 == 0)
  const long long t0 = ((long long)blockIdx.x * slots + slot) * TPT;
  float z0[TPT], z1[TPT], z2[TPT], zz[TPT], best[TPT];
  int bidx[TPT];
#pragma unroll
  for (int t = 0; t < TPT; ++t) {
    const long long tok = t0 + t;
    const bool ok = tok < n_tokens;
    z0[t] = ok ? z[tok * 3 + 0] : 0.f;
    z1[t] = ok ? z[tok * 3 + 1] : 0.f;
    z2[t] = ok ? z[tok * 3 + 2] : 0.f;
    zz[t] = __fadd_rn(__fadd_rn(__fmul_rn(z0[t], z0[t]), __fmul_rn(z1[t], z1[t])),
                      __fmul_rn(z2[t], z2[t]));
    best[t] = __builtin_inff();
    bidx[t] = 0x7fffffff;
  }

  for (int c0 = 0; c0 < n_codes; c0 += VQ_CHUNK) {
    const int nc = min(VQ_CHUNK, n_codes - c0);
    __syncthreads();
    for (int c = threadIdx.x; c < nc; c += VQ_THREADS) {
      const float e0 = codebook[(long long)(c0 + c) * 3 + 0];
      const float e1 = codebook[(long long)(c0 + c) * 3 + 1];
      const float e2 = codebook[(long long)(c0 + c) * 3 + 2];
      const float ee = __fadd_rn(__fadd_rn(__fmul_rn(e0, e0), __fmul_rn(e1, e1)), __fmul_rn(e2, e2));
      f32x4 v = {e0, e1, e2, ee};
      cb[c] = v;
    }
    __syncthreads();
    const int cb0 = (int)((long long)nc * part / parts), cb1 = (int)((long long)nc * (part + 1) / parts);
#pragma unroll 4
    for (int c = cb0; c < cb1; ++c) {
      const f32x4 e = cb[c];  // wave-uniform address: one broadcast ds_read_b128
#pragma unroll
      for (int t = 0; t < TPT; ++t) {
        const float dot = __fmaf_rn(z2[t], e[2], __fmaf_rn(z1[t], e[1], __fmul_rn(z0[t], e[0])));
        const float d = __fsub_rn(__fadd_rn(zz[t], e[3]), __fmul_rn(2.0f, dot));
        if (d < best[t]) {
          best[t] = d;
          bidx[t] = c0 + c;
        }
      }
    }
  }

  if (parts > 1) {   // merge the parts' candidates: smaller distance, then smaller index
    __syncthreads();                                   // every scan is done with the codebook image: its LDS becomes the exchange
    float* xb = reinterpret_cast<float*>(smem);        // [part][t][slot] distances, then the indices
    int* xi = reinterpret_cast<int*>(smem) + VQ_THREADS * TPT;
#pragma unroll
    for (int t = 0; t < TPT; ++t) {
      xb[(part * TPT + t) * slots + slot] = best[t];
      xi[(part * TPT + t) * slots + slot] = bidx[t];
    }
    __syncthreads();
    if (part != 0) return;
#pragma unroll
    for (int t = 0; t < TPT; ++t)
      for (int k = 1; k < parts; ++k) {
        const float d = xb[(k * TPT + t) * slots + slot];
        const int i = xi[(k * TPT + t) * slots + slot];
        if (d < best[t] || (d == best[t] && i < bidx[t])) { best[t] = d; bidx[t] = i; }
      }
  }

#pragma unroll
  for (int t = 0; t < TPT; ++t) {
    const long long tok = t0 + t;
    if (tok < n_tokens) {
      const int bi = bidx[t] == 0x7fffffff ? 0 : bidx[t];   // no distance compared below +inf (all NaN): index 0, as the one-range scan gave
      idx_out[tok] = (long long)bi;
      if (zq_out) {
        const float* e = codebook + (long long)bi * 3;
        zq_out[tok * 3 + 0] = e[0];
        zq_out[tok * 3 + 1] = e[1];
        zq_out[tok * 3 + 2] = e[2];
      }
    }
  }
}

}  // namespace

extern "C" int glare_vq_nearest_f32(const float* z_nhwc, const float* codebook, long long n_tokens,
                                    int n_codes, int dim, long long* idx_i64, float* zq_nhwc,
                                    glare_stream_t stream) {
  if (n_tokens < 0 || n_codes <= 0) return GLARE_ERR_INVALID;
  if (dim != 3) return GLARE_ERR_UNSUPPORTED;  // the GLARE codebook is 8192 x 3 (VQModel_arch.py:44)
  if (n_tokens == 0) return GLARE_OK;          // empty batch: nothing to do, pointers may be NULL
  if (!z_nhwc || !codebook || !idx_i64) return GLARE_ERR_INVALID;
  const size_t lds = (size_t)VQ_CHUNK * sizeof(f32x4);
  // token lanes per workgroup: as many as fill the chip's 256 CUs once, between one wave and a quarter of the workgroup (so that
  // at least 4 code ranges keep 4 waves on every SIMD)
  long long lanes = (n_tokens + VQ_TPT - 1) / VQ_TPT;
  int slots = (int)(((lanes + 255) / 256 + 63) / 64 * 64);
  if (slots < 64) slots = 64;
  if (slots > 128) slots = VQ_THREADS / 4;   // divisors of VQ_THREADS only (64 / 128 / 256): with 192 the last wave had part == parts and
                                             // scanned / exchanged past the resident chunk (its results were dropped, but it read and
                                             // wrote LDS beyond the allocation)
  // per call, not cached: the attribute is per device and the library keeps no state
  if (hipFuncSetAttribute((const void*)vq_nearest_kernel<VQ_TPT>,
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return GLARE_ERR_LAUNCH;
  const long long per_block = (long long)slots * VQ_TPT;
  const long long blocks = (n_tokens + per_block - 1) / per_block;
  if (blocks > 0x7fffffffLL) return GLARE_ERR_INVALID;
  hipLaunchKernelGGL(vq_nearest_kernel<VQ_TPT>, dim3((unsigned)blocks), dim3(VQ_THREADS), lds,
                     (hipStream_t)stream, z_nhwc, codebook, n_tokens, n_codes, idx_i64, zq_nhwc, slots);
  return glare_launch_status();
}
