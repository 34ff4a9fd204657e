// Batched NT GEMM on bf16 MFMA:  C[b][m][n] = alpha * sum_k A[b][m][k] * B[b][n][k]  (+ C).
//
// The training step (LLFlow_model.py:181-250 `loss.backward()`, VQLLFLOWD_model.py:187-232) needs, on top of the
// forward kernels, contractions whose K dimension is the PIXEL axis (weight gradients: cuDNN wgrad in the
// reference) or the token axis (attention backward: the autograd of the two torch.bmm in
// encoder_decoder.py:176-188).  Both are brought to this one shape by the transposing producers in
// train_ops.hip (K-contiguous operands), so that both MFMA operands are dense 16-B fragment reads.
//
// Tile: 256 (m) x 128 (n) per 4-wave workgroup, wave tile 128 x 64 (4 x 2 MFMA 32x32x16 tiles: 6 fragment
// reads per 8 MFMAs), K stage = 32 (64 B per row).  Both operand tiles are double-buffered in LDS and
// filled by LDS-DMA through a buffer descriptor whose range check zero-fills rows beyond M / N; the LDS
// image is row-major with the 16-B chunk index XOR-ed by (row>>2)&3 -- applied on the SOURCE side, the
// DMA destination being lane-linear -- which makes every ds_read_b128 fragment read conflict-free.
// Split-K is the batch dimension with strideA/strideB = the K slice and a partial C per slice.
#include "common.h"

namespace {

constexpr int BM = 256, BN = 128, BK = 32;

struct GemmParams {
  const a16_t* A;
  const a16_t* B;
  void* C;
  int M, N, K;
  long long lda, ldb, ldc, sA, sB, sC;
  int tiles_m, tiles_n;
  float alpha;
  int out_bf16, accumulate;
};

__global__ __launch_bounds__(256, 3) void gemm_nt_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u32x4* lA = reinterpret_cast<u32x4*>(smem);          // [2][BM*4] 16-B chunks
  u32x4* lB = lA + 2 * BM * 4;                         // [2][BN*4]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int tm = blockIdx.x % p.tiles_m, tn = blockIdx.x / p.tiles_m;  // m fastest: neighbours share the B tile in L2
  const int b = blockIdx.y;
  const int row0 = tm * BM, col0 = tn * BN;
  const a16_t* Ab = p.A + (long long)b * p.sA + (long long)row0 * p.lda;
  const a16_t* Bb = p.B + (long long)b * p.sB + (long long)col0 * p.ldb;
  const int rowsA = min(p.M - row0, BM), rowsB = min(p.N - col0, BN);
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<a16_t*>(Ab), 0, (int)((((long long)rowsA - 1) * p.lda + p.K) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<a16_t*>(Bb), 0, (int)((((long long)rowsB - 1) * p.ldb + p.K) * 2), 0x00020000);
  // DMA lane geometry: one instruction = 16 rows x 64 B; lane -> (row r, LDS chunk c'), source chunk c' ^ s(r)
  const int dr = lane >> 2, dc = (lane & 3) ^ ((dr >> 2) & 3);
  const int voffA = (int)((dr * p.lda + dc * 8) * 2), voffB = (int)((dr * p.ldb + dc * 8) * 2);
  const int strideA16 = (int)(16 * p.lda * 2), strideB16 = (int)(16 * p.ldb * 2);
  auto issue = [&](int kt, int buf) {
    const int kofs = kt * BK * 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // 16 A instructions, 4 per wave
      const int j = wave * 4 + i;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (__attribute__((address_space(3))) void*)(lA + buf * (BM * 4) + j * 64), 16,
                                               voffA, j * strideA16 + kofs, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {  // 8 B instructions, 2 per wave
      const int j = wave * 2 + i;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (__attribute__((address_space(3))) void*)(lB + buf * (BN * 4) + j * 64), 16,
                                               voffB, j * strideB16 + kofs, 0, 0);
    }
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int l = lane & 31, khalf = lane >> 5, sw = (l >> 2) & 3;
  const int KT = p.K / BK;
  issue(0, 0);
  for (int kt = 0; kt < KT; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < KT) issue(kt + 1, (kt + 1) & 1);
    const u32x4* cA = lA + (kt & 1) * (BM * 4) + (wm * 128 + l) * 4;
    const u32x4* cB = lB + (kt & 1) * (BN * 4) + (wn * 64 + l) * 4;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int c = (ks * 2 + khalf) ^ sw;
      a16x8 af[4], bfr[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) bfr[j] = __builtin_bit_cast(a16x8, cB[j * 128 + c]);
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = __builtin_bit_cast(a16x8, cA[i * 128 + c]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma_a16_32x32x16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  }

  char* Cb = reinterpret_cast<char*>(p.C);
  const long long cbase = (long long)b * p.sC;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = col0 + wn * 64 + j * 32 + l;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = row0 + wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
        if (m < p.M && n < p.N) {
          const long long idx = cbase + (long long)m * p.ldc + n;
          float v = p.alpha * acc[i][j][r];
          if (p.out_bf16) {
            a16_t* o = reinterpret_cast<a16_t*>(Cb) + idx;
            if (p.accumulate) v += a2f(*o);
            *o = f2a(v);
          } else {
            float* o = reinterpret_cast<float*>(Cb) + idx;
            if (p.accumulate) v += *o;
            *o = v;
          }
        }
      }
    }
}

// out[i] (+)= sum_s parts[s][i]; the deterministic second level of every split-K contraction
__global__ __launch_bounds__(256) void reduce_parts_kernel(const float* __restrict__ parts, int S, long long n, float scale,
                                                            float* __restrict__ out, int accumulate) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  parts += (long long)blockIdx.y * S * n;   // blockIdx.y = group: out[grp][i] = sum_s parts[grp][s][i]
  out += (long long)blockIdx.y * n;
  float s = 0.f;
  int k = 0;
  for (; k + 7 < S; k += 8) {   // eight loads in flight, added in the original order (same bits as the one-by-one loop)
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = parts[(long long)(k + u) * n + i];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; k < S; ++k) s += parts[(long long)k * n + i];
  s *= scale;
  out[i] = accumulate ? out[i] + s : s;
}

// the same sum for MANY parts of FEW elements (per-workgroup partials of the flow / GroupNorm reductions: S = 256, n = 12..576):
// one thread per element would walk S dependent loads.  Thread (e, j) adds the parts s = j (mod G) of element e in ascending
// order, the G subtotals are added in order j = 0..G-1 -- a fixed tree for a given (S, n).
template <int G>
__global__ __launch_bounds__(256) void reduce_many_parts_kernel(const float* __restrict__ parts, int S, long long n, float scale,
                                                                    float* __restrict__ out, int accumulate) {
  constexpr int EL = 256 / G;
  __shared__ float sub[G][EL];
  const int e = threadIdx.x % EL, j = threadIdx.x / EL;
  const long long i = (long long)blockIdx.x * EL + e;
  parts += (long long)blockIdx.y * S * n;
  out += (long long)blockIdx.y * n;
  float s = 0.f;
  if (i < n)
    for (int k = j; k < S; k += G) s += parts[(long long)k * n + i];
  sub[j][e] = s;
  __syncthreads();
  if (j == 0 && i < n) {
    for (int k = 1; k < G; ++k) s += sub[k][e];
    s *= scale;
    out[i] = accumulate ? out[i] + s : s;
  }
}

}  // namespace

static int gemm_launch(const void* A, const void* B, void* C, int M, int N, int K, long long lda, long long ldb, long long ldc,
                       int batch, long long strideA, long long strideB, long long strideC, float alpha, int out_bf16, int accumulate,
                       glare_stream_t stream);

extern "C" int glare_gemm_nt_bf16(const void* A, const void* B, void* C, int M, int N, int K, long long lda, long long ldb,
                                  long long ldc, int batch, long long strideA, long long strideB, long long strideC, float alpha,
                                  int out_bf16, int accumulate, glare_stream_t stream) {
  return gemm_launch(A, B, C, M, N, K, lda, ldb, ldc, batch, strideA, strideB, strideC, alpha, out_bf16, accumulate, stream);
}

static int gemm_launch(const void* A, const void* B, void* C, int M, int N, int K, long long lda, long long ldb, long long ldc,
                       int batch, long long strideA, long long strideB, long long strideC, float alpha, int out_bf16, int accumulate,
                       glare_stream_t stream) {
  if (M < 0 || N < 0 || K < 0 || batch < 0) return GLARE_ERR_INVALID;
  if (M == 0 || N == 0 || batch == 0) return GLARE_OK;
  if (!A || !B || !C) return GLARE_ERR_INVALID;
  if (K == 0 || K % BK != 0 || lda % 8 != 0 || ldb % 8 != 0 || lda < K || ldb < K || ldc < N) return GLARE_ERR_INVALID;
  if (strideA % 8 != 0 || strideB % 8 != 0) return GLARE_ERR_INVALID;
  if (((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) != 0) return GLARE_ERR_INVALID;
  if ((BM * lda + K) * 2 >= (1ll << 31) || (BN * ldb + K) * 2 >= (1ll << 31)) return GLARE_ERR_UNSUPPORTED;  // 32-bit DMA offsets
  if (batch > 65535) return GLARE_ERR_UNSUPPORTED;
  GemmParams p;
  p.A = static_cast<const a16_t*>(A);
  p.B = static_cast<const a16_t*>(B);
  p.C = C;
  p.M = M; p.N = N; p.K = K;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.sA = strideA; p.sB = strideB; p.sC = strideC;
  p.tiles_m = cdiv(M, BM); p.tiles_n = cdiv(N, BN);
  p.alpha = alpha; p.out_bf16 = out_bf16; p.accumulate = accumulate;
  const size_t lds = (size_t)(2 * BM * 4 + 2 * BN * 4) * 16;
  hipLaunchKernelGGL(gemm_nt_kernel, dim3(p.tiles_m * p.tiles_n, batch), dim3(256), lds, static_cast<hipStream_t>(stream), p);
  return glare_launch_status();
}

extern "C" int glare_reduce_parts_grouped_f32(const float* parts, int n_groups, int n_parts, long long n, float scale, float* out,
                                              int accumulate, glare_stream_t stream) {
  if (n_groups < 0 || n_parts < 0 || n < 0 || n_groups > 65535) return GLARE_ERR_INVALID;
  if (n == 0 || n_groups == 0) return GLARE_OK;
  if (!parts || !out) return GLARE_ERR_INVALID;
  if (n_parts >= 32 && n <= 65536)   // many parts of few elements: split the walk over the parts as well
    hipLaunchKernelGGL(reduce_many_parts_kernel<16>, dim3((unsigned)cdivll(n, 16), n_groups), dim3(256), 0, static_cast<hipStream_t>(stream),
                       parts, n_parts, n, scale, out, accumulate);
  else
    hipLaunchKernelGGL(reduce_parts_kernel, dim3((unsigned)cdivll(n, 256), n_groups), dim3(256), 0, static_cast<hipStream_t>(stream), parts,
                       n_parts, n, scale, out, accumulate);
  return glare_launch_status();
}

extern "C" int glare_reduce_parts_f32(const float* parts, int n_parts, long long n, float scale, float* out, int accumulate,
                                      glare_stream_t stream) {
  return glare_reduce_parts_grouped_f32(parts, 1, n_parts, n, scale, out, accumulate, stream);
}
