// Fused (flash-style) backward of the single-head d = 512 attention: no N x N tensor in HBM.
//
// Replaces the autograd of AttnBlock's two torch.bmm + softmax (reference: encoder_decoder.py:176-188 behind `loss.backward()`,
// LLFlow_model.py:231-236 / VQLLFLOWD_model.py:226-229), which keeps S, P, dP and dS of every block -- 4 x N^2 values per image --
// alive in HBM.  With base-2 logits s_ij = q_i . k_j (scale and log2 e folded into q), L_i = log2 sum_j 2^s_ij saved by the
// forward and D_i = do_i . o_i:
//     p_ij = 2^(s_ij - L_i)    dp_ij = do_i . v_j    ds_ij = ln2 . p_ij (dp_ij - D_i)
//     dv_j = sum_i p_ij do_i   dk_j = sum_i ds_ij q_i   dq_i = sum_j ds_ij k_j
// Three passes of ONE kernel template, each recomputing the score tile from the operands:
//   dV : a workgroup keeps a block of KEYS resident and streams the queries     (scores only: P)
//   dK : the same with the dP product as well                                   (dS)
//   dQ : a workgroup keeps a block of QUERIES resident and streams the keys     (dS; = dK with (Q, dO) and (K, V) exchanged)
// so there is no cross-workgroup accumulation and no atomic: every output element is produced by one workgroup in a fixed order.
// (dK and dV in one pass need 256 accumulators + 128 fragment registers + two score tiles per lane: it spilled.)
//
// Shape on CDNA4 (as the forward: d = 512 fills the register file).  Resident rows r, streamed rows t; per tile of 32 streamed rows
//     X[t][r] = T1[t] . R1[r]      Y[t][r] = T2[t] . R2[r]          (dV, dK: T = (Q, dO), R = (K, V); dQ: T = (K, V), R = (Q, dO))
//     Out^T[d][r] += T1^T[d][t] . dS[t][r]   (dK, dQ)        Out^T[d][r] += T2^T[d][t] . P[t][r]   (dV)
//   * workgroup = 4 waves = 2 resident blocks of 32 rows x 2 halves of d.  A wave keeps the B fragments of its 32 resident rows for
//     its 256 channels in registers (64 VGPRs per operand) and the transposed output of those channels (8 tiles of 32 x 32: 128
//     registers) -- 1 wave per SIMD, as in the forward;
//   * the score tiles contract over ALL 512 channels: each wave of a pair computes the partial over its half, publishes it in LDS
//     and adds its partner's (a + b in one wave, b + a in the other: the same bits);
//   * the streamed tiles T1, T2 (32 rows x 512, 32 KB each) are double-buffered in LDS by LDS-DMA (the whole 160 KB with the score
//     exchange area), shared by the four waves and read twice: as rows (`ds_read_b128`: A operand of X, Y) and transposed
//     (`ds_read_b64_tr_b16`, inline asm with counted lgkmcnt as in the forward: A operand of the output) -- the same image and the
//     same XOR swizzle as the forward's shared K / V tile; rows beyond N are beyond the buffer descriptor's range: zeros;
//   * streamed rows are fetched in the forward's bit-swapped order, so a lane's score registers ARE the B fragment of the output
//     product (k index = streamed row): P and dS never leave the registers;
//   * L and D of the streamed query rows (dV, dK): one row per lane, loaded one tile ahead and handed to the lanes that need them
//     by ds_bpermute after the score products.
#include <type_traits>

#include "common.h"


namespace {

constexpr int HD = 512;
constexpr int TB = 32;                 // streamed rows per tile
constexpr int RB = 64;                 // resident rows per workgroup (2 blocks of 32)
constexpr int TILE_B = TB * HD * 2;    // bytes of one streamed tile (32 KB)

struct BwdParams {
  const a16_t* r1;     // resident operand of the score product      [B][N][512]
  const a16_t* r2;     // resident operand of the dP product (WANT_DS)
  const a16_t* t1;     // streamed operands
  const a16_t* t2;
  const float* lse;     // [B][N] log2-sum-exp of the QUERY rows
  const float* dsum;    // [B][N] do . o of the query rows
  a16_t* out;          // [B][N][512] gradient of the resident rows
  int B, N, nblk, n_blocks;
  float ln2;
};

__device__ __forceinline__ int swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

template <int N>
__device__ __forceinline__ void lgkm_wait() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(N) : "memory"); }
__device__ __forceinline__ void tr_read(u32x2& dst, int addr) { asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(dst) : "v"(addr)); }
__device__ __forceinline__ void pin(u32x2& a) { asm volatile("" : "+v"(a)); }
__device__ __forceinline__ a16x8 frag(const u32x2& lo, const u32x2& hi) {
  return __builtin_bit_cast(a16x8, u32x4{lo[0], lo[1], hi[0], hi[1]});
}

// QS: the QUERIES are the streamed rows (dV, dK); WANT_DS: the output contracts dS with T1 (dK, dQ), else P with T2 (dV)
template <bool QS, bool WANT_DS>
__global__ __launch_bounds__(256, 1) void attn_bwd_kernel(const BwdParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // [2 buffers][T1 | T2][32 rows][64 chunks]: chunk c of row r at r*64 + (c ^ swz(r));  then the score exchange area
  f32x4* xbuf = reinterpret_cast<f32x4*>(smem + 4 * TILE_B);   // [4 waves][X | Y][4][64 lanes]

  int bid = blockIdx.x;
  {  // the workgroups of one image (they stream the same tiles) on one XCD
    const int n = p.n_blocks, q = n / 8, r = n % 8, xcd = bid % 8, kk = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + kk;
  }
  const int b = bid / p.nblk, rblk = bid % p.nblk;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rb = wave >> 1, dh = wave & 1;
  const int n_ = lane & 31, hi = lane >> 5;
  const int rrow = rblk * RB + rb * 32 + n_;
  const bool r_ok = rrow < p.N;
  const size_t img = (size_t)b * p.N;

  // ---- resident B fragments: lane (n, hi) holds row n, channels dh*256 + ks*16 + hi*8 .. +7
  a16x8 r1f[16], r2f[WANT_DS ? 16 : 1];
  {
    const size_t base = (img + (r_ok ? rrow : 0)) * HD + dh * 256 + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      u32x4 a = *reinterpret_cast<const u32x4*>(p.r1 + base + ks * 16);
      if (!r_ok) a = u32x4{0u, 0u, 0u, 0u};
      r1f[ks] = __builtin_bit_cast(a16x8, a);
      if (WANT_DS) {
        u32x4 c = *reinterpret_cast<const u32x4*>(p.r2 + base + ks * 16);
        if (!r_ok) c = u32x4{0u, 0u, 0u, 0u};
        r2f[ks] = __builtin_bit_cast(a16x8, c);
      }
    }
  }
  float Lr = 0.f, Dr = 0.f;   // !QS: the lane's column is a query
  if (!QS && r_ok) { Lr = p.lse[img + rrow]; Dr = p.dsum[img + rrow]; }

  f32x16 o[8];
#pragma unroll
  for (int r = 0; r < 16; ++r)
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i][r] = 0.f;

  // ---- DMA: one instruction = one row (64 lanes x 16 B); wave w moves rows w, w + 4, ... of both tiles; LDS chunk `lane` of row
  // r receives source chunk lane ^ swz(r), and swz(w + 4 i) = (w << 2) | (i & 3)
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<a16_t*>(p.t1 + img * HD), 0, (int)((long long)p.N * HD * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<a16_t*>(p.t2 + img * HD), 0, (int)((long long)p.N * HD * 2), 0x00020000);
  int dvo[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) dvo[i] = (lane ^ ((wave << 2) | i)) * 16;
  auto issue_row = [&](int tile, int buf, int i) {   // row wave + 4 i of both tiles
    char* d1 = smem + buf * 2 * TILE_B;
    const int row = wave + 4 * i;
    const int so = (tile * TB + row) * (HD * 2);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (__attribute__((address_space(3))) void*)(d1 + row * 1024), 16, dvo[i & 3], so, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs2, (__attribute__((address_space(3))) void*)(d1 + TILE_B + row * 1024), 16, dvo[i & 3], so, 0, 0);
  };
  auto issue = [&](int tile, int buf) {
#pragma unroll
    for (int i = 0; i < 8; ++i) issue_row(tile, buf, i);
  };

  // ---- fragment addresses
  // rows of X / Y: streamed row slot i of the MFMA holds row krow(i) (bits 2, 3 swapped): register r of lane (n, hi) is then streamed
  // row 16 (r >> 3) + 8 hi + (r & 7), the k order of the output product's B fragment
  const int krow = (n_ & 0x13) | ((n_ & 4) << 1) | ((n_ & 8) >> 1);
  const int kf = swz(krow);
  // the swizzle only touches the low 4 bits of the chunk index, so 8 addresses cover the 16 k-steps (as the forward's kofs):
  //   A fragment of X / Y, k-step ks: kofs[ks & 7] + (ks >> 3) * 256  (+ buffer and matrix offsets)
  int kofs[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) kofs[j] = (krow * 64 + dh * 32 + ((2 * j + hi) ^ kf)) * 16;
  // transposed reads: lane i of a 16-lane group supplies 4 consecutive channels of streamed row 8 hi + 4 h + (i >> 2) (+ 16 e) and
  // receives [those 4 rows][channel 16 g4 + i] of a 32-channel tile:  tofs[mt & 3][h] + (mt >> 2) * 256 + e * 16 KB
  const int i16 = lane & 15, g4 = (lane >> 4) & 1;
  int tofs[4][2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = 8 * hi + 4 * h + (i16 >> 2);
#pragma unroll
    for (int m4 = 0; m4 < 4; ++m4)
      tofs[m4][h] = row * 1024 + (dh * 32 + ((m4 * 4 + 2 * g4 + ((i16 & 3) >> 1)) ^ swz(row))) * 16 + (i16 & 1) * 8 +
                    (WANT_DS ? 0 : TILE_B);   // the output reads T1 (dS) or T2 (P)
  }

  const int n_tiles = (p.N + TB - 1) / TB;
  // L, D of the streamed query rows (QS): lane l loads those of row tile*32 + (l & 31); the 16 rows a lane's score registers stand
  // for are fetched from the lanes that hold them (ds_bpermute) -- two registers instead of 32
  float Lm = 0.f, Dm = 0.f, Lnx = 0.f, Dnx = 0.f;
  auto load_stats = [&](int tile) {   // one tile ahead, see tile_body
    const int t = min(tile * TB + n_, p.N - 1);   // rows >= N are masked below
    Lnx = p.lse[img + t];
    Dnx = p.dsum[img + t];
  };
  if (QS) load_stats(0);
  issue(0, 0);

  auto tile_body = [&](auto bufc, int tt) {
    constexpr int BUF = decltype(bufc)::value;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // tile tt is in LDS; every wave is done with tile tt - 1 (the other buffer) and with xbuf
    if (QS) {
      // hipcc waits vmcnt(0) wherever a loaded register is consumed while LDS-DMAs are in flight.  So this tile's statistics were
      // loaded one tile ago and are MOVED to fresh registers here, where vmcnt is 0 anyway; the next tile's loads go out in front of
      // the DMA issue and are first touched after the next tile's vmcnt(0).
      asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=&v"(Lm), "=&v"(Dm) : "v"(Lnx), "v"(Dnx));
      if (tt + 1 < n_tiles) load_stats(tt + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    // all 16 pieces here: spreading them over the MFMA groups of the score loop measured the same (1.46 vs 1.44 ms) -- with one
    // wave per SIMD at ~30 % of the MFMA rate the wave's in-order issue stream, not the matrix pipe, is what a piece delays
    if (tt + 1 < n_tiles) issue(tt + 1, BUF ^ 1);

    // ---- partial score tiles over this wave's 256 channels
    f32x16 x, y;
#pragma unroll
    for (int r = 0; r < 16; ++r) { x[r] = 0.f; y[r] = 0.f; }
    // fragment reads one group (2 k-steps) ahead of the MFMAs that consume them
    a16x8 fa[2][2][2];   // [set][k-step of the group][T1 | T2]
    auto ldg = [&](int g, a16x8(&f)[2][2]) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int ks = 2 * g + j;
        const char* src = smem + BUF * 2 * TILE_B + (ks >> 3) * 256 + kofs[ks & 7];
        f[j][0] = *reinterpret_cast<const a16x8*>(src);
        if (WANT_DS) f[j][1] = *reinterpret_cast<const a16x8*>(src + TILE_B);
      }
    };
    ldg(0, fa[0]);
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      if (g + 1 < 8) ldg(g + 1, fa[(g + 1) & 1]);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        x = mfma_a16_32x32x16(fa[g & 1][j][0], r1f[2 * g + j], x, 0, 0, 0);
        if (WANT_DS) y = mfma_a16_32x32x16(fa[g & 1][j][1], r2f[2 * g + j], y, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- publish the partial, add the partner's
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      xbuf[((wave * 2 + 0) * 4 + i) * 64 + lane] = f32x4{x[4 * i], x[4 * i + 1], x[4 * i + 2], x[4 * i + 3]};
      if (WANT_DS) xbuf[((wave * 2 + 1) * 4 + i) * 64 + lane] = f32x4{y[4 * i], y[4 * i + 1], y[4 * i + 2], y[4 * i + 3]};
    }
    // LDS-only barrier: __syncthreads() would also wait for the next tile's DMA (vmcnt(0)), exposing it here
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 xp = xbuf[(((wave ^ 1) * 2 + 0) * 4 + i) * 64 + lane];
#pragma unroll
      for (int e = 0; e < 4; ++e) x[4 * i + e] += xp[e];
      if (WANT_DS) {
        const f32x4 yp = xbuf[(((wave ^ 1) * 2 + 1) * 4 + i) * 64 + lane];
#pragma unroll
        for (int e = 0; e < 4; ++e) y[4 * i + e] += yp[e];
      }
    }
    // ---- P / dS as bf16 B fragments: k-step e covers registers 8 e .. 8 e + 7 = streamed rows 16 e + 8 hi + 0..7
    a16x8 pf[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      u32x4 w;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int r = 8 * e + 2 * j + u, rr = 2 * j + u;                       // rr: row 0..7 inside the lane's 8-row run
          const int t = tt * TB + 16 * e + 8 * hi + rr;
          float Lq = Lr, Dq = Dr;
          if (QS) {
            Lq = __shfl(Lm, 16 * e + 8 * hi + rr, 64);
            Dq = WANT_DS ? __shfl(Dm, 16 * e + 8 * hi + rr, 64) : 0.f;
          }
          const float pe = t < p.N ? __builtin_amdgcn_exp2f(x[r] - Lq) : 0.f;
          v[u] = WANT_DS ? pe * (y[r] - Dq) * p.ln2 : pe;
        }
        w[j] = pack_a2(v[0], v[1]);
      }
      pf[e] = __builtin_bit_cast(a16x8, w);
    }
    // ---- output: Out^T[d tile mt][r] += T^T[d][t] . (dS | P)[t][r]; 4 transposed reads per tile, one tile ahead of its MFMAs
    u32x2 tf[2][4];
    auto rd = [&](int mt, u32x2(&f)[4]) {
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int h = 0; h < 2; ++h)
          tr_read(f[2 * e + h], tofs[mt & 3][h] + (BUF * 2 * TILE_B + e * 16 * 1024 + (mt >> 2) * 256));
    };
    rd(0, tf[0]);
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {   // (two tiles ahead measured the same)
      if (mt + 1 < 8) { rd(mt + 1, tf[(mt + 1) & 1]); lgkm_wait<4>(); } else { lgkm_wait<0>(); }
      u32x2(&f)[4] = tf[mt & 1];
      pin(f[0]); pin(f[1]); pin(f[2]); pin(f[3]);
      o[mt] = mfma_a16_32x32x16(frag(f[0], f[1]), pf[0], o[mt], 0, 0, 0);
      o[mt] = mfma_a16_32x32x16(frag(f[2], f[3]), pf[1], o[mt], 0, 0, 0);
    }
  };

  for (int tt = 0; tt < n_tiles; tt += 2) {
    tile_body(std::integral_constant<int, 0>{}, tt);
    if (tt + 1 < n_tiles) tile_body(std::integral_constant<int, 1>{}, tt + 1);
  }

  // ---- store: a lane owns ONE resident row, 4 consecutive channels per 8-B store
  if (r_ok) {
    a16_t* d1 = p.out + (img + rrow) * HD + dh * 256;
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int d = mt * 32 + 8 * rq + 4 * hi;
        *reinterpret_cast<u32x2*>(d1 + d) = u32x2{pack_a2(o[mt][4 * rq], o[mt][4 * rq + 1]), pack_a2(o[mt][4 * rq + 2], o[mt][4 * rq + 3])};
      }
  }
}

// D[row] = do[row] . o[row] (fp32): one wave per row, 8 channels per lane
__global__ __launch_bounds__(256) void attn_bwd_dsum_kernel(const a16_t* __restrict__ o, const a16_t* __restrict__ dout,
                                                            float* __restrict__ dsum, long long rows) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const u32x4 a = *reinterpret_cast<const u32x4*>(o + row * HD + lane * 8);
  const u32x4 g = *reinterpret_cast<const u32x4*>(dout + row * HD + lane * 8);
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) s += alo(a[e]) * alo(g[e]) + ahi(a[e]) * ahi(g[e]);
  s = wave_sum(s);
  if (lane == 0) dsum[row] = s;
}

constexpr size_t BWD_LDS = (size_t)4 * TILE_B + 4 * 2 * 4 * 64 * 16;   // 128 KB of tiles + 32 KB score exchange = all 160 KB

template <bool QS, bool WANT_DS>
int launch_pass(const BwdParams& p, hipStream_t st) {
  static const hipError_t attr =
      hipFuncSetAttribute((const void*)attn_bwd_kernel<QS, WANT_DS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)BWD_LDS);
  if (attr != hipSuccess) return GLARE_ERR_LAUNCH;
  hipLaunchKernelGGL((attn_bwd_kernel<QS, WANT_DS>), dim3(p.n_blocks), dim3(256), BWD_LDS, st, p);
  return glare_launch_status();
}

}  // namespace

extern "C" size_t glare_attention_d512_backward_workspace_bytes(int B, int N) {
  if (B <= 0 || N <= 0) return 0;
  return (size_t)B * N * sizeof(float);   // D = do . o
}

// q (pre-scaled: q.k are base-2 logits), k, v, o, d_o: bf16 [B][N][512] dense; lse: fp32 [B][N] = log2 sum_j 2^(q_i.k_j) as the
// forward (glare_attention_d512_lse_bf16) leaves it.  Writes dq, dk, dv (bf16 [B][N][512]); ds = ln2_scale * p * (dp - D).
extern "C" int glare_attention_d512_backward_bf16(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                                                  const float* lse, void* dq, void* dk, void* dv, int B, int N, float ln2_scale,
                                                  void* workspace, size_t workspace_bytes, glare_stream_t stream) {
  if (!q || !k || !v || !o || !d_o || !lse || !dq || !dk || !dv || B <= 0 || N <= 0) return GLARE_ERR_INVALID;
  if (!workspace || workspace_bytes < glare_attention_d512_backward_workspace_bytes(B, N)) return GLARE_ERR_WORKSPACE;
  if ((long long)B * ((N + RB - 1) / RB) > 0x7fffffffLL || (long long)N * HD * 2 >= (1ll << 31)) return GLARE_ERR_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* dsum = static_cast<float*>(workspace);
  const long long rows = (long long)B * N;
  hipLaunchKernelGGL(attn_bwd_dsum_kernel, dim3((unsigned)cdivll(rows, 4)), dim3(256), 0, st, static_cast<const a16_t*>(o),
                     static_cast<const a16_t*>(d_o), dsum, rows);
  BwdParams p;
  p.B = B; p.N = N; p.nblk = (N + RB - 1) / RB; p.n_blocks = B * p.nblk; p.ln2 = ln2_scale;
  p.lse = lse; p.dsum = dsum;
  // dV, dK: keys resident, queries streamed
  p.r1 = static_cast<const a16_t*>(k); p.r2 = static_cast<const a16_t*>(v);
  p.t1 = static_cast<const a16_t*>(q); p.t2 = static_cast<const a16_t*>(d_o);
  p.out = static_cast<a16_t*>(dv);
  int rc = launch_pass<true, false>(p, st);
  if (rc != GLARE_OK) return rc;
  p.out = static_cast<a16_t*>(dk);
  rc = launch_pass<true, true>(p, st);
  if (rc != GLARE_OK) return rc;
  // dQ: queries resident, keys streamed
  p.r1 = static_cast<const a16_t*>(q); p.r2 = static_cast<const a16_t*>(d_o);
  p.t1 = static_cast<const a16_t*>(k); p.t2 = static_cast<const a16_t*>(v);
  p.out = static_cast<a16_t*>(dq);
  return launch_pass<false, true>(p, st);
}
