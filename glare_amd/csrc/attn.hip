// Blockwise (flash-style) single-head spatial self-attention, head dim 512, bf16 MFMA.
//
// Replaces AttnBlock.forward's two torch.bmm + N x N softmax (reference:
// encoder_decoder.py:176-188), which materialises a [B, N, N] score tensor (N = 16 275 tokens for a
// 400x600 image: 1.06 GB fp32 per block per image, 11 blocks).  Here no N^2 tensor exists:
//   out[q, :] = sum_j softmax_j(q.k_j) v_j        (scale and log2(e) are folded into q by the caller)
//
// Shape of the problem on CDNA4: d = 512 means the output accumulator of 32 query rows is
// 32 x 512 fp32 = 256 registers per lane -- the whole AGPR half of the unified file.  So:
//   * one workgroup = 4 waves = 128 query rows, one wave per SIMD, 32 rows per wave, 512 regs;
//   * Q (32 x 512 bf16 = 128 VGPRs) stays in registers for the whole kernel;
//   * K and V^T tiles of 32 keys stream through LDS (2 x 2 x 32 KB, double-buffered) by LDS-DMA,
//     shared by the 4 waves; V arrives pre-transposed ([d][token], written that way by the
//     projection conv's epilogue) so that BOTH MFMA A-operands are contraction-contiguous;
//   * transposed products keep the softmax lane-local:
//       S^T[kv, q] = K . Q^T   (A = K tile rows, B = Q^T from registers)  -> lane owns column q
//       O^T[d,  q] += V^T . P^T (A = V^T tile rows, B = P^T = the lane's own S^T registers)
//     every accumulator of a lane belongs to ONE query row: max/sum/rescale need no LDS and one
//     cross-half shuffle; K rows are fetched in a bit-swapped order so that the lane's S^T
//     registers are already the P^T B-fragment (no permlane, no LDS round trip for P);
//   * LDS images are XOR-swizzled at 16-B granularity (applied on the DMA source address,
//     the LDS destination stays lane-linear) so both fragment reads are conflict-free;
//   * online softmax with deferred rescale (rescale the 256 accumulators only when the running
//     max grows by more than 2^8), textbook order: decide -> rescale -> exponentiate -> P.V.
//   * workgroups are numbered so that one XCD works on one image: its 32 CUs stream the same
//     K/V through one L2.
#include <type_traits>

#include "common.h"

namespace {

constexpr int AT_THREADS = 256;
constexpr int HD = 512;          // head dim
constexpr int BM = 128;          // query rows per workgroup
constexpr int BN = 32;           // keys per tile
constexpr int KCH = BN * (HD / 8);   // 16-B chunks per K tile  (2048)
constexpr float RESCALE_THR = 8.0f;  // log2 units
// Variants of this kernel measured in rounds 1-2 and not kept (their code left the tree in round 5; figures in DESIGN.md section 3): a
// software-pipelined tile loop (S(j+1) on the matrix pipe while the vector ALU runs softmax(j), the last quarter of Q in LDS): +1 %
// (1036 vs 1023 TFLOP/s; 1206 vs 1303 with the DMA compiled out) -- the serial softmax is NOT what limits the kernel, the 64 LDS-DMA
// pieces per tile are (~25 %); an 8-wave form (16 query rows per wave on v_mfma_f32_16x16x32, two waves per SIMD): 975-995 vs 1 030-1 045
// TFLOP/s -- two waves per SIMD do hide the DMA issue cost (8 % instead of 25 %), but every 1-KB fragment then feeds a 16-row MFMA and
// twice the LDS read traffic per flop (512 KB per 32-key tile per CU) becomes the limit.
constexpr int ATTN_DMA_PER_GROUP = 2;  // DMA pieces issued per 4-MFMA group (2: all 16 during QK^T; 1, over the whole tile, measured slower)

__device__ __forceinline__ void dma16a(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

struct AttnParams {
  const a16_t* q;
  const a16_t* k;
  const a16_t* vt;
  a16_t* o;
  int B, N;
  long long Npad;
  int ldq, ldk, ldo;
  int n_qblocks, n_blocks;
  int key_splits;      // > 1: each workgroup covers 1/key_splits of the keys and leaves an un-normalised partial
  float* part_o;       // [B][key_splits][N][512] fp32
  float* part_ml;      // [B][key_splits][N][2]   (running max in log2 units, sum)
  float* lse;          // optional [B][N]: log2 sum_j 2^(q_i.k_j) of every query row (what the backward needs); attn_fwd_kernel only
  a16_t* o_lo;         // optional remainder half of the output (value = o + o_lo, same ldo): attn_kv_fwd_kernel / attn_combine_kernel
};

__global__ __launch_bounds__(AT_THREADS, 1) void attn_fwd_kernel(const AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u32x4* lK = reinterpret_cast<u32x4*>(smem);   // [2][KCH]
  u32x4* lV = lK + 2 * KCH;                     // [2][KCH]

  int bid = blockIdx.x;
  {
    const int n = p.n_blocks, q = n / 8, r = n % 8, xcd = bid % 8, kk = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + kk;
  }
  // block order: (image, key split, query block) -- neighbours stream the same K/V range through one L2
  const int qb = bid % p.n_qblocks, ksplit = (bid / p.n_qblocks) % p.key_splits, b = bid / (p.n_qblocks * p.key_splits);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ql = lane & 31, hi = lane >> 5;
  const int qrow = qb * BM + wave * 32 + ql;
  const bool q_ok = qrow < p.N;

  // Q^T B-fragments: lane (q, hi) holds Q[q][16*ks + 8*hi .. +8]
  a16x8 qf[HD / 16];
  {
    const a16_t* qp = p.q + ((size_t)b * p.N + (q_ok ? qrow : 0)) * p.ldq + hi * 8;
#pragma unroll
    for (int ks = 0; ks < HD / 16; ++ks) {
      u32x4 v = *reinterpret_cast<const u32x4*>(qp + ks * 16);
      if (!q_ok) v = u32x4{0u, 0u, 0u, 0u};
      qf[ks] = __builtin_bit_cast(a16x8, v);
    }
  }

  f32x16 o[HD / 32];
#pragma unroll
  for (int i = 0; i < HD / 32; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;

  const int n_tiles = (p.N + BN - 1) / BN;
  // this workgroup's key tiles (all of them unless the keys are split across workgroups to fill the chip at small batch)
  const int t_begin = (int)((long long)n_tiles * ksplit / p.key_splits), t_end = (int)((long long)n_tiles * (ksplit + 1) / p.key_splits);
  const a16_t* kbase = p.k + (size_t)b * p.N * p.ldk;
  const a16_t* vbase = p.vt + (size_t)b * HD * p.Npad;

  // DMA source addressing: a wave-uniform 64-bit base (SGPRs: image, tile, row) plus ONE 32-bit per-lane
  // offset.  Per-piece 64-bit pointers kept in VGPRs get spilled around the tile loop, and every
  // scratch reload drags a vmcnt(0) that drains the DMAs in flight (measured 2x slowdown).
  //   K:   one instruction per key row (1 KB); chunk c of row r lands at r*64 + (c ^ (r & 15))
  //   V^T: one instruction per 16 d-rows (64 B each); chunk c of row d lands at d*4 + (c ^ ((d>>2)&3));
  //        d = 16*t + (lane>>2)  =>  (d>>2)&3 == (lane>>4)&3 for every t
  const unsigned v_lane_off = (unsigned)((lane >> 2) * p.Npad + (((lane & 3) ^ ((lane >> 4) & 3)) * 8)) * 2u;  // bytes
  // Both tiles are fetched through per-image buffer descriptors: the per-piece address is a scalar offset (row / d-group and
  // tile) plus a 32-bit per-lane offset -- no 64-bit vector add per piece; key rows beyond N fall outside the descriptor's
  // range and arrive as zeros (their scores are masked to -inf below), so no row clamp either.
  const __amdgpu_buffer_rsrc_t krsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<a16_t*>(kbase), 0, (int)((((long long)p.N - 1) * p.ldk + HD) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t vrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<a16_t*>(vbase), 0, (int)((long long)HD * p.Npad * 2), 0x00020000);
  const int lane16 = lane * 16;
  auto issue_piece = [&](auto ic, int tile, int buf) {
    constexpr int i = decltype(ic)::value;
    if constexpr (i < BN / 4) {
      const int r = wave + 4 * i;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(krsrc, (__attribute__((address_space(3))) void*)(lK + buf * KCH + r * 64), 16,
                                               lane16 ^ ((r & 15) * 16), (tile * BN + r) * p.ldk * 2, 0, 0);
    } else {
      constexpr int j = i - BN / 4;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(vrsrc, (__attribute__((address_space(3))) void*)(lV + buf * KCH + (wave + 4 * j) * 64), 16,
                                               v_lane_off, (int)(((long long)(wave + 4 * j) * 16 * p.Npad + (long long)tile * BN) * 2), 0, 0);
    }
  };
  auto issue = [&](int tile, int buf) { static_for<16>([&](auto ic) { issue_piece(ic, tile, buf); }); };

  // K row fetched for MFMA row slot i: bits 2 and 3 swapped, so that output register r of lane
  // (q, hi) is key 16*(r>>3) + 8*hi + (r&7) -- exactly the P^T B-fragment order.
  const int krow = (ql & 0x13) | ((ql & 4) << 1) | ((ql & 8) >> 1);
  const int kswz = krow & 15;

  // per-lane LDS byte addresses of the fragment reads; everything else is an instruction immediate:
  //   K frag (ks)     : kofs[ks & 7] + (ks >> 3) * 256 + buf * 32 KB            (kofs holds the XOR swizzle)
  //   V^T frag (dt,ks): vofs[ks] + dt * 2048 + buf * 32 KB                      (vofs includes the 64 KB V base)
  int kofs[8], vofs[2];
#pragma unroll
  for (int bb = 0; bb < 8; ++bb) kofs[bb] = (krow * 64 + ((2 * bb + hi) ^ kswz)) * 16;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) vofs[ks] = 2 * KCH * 16 + (ql * 4 + ((2 * ks + hi) ^ ((ql >> 2) & 3))) * 16;

  // One key tile.  BUF is a compile-time constant so that the buffer offset folds into the ds_read
  // immediate.  Fragment reads run two groups (8 x ds_read_b128) ahead of the MFMAs that consume them
  // through a 3-deep rotating register set; the first V^T groups are fetched before the softmax.
  auto tile_body = [&](auto bufc, int tile) {
    constexpr int BUF = decltype(bufc)::value;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // The next tile's 16 DMA pieces are NOT issued here in one burst (all four waves would queue on the
    // texture-address path with the matrix pipe idle: measured -27 %); they are spread, ATTN_DMA_PER_GROUP per
    // MFMA group, so the address path works under the MFMAs.
    const int nxt = min(tile + 1, t_end - 1);  // last tile: a redundant reload keeps the loop branch-free
    const char* kb = smem + BUF * KCH * 16;
    const char* vb = smem + BUF * KCH * 16;
    a16x8 fr[3][4];
    auto ldk = [&](auto gc, a16x8(&f)[4]) {
      constexpr int g = decltype(gc)::value;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int ks = 4 * g + e;
        f[e] = *reinterpret_cast<const a16x8*>(kb + kofs[ks & 7] + (ks >> 3) * 256);
      }
    };
    auto ldv = [&](auto gc, a16x8(&f)[4]) {
      constexpr int g = decltype(gc)::value;
#pragma unroll
      for (int e = 0; e < 4; ++e) f[e] = *reinterpret_cast<const a16x8*>(vb + vofs[e & 1] + (2 * g + (e >> 1)) * 2048);
    };

    // ---- S^T = K . Q^T  (32 keys x 32 queries, contraction over d = 512)
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    ldk(std::integral_constant<int, 0>{}, fr[0]);
    ldk(std::integral_constant<int, 1>{}, fr[1]);
    static_for<8>([&](auto gc) {
      constexpr int g = decltype(gc)::value;
      if constexpr (g + 2 < 8) ldk(std::integral_constant<int, g + 2>{}, fr[(g + 2) % 3]);
      else ldv(std::integral_constant<int, g + 2 - 8>{}, fr[(g + 2) % 3]);
      static_for<ATTN_DMA_PER_GROUP>([&](auto kc) {
        constexpr int piece = g * ATTN_DMA_PER_GROUP + decltype(kc)::value;
        if constexpr (piece < 16) issue_piece(std::integral_constant<int, piece>{}, nxt, BUF ^ 1);
      });
#pragma unroll
      for (int e = 0; e < 4; ++e) s = mfma_a16_32x32x16(fr[g % 3][e], qf[4 * g + e], s, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    });
    if (tile == n_tiles - 1) {  // mask keys beyond N
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kv = tile * BN + 16 * (r >> 3) + 8 * hi + (r & 7);
        if (kv >= p.N) s[r] = -__builtin_inff();
      }
    }
    // ---- online softmax, lane-local per query column
    float mx = s[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
    {  // the other 16 keys of this query live in lane ^ 32: v_permlane32_swap, no LDS round trip
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
      mx = fmaxf(mx, __uint_as_float(hi ? sw[0] : sw[1]));
    }
    if (__any(mx > m_run + RESCALE_THR)) {  // wave-uniform: rescale everything still at the old max
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
#pragma unroll
      for (int i = 0; i < HD / 32; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          // in place on the accumulator register: a plain `o *= alpha` under this (wave-uniform)
          // branch makes the register allocator copy all 256 accumulators and spill
          float x = o[i][r], tmp;
          asm volatile("v_accvgpr_read_b32 %1, %0\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 %0, %1"
                       : "+a"(x), "=&v"(tmp)
                       : "v"(alpha));
          o[i][r] = x;
        }
      m_run = m_new;
    }
    float psum = 0.f;
    a16x8 pf[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      u32x4 w;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float p0 = __builtin_amdgcn_exp2f(s[8 * h + 2 * e] - m_run);
        const float p1 = __builtin_amdgcn_exp2f(s[8 * h + 2 * e + 1] - m_run);
        psum += p0 + p1;
        w[e] = pack_a2(p0, p1);
      }
      pf[h] = __builtin_bit_cast(a16x8, w);
    }
    l_run += psum;
    __builtin_amdgcn_sched_barrier(0);
    // ---- O^T += V^T . P^T  (512 d x 32 queries, contraction over the 32 keys); group g = d-tiles 2g, 2g+1
    static_for<8>([&](auto gc) {
      constexpr int g = decltype(gc)::value;
      if constexpr (g + 2 < 8) ldv(std::integral_constant<int, g + 2>{}, fr[(8 + g + 2) % 3]);
      static_for<ATTN_DMA_PER_GROUP>([&](auto kc) {
        constexpr int piece = (8 + g) * ATTN_DMA_PER_GROUP + decltype(kc)::value;
        if constexpr (piece < 16) issue_piece(std::integral_constant<int, piece>{}, nxt, BUF ^ 1);
      });
#pragma unroll
      for (int e = 0; e < 4; ++e)
        o[2 * g + (e >> 1)] = mfma_a16_32x32x16(fr[(8 + g) % 3][e], pf[e & 1], o[2 * g + (e >> 1)], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  issue(t_begin, 0);
  for (int tile = t_begin; tile < t_end; tile += 2) {
    tile_body(std::integral_constant<int, 0>{}, tile);
    if (tile + 1 < t_end) tile_body(std::integral_constant<int, 1>{}, tile + 1);
  }


  // ---- normalise and store O[q][d] (bf16): a lane owns ONE query row, 4 consecutive d per store
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  if (p.key_splits > 1) {   // partial: un-normalised accumulators + (max, sum); attn_combine_kernel merges the splits
    if (q_ok) {
      const size_t row = ((size_t)b * p.key_splits + ksplit) * p.N + qrow;
      float* po = p.part_o + row * HD;
#pragma unroll
      for (int dt = 0; dt < HD / 32; ++dt)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const int d = dt * 32 + 8 * rq + 4 * hi;
          *reinterpret_cast<f32x4*>(po + d) = f32x4{o[dt][4 * rq], o[dt][4 * rq + 1], o[dt][4 * rq + 2], o[dt][4 * rq + 3]};
        }
      if (hi == 0) { p.part_ml[row * 2] = m_run; p.part_ml[row * 2 + 1] = l_tot; }
    }
    return;
  }
  const float inv = 1.0f / l_tot;
  if (p.lse && q_ok && hi == 0) p.lse[(size_t)b * p.N + qrow] = m_run + __builtin_amdgcn_logf(l_tot);   // v_log_f32 = log2
  if (q_ok) {
    a16_t* op = p.o + ((size_t)b * p.N + qrow) * p.ldo;
#pragma unroll
    for (int dt = 0; dt < HD / 32; ++dt)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int d = dt * 32 + 8 * rq + 4 * hi;
        u32x2 w = {pack_a2(o[dt][4 * rq] * inv, o[dt][4 * rq + 1] * inv),
                   pack_a2(o[dt][4 * rq + 2] * inv, o[dt][4 * rq + 3] * inv)};
        *reinterpret_cast<u32x2*>(op + d) = w;
      }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Shared-K/V form (attn_kv_fwd_kernel): out[i] = sum_j softmax_j(q'_i . x_j) x_j  -- keys AND values are the same tensor x.
//
// AttnBlock (encoder_decoder.py:168-192) is single-head self-attention on ONE input h = GroupNorm(x):
//     s_ij = (Wq h_i + bq) . (Wk h_j + bk),   out_i = Wp (sum_j P_ij (Wv h_j + bv)) + bp
// Every term of s_ij that does not depend on j cancels in softmax_j, so s_ij ~ (Wk^T (Wq h_i + bq)) . h_j: the key projection
// folds into the query projection (q'_i = Wk^T Wq h_i + Wk^T bq), and because softmax rows sum to 1 the value projection
// commutes with the average (sum_j P_ij (Wv h_j + bv) = Wv (sum_j P_ij h_j) + bv) and folds into proj_out.  What is left in the
// N^2 part is attention with K = V = h:
//   * ONE tile per 32 keys streams through LDS instead of two (32 KB instead of 64 KB per tile: 8 LDS-DMA pieces per wave
//     and tile instead of 16 -- the issue cost of those pieces was ~25 % of the two-tensor kernel), and h is read from HBM / L2
//     once per query block instead of K and V^T;
//   * the K and V projections (two of the block's four 512x512 1x1 convs) disappear, and so does the transposed V^T copy.
// QK^T reads the tile as before (K-row A-fragments, ds_read_b128).  P.V needs the same tile TRANSPOSED (A = V^T rows = tile
// columns): gfx950's ds_read_b64_tr_b16 delivers exactly that from the row-major image -- per 16-lane group a 4-key x 16-d block,
// two reads per A-fragment.  The XOR swizzle f(key) = ((key & 3) << 2) | ((key >> 2) & 3) on the 16-B chunk index keeps BOTH
// read patterns bank-conflict free (b128: 16 lanes of a group have distinct key & 15, f is a bijection; tr_b16: the 4 keys x 2
// d-halves of a 32-lane pass fall into 8 distinct 32-B slots of the 256-B bank row).
// Everything else (one wave per SIMD, transposed products, lane-local online softmax, deferred rescale, split keys) is the
// structure of attn_fwd_kernel above.
typedef __attribute__((ext_vector_type(4))) short s16x4;


__device__ __forceinline__ int kv_swz(int key) { return ((key & 3) << 2) | ((key >> 2) & 3); }

// ds_read_b64_tr_b16 / its wait as free functions: clang rejects asm operands that are lambda captures
template <int OFF>
__device__ __forceinline__ void tr_read_b64(u32x2& dst, int addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
template <int N>
__device__ __forceinline__ void lgkm_wait8(u32x2 (&f)[8]) {   // ties the wait to the 8 destinations: nothing that reads them moves above
  asm volatile("s_waitcnt lgkmcnt(%8)"
               : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7])
               : "i"(N));
}

// PAIR: the output leaves as a hi / lo pair (p.o_lo); its own instantiation so that the plain kernel's register allocation -- the
// file is exactly full -- does not change
template <bool PAIR>
__global__ __launch_bounds__(AT_THREADS, 1) void attn_kv_fwd_kernel(const AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u32x4* lKV = reinterpret_cast<u32x4*>(smem);   // [2][KCH]: tile image [32 keys][64 chunks], chunk c of key r at r*64 + (c ^ f(r))

  int bid = blockIdx.x;
  {
    const int n = p.n_blocks, q = n / 8, r = n % 8, xcd = bid % 8, kk = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + kk;
  }
  const int qb = bid % p.n_qblocks, ksplit = (bid / p.n_qblocks) % p.key_splits, b = bid / (p.n_qblocks * p.key_splits);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ql = lane & 31, hi = lane >> 5;
  const int qrow = qb * BM + wave * 32 + ql;
  const bool q_ok = qrow < p.N;

  a16x8 qf[HD / 16];   // Q^T B-fragments: lane (q, hi) holds Q[q][16*ks + 8*hi .. +8]
  {
    const a16_t* qp = p.q + ((size_t)b * p.N + (q_ok ? qrow : 0)) * p.ldq + hi * 8;
#pragma unroll
    for (int ks = 0; ks < HD / 16; ++ks) {
      u32x4 v = *reinterpret_cast<const u32x4*>(qp + ks * 16);
      if (!q_ok) v = u32x4{0u, 0u, 0u, 0u};
      qf[ks] = __builtin_bit_cast(a16x8, v);
    }
  }
  f32x16 o[HD / 32];
#pragma unroll
  for (int i = 0; i < HD / 32; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  // Round 3, measured and NOT kept (alternated A/B on one box): (1) starting the score accumulator at -m_run
  // (written among the previous tile's last P.V MFMAs) so that the softmax needs no subtraction -- bit-correct, but 16 registers
  // carried across the loop in a register file that is exactly full (O 256 + Q 128) became 1036 B/lane of scratch: 21.4 ms
  // instead of 3.8; (2) the 16-score maximum as 7 v_max3_f32 + 1 v_max_f32 instead of hipcc's 15 v_max_f32: 3.82 vs 3.82 ms.
  float m_run = -1e30f, l_run = 0.f;

  const int n_tiles = (p.N + BN - 1) / BN;
  const int t_begin = (int)((long long)n_tiles * ksplit / p.key_splits), t_end = (int)((long long)n_tiles * (ksplit + 1) / p.key_splits);
  const a16_t* kbase = p.k + (size_t)b * p.N * p.ldk;
  // rows beyond N fall outside the descriptor's range and arrive as zeros: their scores are masked, their values are zero
  const __amdgpu_buffer_rsrc_t krsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<a16_t*>(kbase), 0, (int)((((long long)p.N - 1) * p.ldk + HD) * 2), 0x00020000);
  // piece i of this wave = key row r = wave + 4*i (1 KB): lane L writes LDS chunk r*64 + L and therefore fetches source chunk
  // L ^ f(r); f(r) = (wave << 2) | (i & 3) for these rows
  const int lane16w = (lane ^ (wave << 2)) * 16;
  auto issue_piece = [&](auto ic, int tile, int buf) {
    constexpr int i = decltype(ic)::value;
    const int r = wave + 4 * i;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(krsrc, (__attribute__((address_space(3))) void*)(lKV + buf * KCH + r * 64), 16,
                                             lane16w ^ ((i & 3) * 16), (tile * BN + r) * p.ldk * 2, 0, 0);
  };

  // QK^T: K row fetched for MFMA row slot ql: bits 2 and 3 swapped, so that output register r of lane (q, hi) is key
  // 16*(r>>3) + 8*hi + (r&7) -- the P^T B-fragment order (as in attn_fwd_kernel)
  const int krow = (ql & 0x13) | ((ql & 4) << 1) | ((ql & 8) >> 1);
  const int kf = kv_swz(krow);
  int kofs[8];          // K frag (ks): kofs[ks & 7] + (ks >> 3) * 256 + buf * 32 KB
#pragma unroll
  for (int bb = 0; bb < 8; ++bb) kofs[bb] = (krow * 64 + ((2 * bb + hi) ^ kf)) * 16;
  // P.V: transpose reads.  For MFMA (dt, kstep) and half h (keys 8*hi + 4*h .. +3 of the k-step), lane i of a 16-lane group
  // supplies the address of 4 consecutive d of key row 8*hi + 4*h + (i >> 2): d = 32*dt + 16*g4 + 4*(i & 3); it receives
  // V[those 4 keys][d = 32*dt + 16*g4 + i].   address = vofs[dt & 3][h] + kstep * 16 KB + (dt >> 2) * 256 + buf * 32 KB
  int vofs[4][2];
  {
    const int i16 = lane & 15, g4 = (lane >> 4) & 1;
#pragma unroll
    for (int dtl = 0; dtl < 4; ++dtl)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int key = 8 * hi + 4 * h + (i16 >> 2);
        const int chunk = 4 * dtl + 2 * g4 + ((i16 & 3) >> 1);
        vofs[dtl][h] = key * 1024 + ((chunk ^ kv_swz(key)) * 16) + (i16 & 1) * 8;
      }
  }

#define PROF_MARK(i) do {} while (0)
  auto tile_body = [&](auto bufc, int tile) {
    constexpr int BUF = decltype(bufc)::value;
    PROF_MARK(0);                                  // loop overhead
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    PROF_MARK(1);                                  // wait + barrier
    const int nxt = min(tile + 1, t_end - 1);  // last tile: a redundant reload keeps the loop branch-free
    const char* tb = smem + BUF * KCH * 16;
    a16x8 fr[3][4];
    auto ldk = [&](auto gc, a16x8(&f)[4]) {
      constexpr int g = decltype(gc)::value;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int ks = 4 * g + e;
        f[e] = *reinterpret_cast<const a16x8*>(tb + kofs[ks & 7] + (ks >> 3) * 256);
      }
    };
    // P.V fragments: transpose reads in inline asm.  Through the builtin (__builtin_amdgcn_ds_read_tr16_b64) hipcc cannot tell
    // that the read does not alias the LDS-DMA writes in flight into the OTHER buffer and puts `s_waitcnt vmcnt(0)` in front of
    // every transpose read that follows a DMA issue (measured: +800 cycles per tile, the whole DMA latency exposed twice).  An asm
    // read is invisible to that pass -- and to hipcc's lgkmcnt bookkeeping, so the P.V phase counts its own reads: they are the
    // only LDS operations between the end of QK^T and the next barrier, issued one group (8 reads) ahead, retired in order.
    auto ldv1 = [&](auto gc, auto ic, u32x2(&f)[8]) {   // read i of group g: f[i], i = 2*e + h
      constexpr int g = decltype(gc)::value, i = decltype(ic)::value, e = i >> 1;
      constexpr int dt = 2 * g + (e >> 1), ks = e & 1;
      tr_read_b64<BUF * KCH * 16 + ks * 16384 + (dt >> 2) * 256>(f[i], vofs[dt & 3][i & 1]);
    };
    auto ldv = [&](auto gc, u32x2(&f)[8]) {   // group g: d-tiles 2g, 2g+1; f[2*e + h], e = (dt & 1) * 2 + kstep, h = key half
      static_for<8>([&](auto ic) { ldv1(gc, ic, f); });
    };
    auto wait_v = [&](auto nc, u32x2(&f)[8]) {   // the 8 reads of `f` have landed once at most N newer LDS operations are in flight
      lgkm_wait8<decltype(nc)::value>(f);
    };
    auto vfrag = [&](const u32x2(&f)[8], int e) {
      return __builtin_bit_cast(a16x8, u32x4{f[2 * e][0], f[2 * e][1], f[2 * e + 1][0], f[2 * e + 1][1]});
    };
    u32x2 vf[2][8];

    // ---- S^T = K . Q^T
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    ldk(std::integral_constant<int, 0>{}, fr[0]);
    ldk(std::integral_constant<int, 1>{}, fr[1]);
    static_for<8>([&](auto gc) {
      constexpr int g = decltype(gc)::value;
      if constexpr (g + 2 < 8) ldk(std::integral_constant<int, g + 2>{}, fr[(g + 2) % 3]);
      issue_piece(gc, nxt, BUF ^ 1);
#pragma unroll
      for (int e = 0; e < 4; ++e) s = mfma_a16_32x32x16(fr[g % 3][e], qf[4 * g + e], s, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    });
    PROF_MARK(2);                                  // QK^T
    ldv(std::integral_constant<int, 0>{}, vf[0]);  // the first P.V fragments fly under the softmax
    if (tile == n_tiles - 1) {  // mask keys beyond N
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kv = tile * BN + 16 * (r >> 3) + 8 * hi + (r & 7);
        if (kv >= p.N) s[r] = -__builtin_inff();
      }
    }
    // ---- online softmax, lane-local per query column
    float mx = s[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
    {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
      mx = fmaxf(mx, __uint_as_float(hi ? sw[0] : sw[1]));
    }
    if (__any(mx > m_run + RESCALE_THR)) {  // wave-uniform: rescale everything still at the old max
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
#pragma unroll
      for (int i = 0; i < HD / 32; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float x = o[i][r], tmp;
          asm volatile("v_accvgpr_read_b32 %1, %0\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 %0, %1"
                       : "+a"(x), "=&v"(tmp)
                       : "v"(alpha));
          o[i][r] = x;
        }
      m_run = m_new;
    }
    float psum = 0.f;
    a16x8 pf[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      u32x4 w;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float p0 = __builtin_amdgcn_exp2f(s[8 * h + 2 * e] - m_run);
        const float p1 = __builtin_amdgcn_exp2f(s[8 * h + 2 * e + 1] - m_run);
        psum += p0 + p1;
        w[e] = pack_a2(p0, p1);
      }
      pf[h] = __builtin_bit_cast(a16x8, w);
    }
    l_run += psum;
    __builtin_amdgcn_sched_barrier(0);
    PROF_MARK(3);                                  // softmax
    // ---- O^T += V^T . P^T  (512 d x 32 queries, contraction over the 32 keys); group g = d-tiles 2g, 2g+1
    static_for<8>([&](auto gc) {
      constexpr int g = decltype(gc)::value;
      if constexpr (g + 1 < 8) {
        ldv(std::integral_constant<int, g + 1>{}, vf[(g + 1) & 1]);
        wait_v(std::integral_constant<int, 8>{}, vf[g & 1]);
      } else {
        wait_v(std::integral_constant<int, 0>{}, vf[g & 1]);
      }
#define VF_CUR vf[g & 1]
#pragma unroll
      for (int e = 0; e < 4; ++e)
        o[2 * g + (e >> 1)] = mfma_a16_32x32x16(vfrag(VF_CUR, e), pf[e & 1], o[2 * g + (e >> 1)], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    });
    PROF_MARK(4);                                  // P.V
  };

  static_for<8>([&](auto ic) { issue_piece(ic, t_begin, 0); });
  for (int tile = t_begin; tile < t_end; tile += 2) {
    tile_body(std::integral_constant<int, 0>{}, tile);
    if (tile + 1 < t_end) tile_body(std::integral_constant<int, 1>{}, tile + 1);
  }

  // ---- normalise and store O[q][d] (bf16): a lane owns ONE query row, 4 consecutive d per store
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  if (p.key_splits > 1) {
    if (q_ok) {
      const size_t row = ((size_t)b * p.key_splits + ksplit) * p.N + qrow;
      float* po = p.part_o + row * HD;
#pragma unroll
      for (int dt = 0; dt < HD / 32; ++dt)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const int d = dt * 32 + 8 * rq + 4 * hi;
          *reinterpret_cast<f32x4*>(po + d) = f32x4{o[dt][4 * rq], o[dt][4 * rq + 1], o[dt][4 * rq + 2], o[dt][4 * rq + 3]};
        }
      if (hi == 0) { p.part_ml[row * 2] = m_run; p.part_ml[row * 2 + 1] = l_tot; }
    }
    return;
  }
  const float inv = 1.0f / l_tot;
  if (q_ok) {
    a16_t* op = p.o + ((size_t)b * p.N + qrow) * p.ldo;
#pragma unroll
    for (int dt = 0; dt < HD / 32; ++dt)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int d = dt * 32 + 8 * rq + 4 * hi;
        u32x2 w = {pack_a2(o[dt][4 * rq] * inv, o[dt][4 * rq + 1] * inv),
                   pack_a2(o[dt][4 * rq + 2] * inv, o[dt][4 * rq + 3] * inv)};
        *reinterpret_cast<u32x2*>(op + d) = w;
        if constexpr (PAIR) {   // the output as a hi / lo pair: the operand of the fp32-class output projection (glare_conv_desc.k_wrap)
          const u32x2 l = {pack_a2(o[dt][4 * rq] * inv - alo(w[0]), o[dt][4 * rq + 1] * inv - ahi(w[0])),
                           pack_a2(o[dt][4 * rq + 2] * inv - alo(w[1]), o[dt][4 * rq + 3] * inv - ahi(w[1]))};
          *reinterpret_cast<u32x2*>(p.o_lo + ((size_t)b * p.N + qrow) * p.ldo + d) = l;
        }
      }
  }
}

// (Software-pipelined forms of this kernel -- three-stage, and QK^T(i+1) || P.V(i-1) + softmax(i) -- measured 1 035-1 089 / 997 TFLOP/s
// against 1 127-1 163 in round 2; 64-key tiles 8 % slower.  DESIGN.md section 3.)

// out[q] = sum_s 2^(m_s - m) O_s[q] / sum_s 2^(m_s - m) l_s,  m = max_s m_s: merges the key splits (one wave per query row)
__global__ __launch_bounds__(256) void attn_combine_kernel(const float* __restrict__ part_o, const float* __restrict__ part_ml,
                                                           a16_t* __restrict__ out, int ldo, int B, int N, int KS,
                                                           float* __restrict__ lse, a16_t* __restrict__ out_lo = nullptr) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);   // b * N + q
  if (row >= (long long)B * N) return;
  const int lane = threadIdx.x & 63, b = (int)(row / N), q = (int)(row % N);
  float m = -1e30f;
  for (int s = 0; s < KS; ++s) m = fmaxf(m, part_ml[(((size_t)b * KS + s) * N + q) * 2]);
  float acc[8], L = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  for (int s = 0; s < KS; ++s) {
    const size_t r = ((size_t)b * KS + s) * N + q;
    const float w = __builtin_amdgcn_exp2f(part_ml[r * 2] - m);
    L += w * part_ml[r * 2 + 1];
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(part_o + r * HD + lane * 8);
    const f32x4 v1 = *reinterpret_cast<const f32x4*>(part_o + r * HD + lane * 8 + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { acc[e] = fmaf(w, v0[e], acc[e]); acc[4 + e] = fmaf(w, v1[e], acc[4 + e]); }
  }
  const float inv = 1.0f / L;
  if (lse && lane == 0) lse[row] = m + __builtin_amdgcn_logf(L);
  u32x4 o = {pack_a2(acc[0] * inv, acc[1] * inv), pack_a2(acc[2] * inv, acc[3] * inv), pack_a2(acc[4] * inv, acc[5] * inv),
             pack_a2(acc[6] * inv, acc[7] * inv)};
  *reinterpret_cast<u32x4*>(out + (size_t)row * ldo + lane * 8) = o;
  if (out_lo) {
    u32x4 l;
#pragma unroll
    for (int e = 0; e < 4; ++e) l[e] = pack_a2(acc[2 * e] * inv - alo(o[e]), acc[2 * e + 1] * inv - ahi(o[e]));
    *reinterpret_cast<u32x4*>(out_lo + (size_t)row * ldo + lane * 8) = l;
  }
}

}  // namespace

static int attn_launch(const void* q, int ldq, const void* k, int ldk, const void* v_t, long long v_pitch, void* out, int ldo, int B,
                       int N, int key_splits, void* workspace, size_t workspace_bytes, glare_stream_t stream, float* lse = nullptr);

extern "C" int glare_attention_d512_lse_bf16(const void* q, int ldq, const void* k, int ldk, const void* v_t, long long v_pitch, void* out,
                                             int ldo, float* lse, int B, int N, int key_splits, void* workspace, size_t workspace_bytes,
                                             glare_stream_t stream) {
  if (!lse) return GLARE_ERR_INVALID;
  return attn_launch(q, ldq, k, ldk, v_t, v_pitch, out, ldo, B, N, key_splits, workspace, workspace_bytes, stream, lse);
}

extern "C" size_t glare_attention_d512_splitk_workspace_bytes(int B, int N, int key_splits) {
  if (B <= 0 || N <= 0 || key_splits <= 1) return 0;
  return (size_t)B * key_splits * N * (HD + 2) * sizeof(float);
}

extern "C" int glare_attention_d512_splitk_bf16(const void* q, int ldq, const void* k, int ldk, const void* v_t, long long v_pitch,
                                                void* out, int ldo, int B, int N, int key_splits, void* workspace,
                                                size_t workspace_bytes, glare_stream_t stream) {
  return attn_launch(q, ldq, k, ldk, v_t, v_pitch, out, ldo, B, N, key_splits, workspace, workspace_bytes, stream);
}

extern "C" int glare_attention_kv512_bf16(const void* q, int ldq, const void* kv, int ldkv, void* out, int ldo, int B, int N,
                                          int key_splits, void* workspace, size_t workspace_bytes, glare_stream_t stream) {
  return glare_attention_kv512_pair_bf16(q, ldq, kv, ldkv, out, nullptr, ldo, B, N, key_splits, workspace, workspace_bytes, stream);
}

extern "C" int glare_attention_kv512_pair_bf16(const void* q, int ldq, const void* kv, int ldkv, void* out, void* out_lo, int ldo, int B,
                                               int N, int key_splits, void* workspace, size_t workspace_bytes, glare_stream_t stream) {
  if (!q || !kv || !out || B <= 0 || N <= 0 || key_splits < 1) return GLARE_ERR_INVALID;
  if (key_splits > (N + BN - 1) / BN) return GLARE_ERR_INVALID;
  if (((long long)N * ldkv + HD) * 2 >= 0x7ff00000LL) return GLARE_ERR_UNSUPPORTED;   // 32-bit DMA offsets per image
  if (key_splits > 1 &&
      ((ldo % 8) || !workspace || workspace_bytes < glare_attention_d512_splitk_workspace_bytes(B, N, key_splits)))
    return GLARE_ERR_WORKSPACE;
  if ((ldq % 8) || (ldkv % 8) || (ldo % 4) || ldq < HD || ldkv < HD || ldo < HD) return GLARE_ERR_UNSUPPORTED;
  AttnParams p;
  p.q = (const a16_t*)q; p.k = (const a16_t*)kv; p.vt = nullptr; p.o = (a16_t*)out; p.o_lo = (a16_t*)out_lo;
  p.B = B; p.N = N; p.Npad = 0; p.ldq = ldq; p.ldk = ldkv; p.ldo = ldo; p.lse = nullptr;
  p.n_qblocks = (N + BM - 1) / BM;
  p.key_splits = key_splits;
  p.part_o = static_cast<float*>(workspace);
  p.part_ml = p.part_o ? p.part_o + (size_t)B * key_splits * N * HD : nullptr;
  const long long nb = (long long)B * p.n_qblocks * key_splits;
  if (nb > 0x7fffffffLL) return GLARE_ERR_INVALID;
  p.n_blocks = (int)nb;
  const size_t lds = (size_t)2 * KCH * 16;   // 64 KB: the double-buffered 32-key tile
  auto kern = (out_lo && key_splits == 1) ? attn_kv_fwd_kernel<true> : attn_kv_fwd_kernel<false>;   // with key splits the combine kernel writes the pair
  if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return GLARE_ERR_LAUNCH;
  hipLaunchKernelGGL(kern, dim3(p.n_blocks), dim3(AT_THREADS), lds, (hipStream_t)stream, p);
  if (key_splits > 1)
    hipLaunchKernelGGL(attn_combine_kernel, dim3((unsigned)(((long long)B * N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p.part_o,
                       p.part_ml, p.o, ldo, B, N, key_splits, (float*)nullptr, p.o_lo);
  return glare_launch_status();
}

extern "C" int glare_attention_d512_bf16(const void* q, int ldq, const void* k, int ldk, const void* v_t,
                                         long long v_pitch, void* out, int ldo, int B, int N, glare_stream_t stream) {
  return attn_launch(q, ldq, k, ldk, v_t, v_pitch, out, ldo, B, N, 1, nullptr, 0, stream);
}

static int attn_launch(const void* q, int ldq, const void* k, int ldk, const void* v_t, long long v_pitch, void* out, int ldo, int B,
                       int N, int key_splits, void* workspace, size_t workspace_bytes, glare_stream_t stream, float* lse) {
  if (!q || !k || !v_t || !out || B <= 0 || N <= 0 || key_splits < 1) return GLARE_ERR_INVALID;
  
  if (key_splits > (N + BN - 1) / BN) return GLARE_ERR_INVALID;   // every split owns at least one key tile
  if (((long long)N * ldk + HD) * 2 >= 0x7ff00000LL || (long long)HD * v_pitch * 2 >= 0x7ff00000LL) return GLARE_ERR_UNSUPPORTED;  // 32-bit DMA offsets per image
  if (key_splits > 1) {

    if ((ldo % 8) || !workspace || workspace_bytes < glare_attention_d512_splitk_workspace_bytes(B, N, key_splits)) return GLARE_ERR_WORKSPACE;
  }
  if ((ldq % 8) || (ldk % 8) || (ldo % 4) || (v_pitch % 8) || ldq < HD || ldk < HD || ldo < HD) return GLARE_ERR_UNSUPPORTED;
  if (v_pitch < (long long)((N + BN - 1) / BN) * BN) return GLARE_ERR_INVALID;  // tiles read whole 32-key groups
  AttnParams p;
  p.q = (const a16_t*)q; p.k = (const a16_t*)k; p.vt = (const a16_t*)v_t; p.o = (a16_t*)out; p.o_lo = nullptr;
  p.B = B; p.N = N; p.Npad = v_pitch; p.ldq = ldq; p.ldk = ldk; p.ldo = ldo; p.lse = lse;
  p.n_qblocks = (N + BM - 1) / BM;
  p.key_splits = key_splits;
  p.part_o = static_cast<float*>(workspace);
  p.part_ml = p.part_o ? p.part_o + (size_t)B * key_splits * N * HD : nullptr;
  const long long nb = (long long)B * p.n_qblocks * key_splits;
  if (nb > 0x7fffffffLL) return GLARE_ERR_INVALID;
  p.n_blocks = (int)nb;
  const size_t lds = (size_t)4 * KCH * 16;  // 128 KB K/V ring (+ 32 KB Q tail)
  if (hipFuncSetAttribute((const void*)attn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return GLARE_ERR_LAUNCH;
  hipLaunchKernelGGL(attn_fwd_kernel, dim3(p.n_blocks), dim3(AT_THREADS), lds, (hipStream_t)stream, p);
  if (key_splits > 1)
    hipLaunchKernelGGL(attn_combine_kernel, dim3((unsigned)(((long long)B * N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p.part_o,
                       p.part_ml, p.o, ldo, B, N, key_splits, lse);
  return glare_launch_status();
}
