// Shared by the translation units of the implicit-GEMM convolution (conv_igemm.hip: C ABI, packing, dispatch;
// conv_igemm_k3s1.hip / _k3s2.hip / _k1k2.hip: the kernel instantiations, split so that they compile in parallel -- one file
// with all of them was 4.5 minutes of the build's critical path).  The kernel itself is documented at the top of conv_igemm.hip.
#pragma once
#include <type_traits>

#include "common.h"



struct ConvParams {
  const a16_t* in0;
  const a16_t* in1;
  const a16_t* wpk;
  const float* bias;
  const a16_t* res;
  void* out;
  int B, H, W;        // source spatial size (before upsample)
  int IHs, IWs;       // conv input size (after upsample)
  int OH, OW;
  int Cin0, Cin1, CinTot;
  int p0, o0, p1, o1; // pitch / channel offset of the two sources
  int Cout, opitch, ooff, rpitch, roff;
  int upsample, act, out_mode;
  long long plane_pitch;
  int tiles_x, tiles_y, co_tiles, n_blocks, n_stages;
  int fast_epilogue;
  unsigned in0_bytes, in1_bytes;   // bytes of ONE image of each source (range of the halo DMA's buffer descriptors)
  unsigned res_bytes;              // bytes of ONE image of the residual, or 0: no L2 prefetch of the residual tile
  float* gn_part;   // optional GroupNorm partial sums of the OUTPUT: [b][part][Cout/4][2], part = tile*WM + wm
  int gn_nparts;
  const a16_t* res_lo;   // hi / lo epilogue (HILO instantiations): remainder halves of the residual and of the output
  a16_t* out_lo;
  // grouped launch (blockIdx.y = group): `groups` independent filters of one shape on channel slices of one tensor -- the flow's
  // 24 z-independent coupling nets, FlowAffineCouplingsAblation.py:143-151.  Group g reads input channels o0 + g * g_in_step,
  // writes output channels ooff + g * g_out_step, with the g-th packed filter / bias.
  int groups, g_in_step, g_out_step, g_bias_step;
  long long g_w_elems;
  int k_wrap;            // 1: K segments [in0 | in1 | in0 again]: the fp32-class contraction on hi / lo operand pairs (glare_conv_desc.k_wrap); 2: in0 tiles staged once
  const float* gn_coef;  // GNP instantiations: [B][Cin0][2] = (a, d) of the input's GroupNorm, applied to the halo tile in LDS
  int gn_swish;
};

// Which epilogue a launch takes is a COMPILE-TIME property of the kernel (round 5): with the three of them selected at run time inside
// one kernel, hipcc's register allocator never reused a dead accumulator register -- everything behind the K loop lived in the ~38
// registers above the 128 accumulators, with 80-500 B/lane of scratch (see ActSel below); one epilogue per kernel: none.
enum { EPI_FAST = 0,      // 16-bit NHWC, 16-B records: LDS-staged slabs, residual, fused GroupNorm statistics, sub-pixel scatter
       EPI_PLANAR = 1,    // fp32 planes through an fp32 LDS slab (the DCN's offset / mask-logit planes); 3x3 stride-1 4-wave kernels
       EPI_GENERAL = 2 }; // everything else, element by element from the C/D layout (odd pitches, fp32 NHWC, 16-bit planes, sigmoid /
                          // swish on the accumulators)
__host__ inline int conv_pick_epilogue(const ConvParams& p, bool planar_kernel) {
  const bool expk = p.act == GLARE_ACT_SIGMOID || p.act == GLARE_ACT_SWISH;
  if (p.out_mode == GLARE_OUT_NHWC_BF16 && p.fast_epilogue && !(p.res == nullptr && expk)) return EPI_FAST;
  if (planar_kernel && p.out_mode == GLARE_OUT_PLANAR_F32 && !p.res && !expk) return EPI_PLANAR;
  return EPI_GENERAL;
}

// kernel-family dispatchers: tn = the output-channel tile (128 / 64 / 32); hilo = the hi / lo epilogue.  The instantiations of a
// family are spread over one translation unit per epilogue kind so that the build compiles them in parallel.
int glare_conv_launch_k3s1(const ConvParams& p, int tn, bool hilo, hipStream_t stream);
int glare_conv_launch_k3s1_planar(const ConvParams& p, int tn, hipStream_t stream);
int glare_conv_launch_k3s1_general(const ConvParams& p, int tn, hipStream_t stream);
int glare_conv_launch_k3s2(const ConvParams& p, int tn, bool hilo, hipStream_t stream);
int glare_conv_launch_k3s2_general(const ConvParams& p, int tn, hipStream_t stream);
int glare_conv_launch_k1(const ConvParams& p, int tn, bool hilo, hipStream_t stream);
int glare_conv_launch_k1_general(const ConvParams& p, int tn, hipStream_t stream);
int glare_conv_launch_k2(const ConvParams& p, int tn, hipStream_t stream);   // sub-pixel upsample form (fast epilogue only)

namespace {

constexpr int TW = 32;   // output tile cols == MFMA M



template <int KS, int STRIDE, int TH>
struct TileGeom {
  static constexpr int IH = (TH - 1) * STRIDE + KS;
  static constexpr int IW = (TW - 1) * STRIDE + KS;
  static constexpr int NPOS = IH * IW;
  static constexpr int PAD = (KS == 3 && STRIDE == 1) ? 1 : 0;  // stride 2: pad (0,1,0,1) only
};

template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

// The activation inside the epilogues.  History: as a runtime `switch` per output element it compiled to ~3 scalar branches per element
// (round 1: a quarter of a 128-channel tile's MFMA time); dispatched ONCE per workgroup into four compile-time copies of the whole
// epilogue (rounds 2-4) the branches were gone -- but so was the register allocator's view of the accumulators' lifetime: with the
// epilogue cloned behind a switch, hipcc never reused a dead accumulator register, the entire epilogue lived in the ~38 registers above
// the 128 accumulators (80 B/lane of scratch, every residual piece loaded into the SAME four registers and waited for one by one), and
// any attempt to keep more in flight spilled (round 3's residual-rows-ahead: 196 B/lane; round 5's first try: 276 B/lane with
// `s_waitcnt vmcnt(0)` behind every piece).  With ONE copy of the epilogue the residual pieces land in the dead accumulators.
// So the activation is a run-time value again, but branch-FREE where it is hot: the path only ever fuses none / relu into this kernel
// (flow nets, VGG), which is a v_max + a select on a wave-uniform flag; sigmoid / swish (tests, training-side convs) sit behind ONE
// uniform branch per 8-element group, or -- where they act on the accumulators themselves -- in a pre-pass over them.
// (ActSel, act_sel, act_cheap, act_exp, act_any: common.h -- shared with conv1x1.hip)


// The lane id again, from nothing (v_mbcnt): the epilogues' `lane` would otherwise be carried -- at 168 registers: spilled -- through the
// K loop, which needs only values derived from it.
__device__ __forceinline__ int fresh_lane_id() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

// One 16-B piece (8 channels of one halo position) of the GNP prologue, in place in LDS.
__device__ __forceinline__ void gn_prologue_piece(u32x4* slot, const float* cf, int swish) {
  __builtin_amdgcn_sched_barrier(0);       // nothing of the MFMA loop is scheduled into the transform, and vice versa
  u32x4 v = *slot;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const f32x4 q = *reinterpret_cast<const f32x4*>(cf + 4 * e);     // (a, d, a, d) of two channels
    float lo = alo(v[e]) * q[0] + q[1];
    float hi = ahi(v[e]) * q[2] + q[3];
    if (swish) { lo = swishf_(lo); hi = swishf_(hi); }
    v[e] = pack_a2(lo, hi);
    __builtin_amdgcn_sched_barrier(0);
  }
  *slot = v;
  __builtin_amdgcn_sched_barrier(0);
}

// KS kernel size, STRIDE, MT/NT 32x32 MFMA tiles per wave along pixel rows / couts,
// WM x WN waves (WM*MT == 8 rows), KSTEPS 16-channel k-steps per input stage.
//
// Pipeline: an "A stage" is KC = 16*KSTEPS channels of the halo tile; it is consumed in KS
// "B stages" (one tap row each: KS taps x KSTEPS k-steps of weights).  Both images are
// double-buffered in LDS and filled by LDS-DMA (global_load_lds_dwordx4), issued one B stage
// ahead, right after the barrier that retires the buffer they overwrite; one barrier per B stage.
// GNP (round 4): GroupNorm (+ swish) of the INPUT applied by the loader.  The halo tile reaches LDS raw by DMA as
// always; every thread then rewrites THE PIECES IT ISSUED ITSELF in place -- y = round16(swish(a x + d)), (a, d) per (image, channel)
// from a table in LDS -- behind its own `s_waitcnt vmcnt(0)` and in front of the stage barrier that is there anyway: no extra barrier
// per stage, no cross-wave dependency, and the stage being transformed (chunk c + 1, landed during tap row 0 of chunk c) is not the
// one the MFMAs read.  Padding positions (out-of-range DMA offsets = zeros) are skipped: zero padding stays zero, as the reference
// pads the NORMALISED tensor.  The arithmetic is gn_apply_kernel's, term for term: the result is bit-identical to conv(gn_apply(x)).
template <int KS, int STRIDE, int MT, int NT, int WM, int WN, int KSTEPS, bool HILO = false, bool GNP = false, int EPI = EPI_FAST>
__global__ __launch_bounds__(64 * WM * WN, ((STRIDE == 1 && WM * WN == 4) ? 3 : 2)) void conv_igemm_kernel(const ConvParams p_in) {
  ConvParams p = p_in;
  if (p.groups > 1) {               // uniform (scalar) adjustments: this workgroup's group
    const int g = blockIdx.y;
    p.o0 += g * p.g_in_step;
    p.o1 += g * p.g_in_step;
    p.ooff += g * p.g_out_step;
    p.wpk += (size_t)g * p.g_w_elems;
    if (p.bias) p.bias += g * p.g_bias_step;
  }
  constexpr int TH = WM * MT;       // output tile rows
  constexpr int NW = WM * WN;       // waves per workgroup (4, or 6 for the 12 x 32 x 128 tile)
  using G = TileGeom<KS, STRIDE, TH>;
  static_assert(NW == 4 || NW == 6 || NW == 8, "wave layout");
  constexpr int TN = WN * NT * 32;
  constexpr int A_CHUNKS = KSTEPS * 2 * G::NPOS;           // 16-B chunks of one A stage
  constexpr int A_INSTR = (A_CHUNKS + 63) / 64;            // wave-level DMA instructions per A stage
  constexpr int A_SLOTS = A_INSTR * 64;
  constexpr int A_PER_W = (A_INSTR + NW - 1) / NW;
  constexpr int B_CHUNKS = KS * KSTEPS * 2 * TN;           // 16-B chunks of one B stage (one tap row)
  static_assert(B_CHUNKS % 64 == 0, "B stage is a whole number of wave DMAs");
  constexpr int B_INSTR = B_CHUNKS / 64;
  constexpr int B_PER_W = (B_INSTR + NW - 1) / NW;
  constexpr int KC = 16 * KSTEPS;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  u32x4* lA = reinterpret_cast<u32x4*>(smem);              // [2][A_SLOTS]
  u32x4* lB = lA + 2 * A_SLOTS;                            // [2][B_CHUNKS]

  // XCD-aware order: consecutive logical tiles (same pixels, different couts; then neighbouring
  // pixels) run on one XCD and share its L2 (dispatch is round-robin over the 8 XCDs).
  int bid = blockIdx.x;
  {
    const int n = p.n_blocks, q = n / 8, r = n % 8, xcd = bid % 8, k = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int ct = bid % p.co_tiles;
  int t = bid / p.co_tiles;
  int phase = 0;
  if (KS == 2) { phase = t & 3; t >>= 2; }
  const int pa = phase >> 1, pb = phase & 1;   // output row / column parity of this sub-pixel phase
  const int tx = t % p.tiles_x;
  t /= p.tiles_x;
  const int ty = t % p.tiles_y;
  const int b = t / p.tiles_y;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int oy0 = ty * TH, ox0 = tx * TW;
  const int iy0 = oy0 * STRIDE - G::PAD - (KS == 2 ? 1 - pa : 0), ix0 = ox0 * STRIDE - G::PAD - (KS == 2 ? 1 - pb : 0);

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // per-lane invariant part of the A addressing: this wave issues DMA instructions j = wave + NW*i; lane handles chunk
  // c = j*64 + lane = (halo row * 2*KSTEPS + kstep*2 + khalf) * IW + x.  The halo goes through buffer descriptors based at THIS IMAGE (32-bit byte
  // offsets: one image of one source stays below 2 GB): a loop-invariant offset per lane and piece (pixel, 8-channel half)
  // plus the stage's channel offset as the scalar offset; padded positions, slots beyond the tile and channels beyond Cin
  // carry an offset beyond the descriptor's range, which the hardware turns into zeros -- no 64-bit address arithmetic, no
  // divergent branches and no zero source in the issue sequence.
  constexpr unsigned A_OOB = 0x80000000u;
  unsigned a_vo0[A_PER_W], a_vo1[A_PER_W];
  int a_ck[A_PER_W];         // channel offset inside the stage: kstep*16 + khalf*8
#pragma unroll
  for (int i = 0; i < A_PER_W; ++i) {
    const int c = (wave + NW * i) * 64 + lane;
    a_vo0[i] = a_vo1[i] = A_OOB;
    a_ck[i] = 0;
    if (c < A_CHUNKS) {
      // LDS image [halo row][8-channel plane][x]: the planes of one halo row are NEIGHBOURS in a DMA piece, so the 2 * KSTEPS lanes that
      // read the same pixel's 128-B line sit in the same instruction (or the next) instead of NPOS slots apart -- the guide's "fragment-
      // shaped loads cost TA cycles" at the scale this tile allows (round 5: 3x3 -1...2 %, the 1x1 form -11...18 %, +1.05 % on the bench)
      const int prow = c / (2 * KSTEPS * G::IW), rem = c % (2 * KSTEPS * G::IW);
      const int kk = rem / G::IW, pos = prow * G::IW + rem % G::IW;
      a_ck[i] = kk * 8;
      const int iy = iy0 + pos / G::IW, ix = ix0 + pos % G::IW;
      if (iy >= 0 && iy < p.IHs && ix >= 0 && ix < p.IWs) {
        const int sy = p.upsample ? (iy >> 1) : iy, sx = p.upsample ? (ix >> 1) : ix;
        const unsigned pix = (unsigned)(sy * p.W + sx);
        a_vo0[i] = (pix * (unsigned)p.p0 + (unsigned)(p.o0 + kk * 8)) * 2u;
        if constexpr (!GNP) a_vo1[i] = (pix * (unsigned)p.p1 + (unsigned)(p.o1 + kk * 8)) * 2u;   // (the GroupNorm-prologue form has one source)
      }
    }
  }
  const size_t img = (size_t)b * p.H * p.W;
  const __amdgpu_buffer_rsrc_t arsrc0 =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<a16_t*>(p.in0 + img * p.p0), 0, (int)p.in0_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t arsrc1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<a16_t*>(p.in1 ? p.in1 + img * p.p1 : p.in0), 0, (int)(p.in1 ? p.in1_bytes : p.in0_bytes), 0x00020000);
  const a16_t* wbase = p.wpk + ((size_t)phase * p.co_tiles + ct) * p.n_stages * (size_t)(KS * B_CHUNKS) * 8;

  // k_wrap = 2 (round 5): stages 2c and 2c + 1 contract the SAME x_hi halo tile (chunk c of source 0) with the filter's hi and lo halves,
  // and the x_lo segment follows behind all of them: x_hi is staged once instead of twice (ops.split_filter, reuse_kc)
  const bool reuse = KS == 3 && !GNP && p.k_wrap == 2;                 // (glare_conv2d_bf16 rejects k_wrap = 2 anywhere else)
  const int n_pair = reuse ? 2 * (p.Cin0 / KC) : 0;                    // stages of the paired (x_hi) part
  auto a_is_new = [&](int stage) { return !(reuse && stage < n_pair && (stage & 1)); };
  auto issue_a = [&](int chunk, int buf, int i_lo = 0, int i_hi = 1 << 20) {   // pieces [i_lo, i_hi) of this wave; chunk = K stage
    const int c0 = chunk * KC;
    // uniform: a stage never straddles two K segments -- [in0 | in1] and, with k_wrap, in0 once more behind them
    bool src0 = GNP || c0 < p.Cin0 || c0 >= p.Cin0 + p.Cin1;
    int cbase = c0 < p.Cin0 ? c0 : (src0 ? c0 - p.Cin0 - p.Cin1 : c0 - p.Cin0);
    if (reuse) {
      src0 = chunk < n_pair;
      cbase = src0 ? (chunk >> 1) * KC : (chunk - n_pair) * KC;
    }
    const int climit = src0 ? p.Cin0 : p.Cin1;
#pragma unroll
    for (int i = 0; i < A_PER_W; ++i) {
      const int j = wave + NW * i;
      if (i >= i_lo && i < i_hi && j < A_INSTR) {
        unsigned vo = src0 ? a_vo0[i] : a_vo1[i];
        if (cbase + a_ck[i] >= climit) vo = A_OOB;       // channels beyond Cin (last stage only): v_cndmask, not a branch
        if (src0)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(arsrc0, (__attribute__((address_space(3))) void*)(lA + buf * A_SLOTS + j * 64), 16, vo,
                                                   cbase * 2, 0, 0);
        else
          __builtin_amdgcn_raw_ptr_buffer_load_lds(arsrc1, (__attribute__((address_space(3))) void*)(lA + buf * A_SLOTS + j * 64), 16, vo,
                                                   cbase * 2, 0, 0);
      }
    }
  };
  // Weight stages go through a buffer descriptor: `buffer_load_dwordx4 ... offen lds` takes ONE per-lane
  // 32-bit offset (lane*16, loop-invariant) plus a scalar offset -- no per-instruction 64-bit VALU address.
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<a16_t*>(wbase), 0, (int)((size_t)p.n_stages * KS * B_CHUNKS * 16), 0x00020000);
  const int lane16 = lane * 16;
  auto issue_b = [&](int bstage, int buf, int i_lo = 0, int i_hi = 1 << 20) {
#pragma unroll
    for (int i = 0; i < B_PER_W; ++i) {
      const int j = wave + NW * i;
      if (i >= i_lo && i < i_hi && j < B_INSTR)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (__attribute__((address_space(3))) void*)(lB + buf * B_CHUNKS + j * 64),
                                                 16, lane16, (bstage * B_CHUNKS + j * 64) * 16, 0, 0);
    }
  };

  // The residual tile is read by the epilogue only, behind the last MFMA: 16 B per lane straight from HBM, a full memory round
  // trip (several us with every CU streaming) during which this workgroup does nothing else -- measured as +11 us per round of 768
  // tiles whatever the length of the K loop (tools/probes/conv_k_sweep.py).  So the tile's 128-B lines are pulled into the L2 three
  // B stages before the end: one dword per line by LDS-DMA (no destination registers) into the A buffer that the last chunk leaves
  // unused; the data is never read, the epilogue's loads then hit the L2.
  [[maybe_unused]] auto prefetch_residual = [&](int free_buf) {
    if constexpr (KS == 3 && STRIDE == 1 && NW == 4 && TN == 128) {
      static_assert(A_SLOTS * 16 >= NW * 4 * 256, "room for the prefetch's landing zone");
      const size_t rimg = (size_t)b * p.OH * p.OW * p.rpitch;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int l = (wave * 2 + k) * 64 + lane;                  // line of the tile: pixel (l >> 1), 64-channel half (l & 1)
        const int oy = oy0 + (l >> 6), ox = ox0 + ((l >> 1) & 31), co = ct * TN + (l & 1) * 64;
        const unsigned vo = (oy < p.OH && ox < p.OW && co < p.Cout)
                                ? (unsigned)(((oy * p.OW + ox) * p.rpitch + p.roff + co) * 2) : 0x80000000u;
        const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<a16_t*>(p.res + rimg), 0, (int)p.res_bytes, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rr, (__attribute__((address_space(3))) void*)(lA + free_buf * A_SLOTS + (wave * 4 + k) * 16),
                                                 4, vo, 0, 0, 0);
        if constexpr (HILO) {
          if (p.res_lo) {
            const __amdgpu_buffer_rsrc_t rl =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<a16_t*>(p.res_lo + rimg), 0, (int)p.res_bytes, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rl, (__attribute__((address_space(3))) void*)(lA + free_buf * A_SLOTS + (wave * 4 + 2 + k) * 16),
                                                     4, vo, 0, 0, 0);
          }
        }
      }
    }
  };

  [[maybe_unused]] float* lC = reinterpret_cast<float*>(lB + 2 * B_CHUNKS);   // GNP: (a, d) of this image's Cin0 channels
  [[maybe_unused]] auto gn_transform = [&](int chunk, int buf) {
    if constexpr (GNP) {
      const int c0 = chunk * KC;
#pragma unroll
      for (int i = 0; i < A_PER_W; ++i) {
        const int j = wave + NW * i;
        if (j < A_INSTR && a_vo0[i] != A_OOB && c0 + a_ck[i] < p.Cin0) {
          // the addresses are formed HERE, from a value the compiler cannot see through: hoisted out of the K loop as invariants
          // (nine more live registers in a kernel that has none) they put a scratch store + load next to every MFMA -- 4.8 ms
          // for a 0.6 ms conv, measured
          int ln = fresh_lane_id(), ck = a_ck[i];
          asm volatile("" : "+v"(ln), "+v"(ck));
          gn_prologue_piece(lA + buf * A_SLOTS + j * 64 + ln, lC + (c0 + ck) * 2, p.gn_swish);
        }
      }
    }
  };
  if constexpr (GNP) {
    const float* src = p.gn_coef + (size_t)b * p.Cin0 * 2;
    for (int i = tid; i < p.Cin0 * 2; i += 64 * NW) lC[i] = src[i];
    __syncthreads();
  }

  const int khalf = lane >> 5, px = lane & 31;
  const int n_bstages = p.n_stages * KS;
  issue_a(0, 0);
  issue_b(0, 0);
  int bs = 0;
  int a_cnt = 0;       // halo tiles staged so far - 1: tile t lives in A buffer t & 1 (= the stage index unless k_wrap = 2 reuses tiles)
  for (int chunk = 0; chunk < p.n_stages; ++chunk) {
    const u32x4* cA = lA + (a_cnt & 1) * A_SLOTS;
    const bool next_new = chunk + 1 < p.n_stages && a_is_new(chunk + 1);
#pragma unroll
    for (int trow = 0; trow < KS; ++trow, ++bs) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA pieces have landed
      if constexpr (GNP) {                               // ... and are normalised in place before anybody reads them
        if (trow == 0 && chunk == 0) gn_transform(0, 0);
        if (trow == 1 && chunk + 1 < p.n_stages) gn_transform(chunk + 1, (a_cnt + 1) & 1);
      }
      __syncthreads();                                   // ... and everybody else's; stage bs-1 retired
      // The next stage's DMA pieces are not issued in one burst behind the barrier (12 waves would queue 47 pieces on the
      // CU's address path with the matrix pipe waiting): they are spread over the stage's KS*KSTEPS MFMA groups.
      constexpr int NGRP = KS * KSTEPS;
      [[maybe_unused]] constexpr int B_PG = (B_PER_W + NGRP - 1) / NGRP, A_PG = (A_PER_W + NGRP - 1) / NGRP;
      const bool more_b = bs + 1 < n_bstages, more_a = trow == 0 && next_new;
      if (trow == 0 && chunk + 1 == p.n_stages && p.res_bytes) prefetch_residual((a_cnt + 1) & 1);
      const u32x4* cB = lB + (bs & 1) * B_CHUNKS;
#pragma unroll
      for (int tcol = 0; tcol < KS; ++tcol) {
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
          {
            const int grp = tcol * KSTEPS + ks;
            if (more_b) issue_b(bs + 1, (bs + 1) & 1, grp * B_PG, (grp + 1) * B_PG);
            if (more_a) issue_a(chunk + 1, (a_cnt + 1) & 1, grp * A_PG, (grp + 1) * A_PG);
          }
          a16x8 bf[NT], af[MT];
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            const int co = (wn * NT + j) * 32 + px;
            bf[j] = __builtin_bit_cast(a16x8, cB[((tcol * KSTEPS + ks) * 2 + khalf) * TN + co]);
          }
#pragma unroll
          for (int i = 0; i < MT; ++i) {
            const int row = wm * MT + i;
            af[i] = __builtin_bit_cast(a16x8, cA[((row * STRIDE + trow) * (2 * KSTEPS) + ks * 2 + khalf) * G::IW + px * STRIDE + tcol]);
          }
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
              acc[i][j] = mfma_a16_32x32x16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
      }
    }
    if (next_new) ++a_cnt;
  }

  // ---- hi / lo epilogue: the output keeps 22 mantissa bits as two 16-bit tensors (the residual stream of the conditional
  // encoder in the fp16 precision).  The accumulators go through a wave-private fp32 slab, one tile row
  // (32 pixels x NT*32 couts) at a time; v = acc + bias + residual_hi + residual_lo in fp32, then hi = round16(v) and
  // lo = round16(v - hi) leave as two 16-B stores per lane.  Unlike the plain epilogue nothing is rounded before the residual add.
  // (Requesting a tile row's residual halves BEFORE its pass through the slab -- 32 more registers next to the 128 accumulators at
  // 168 -- spilled: 196 B/lane of scratch, 892 instead of 823 us per launch on the path's eight launches; kept out.)
  if constexpr (HILO) {
    constexpr int ROWF = NT * 128 + 16;                  // slab row pitch in bytes (pad: bank spread between rows)
    constexpr int CPR = NT * 4;                          // 8-channel chunks per slab row
    static_assert(NW * 32 * ROWF <= (2 * A_SLOTS + 2 * B_CHUNKS) * 16, "hi/lo epilogue slab fits the pipeline LDS");
    static_assert(KS != 2, "the sub-pixel form has no hi/lo epilogue");
    __syncthreads();  // every wave is done reading the pipeline buffers
    char* slab = smem + wave * (32 * ROWF);
    const int lane = fresh_lane_id();
    const int ncol = lane & 31, rhalf = lane >> 5;
    float gs0 = 0.f, gq0 = 0.f, gs1 = 0.f, gq1 = 0.f;
    const ActSel asel = act_sel(p.act);
    static_for<MT>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      static_for<NT>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const int co = ct * TN + (wn * NT + j) * 32 + ncol;
        const float bv = (p.bias && co < p.Cout) ? p.bias[co] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = (r & 3) + 8 * (r >> 2) + 4 * rhalf;           // pixel x within the tile row
          *reinterpret_cast<float*>(slab + m * ROWF + (j * 32 + ncol) * 4) = acc[i][j][r] + bv;
        }
      });
      // The residual pieces of THIS tile row (hi and lo halves: 2 x 16 B per lane and iteration) are all requested here, behind the
      // slab writes -- the row's accumulators died with those writes, so the 8 x 4 destination registers cost nothing -- and the loop
      // below consumes them in issue order.  Loaded inside the loop (rounds 3-4) every iteration paid its own L2 round trip with
      // nothing else in flight: 16 dependent trips per wave and tile, the larger half of what "+ pair residual" cost (r04_kbench).
      // (Requested BEFORE the slab pass they competed with the live accumulators and spilled: 892 vs 823 us, round 3.)
      constexpr int NIT = 32 * CPR / 64;
      [[maybe_unused]] u32x4 rvv[NIT], rlv[NIT];
      // through the per-image buffer descriptor of the L2 prefetch (p.res_bytes != 0): ONE loop-invariant 32-bit lane offset, the
      // iteration's part a constant -- no 64-bit address pair per piece (eight of them spilled: 252 B/lane of scratch) -- and rows /
      // channels beyond the image read zeros or unused neighbours instead of branching
      const bool res_buf = p.res && p.res_bytes;
      __builtin_amdgcn_sched_barrier(0);   // the requests stay BEHIND the slab writes: hoisted above them they meet the row's live accumulators
      if (res_buf) {
        const size_t rimg = (size_t)b * p.OH * p.OW * p.rpitch;
        const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<a16_t*>(p.res + rimg), 0, (int)p.res_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(const_cast<a16_t*>((p.res_lo ? p.res_lo : p.res) + rimg), 0,
                                                                            (int)p.res_bytes, 0x00020000);
        const int oy = oy0 + wm * MT + i;
        const unsigned vo = oy < p.OH ? (unsigned)((((oy * p.OW + ox0 + lane / CPR) * p.rpitch) + p.roff + ct * TN + wn * NT * 32 + (lane % CPR) * 8) * 2)
                                      : 0x80000000u;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const unsigned o = vo + (unsigned)(it * (64 / CPR) * p.rpitch * 2);
          rvv[it] = __builtin_amdgcn_raw_buffer_load_b128(rr, o, 0, 0);
          if (p.res_lo) rlv[it] = __builtin_amdgcn_raw_buffer_load_b128(rl, o, 0, 0);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int idx = lane + 64 * it;
        const int row = idx / CPR, ch = idx % CPR;
        const int oy = oy0 + wm * MT + i, ox = ox0 + row;
        const int co = ct * TN + wn * NT * 32 + ch * 8;
        if (oy < p.OH && ox < p.OW && co < p.Cout) {
          const f32x4 f0 = *reinterpret_cast<const f32x4*>(slab + row * ROWF + ch * 32);
          const f32x4 f1 = *reinterpret_cast<const f32x4*>(slab + row * ROWF + ch * 32 + 16);
          float v[8] = {f0[0], f0[1], f0[2], f0[3], f1[0], f1[1], f1[2], f1[3]};
          const size_t pix = ((size_t)b * p.OH + oy) * p.OW + ox;
          if (p.res) {
            const u32x4 rv = res_buf ? rvv[it] : *reinterpret_cast<const u32x4*>(p.res + pix * p.rpitch + p.roff + co);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[2 * e] += alo(rv[e]); v[2 * e + 1] += ahi(rv[e]); }
            if (p.res_lo) {
              const u32x4 rl = res_buf ? rlv[it] : *reinterpret_cast<const u32x4*>(p.res_lo + pix * p.rpitch + p.roff + co);
#pragma unroll
              for (int e = 0; e < 4; ++e) { v[2 * e] += alo(rl[e]); v[2 * e + 1] += ahi(rl[e]); }
            }
          }
          if (asel.expk) {            // one uniform branch per 8-element group
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = act_exp(v[e], asel.expk);
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = act_cheap(v[e], asel.relu);
          }
          u32x4 hi, lo;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float a = v[2 * e], c = v[2 * e + 1];
            hi[e] = pack_a2(a, c);
            lo[e] = pack_a2(a - alo(hi[e]), c - ahi(hi[e]));
            if (e < 2) { gs0 += a + c; gq0 += a * a + c * c; } else { gs1 += a + c; gq1 += a * a + c * c; }
          }
          *reinterpret_cast<u32x4*>(reinterpret_cast<a16_t*>(p.out) + pix * p.opitch + p.ooff + co) = hi;
          *reinterpret_cast<u32x4*>(p.out_lo + pix * p.opitch + p.ooff + co) = lo;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    });
    if (p.gn_part) {  // lanes with the same channel chunk are CPR apart: fold them, one lane per chunk writes
#pragma unroll
      for (int o = CPR; o < 64; o <<= 1) {
        gs0 += __shfl_xor(gs0, o, 64); gq0 += __shfl_xor(gq0, o, 64);
        gs1 += __shfl_xor(gs1, o, 64); gq1 += __shfl_xor(gq1, o, 64);
      }
      const int co = ct * TN + wn * NT * 32 + lane * 8;
      if (lane < CPR && co < p.Cout) {
        const int row0 = oy0 + wm * MT;
        const int part = ((phase * ((p.OH + 7) / 8) + row0 / 8) * p.tiles_x + tx) * 2 + ((row0 / 4) & 1);
        if (part < p.gn_nparts) {
          float* dst = p.gn_part + (((size_t)b * p.gn_nparts + part) * (p.Cout / 4) + co / 4) * 2;
          dst[0] = gs0; dst[1] = gq0; dst[2] = gs1; dst[3] = gq1;
        }
      }
    }
    return;
  }

  // ---- fast epilogue (bf16 NHWC output, 16-B aligned records): the accumulators go through LDS so
  // that the residual read and the output write are 16 B per lane, 128 contiguous bytes per pixel
  // (the direct C/D layout would store 2 B per lane).  Each wave stages its own 64-row slab (wave-
  // private region: no workgroup barrier inside), two slabs of MT/2 tile rows per wave.
  //   phase 1: acc + bias (+act when there is no residual) -> bf16; neighbouring lanes (co n, n+1) swap one
  //            value so every lane writes one packed 4-B word: even lanes row m(2t), odd lanes row m(2t+1)
  //   phase 2: 16-B LDS reads, + residual (16-B global load, fp32 add), act, 16-B global store
  // (sigmoid / swish on the accumulators themselves -- no residual in between -- take the general epilogue: any form of it here, a
  // pre-pass over the accumulators or a branch per 32 x 32 block, cost the hot none / relu path registers; conv_pick_epilogue)
  if constexpr (EPI == EPI_FAST) {
    constexpr int HT = (NW == 8 || MT < 2) ? 1 : MT / 2;  // tile rows per slab (smaller slabs when 8 waves share the LDS)
    constexpr int HROWS = HT * 32;                       // slab rows
    constexpr int ROWB = NT * 64 + 16;                   // slab row pitch in bytes (pad: bank spread)
    constexpr int CPR = NT * 4;                          // 16-B chunks per slab row
    static_assert(NW * HROWS * ROWB <= (2 * A_SLOTS + 2 * B_CHUNKS) * 16, "epilogue slab fits the pipeline LDS");
    __syncthreads();  // every wave is done reading the pipeline buffers
    char* slab = smem + wave * (HROWS * ROWB);
    const int lane = fresh_lane_id();
    const int ncol = lane & 31, rhalf = lane >> 5, odd = lane & 1;
    const bool act_early = p.res == nullptr;
    // fused GroupNorm statistics of the tensor being written (the consumer's gn_stats pass would re-read it):
    // per lane the sum / sum of squares of its two 4-channel units over its pixels, from the ROUNDED values
    float gs0 = 0.f, gq0 = 0.f, gs1 = 0.f, gq1 = 0.f;
    const ActSel asel = act_sel(p.act);
    const bool relu_early = act_early && asel.relu;
    static_for<MT / HT>([&](auto hc) {
      constexpr int half = decltype(hc)::value;
      static_for<NT>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const int co = ct * TN + (wn * NT + j) * 32 + ncol;
        const float bv = (p.bias && co < p.Cout) ? p.bias[co] : 0.f;
        static_for<HT>([&](auto ic) {
          constexpr int il = decltype(ic)::value;
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            constexpr int i = half * HT + il;
            const float a = act_cheap(acc[i][j][2 * t] + bv, relu_early), c = act_cheap(acc[i][j][2 * t + 1] + bv, relu_early);
            const float send = odd ? a : c;
            const float recv = __shfl_xor(send, 1, 64);
            const int r = 2 * t + odd;                                  // the register (row) this lane writes
            const int m = (r & 3) + 8 * (r >> 2) + 4 * rhalf;            // pixel x within the tile row
            const uint32_t w = odd ? pack_a2(recv, c) : pack_a2(a, recv);
            *reinterpret_cast<uint32_t*>(slab + (il * 32 + m) * ROWB + (j * 32 + (ncol & ~1)) * 2) = w;
          }
        });
      });
      // the residual pieces of this slab, all requested at once behind phase 1 (whose accumulators are dead by now): see the hi / lo
      // epilogue above -- one L2 round trip per slab instead of one per iteration
      constexpr int NIT = HROWS * CPR / 64;
      constexpr bool RES_AHEAD = NIT <= 8 && KS != 2;
      [[maybe_unused]] u32x4 rvv[RES_AHEAD ? NIT : 1];
      bool res_buf = false;
      if constexpr (RES_AHEAD) {
        res_buf = p.res && p.res_bytes;     // the per-image descriptor of the L2 prefetch: one 32-bit lane offset + a constant per piece
        __builtin_amdgcn_sched_barrier(0);  // the requests stay BEHIND phase 1: hoisted above it they meet the slab's live accumulators
        if (res_buf) {
          const size_t rimg = (size_t)b * p.OH * p.OW * p.rpitch;
          const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<a16_t*>(p.res + rimg), 0, (int)p.res_bytes, 0x00020000);
          const int oyb = oy0 + wm * MT + half * HT;
          const unsigned vo = (unsigned)((((oyb * p.OW + ox0 + lane / CPR) * p.rpitch) + p.roff + ct * TN + wn * NT * 32 + (lane % CPR) * 8) * 2);
#pragma unroll
          for (int it = 0; it < NIT; ++it) {
            constexpr int RPI = 64 / CPR;                       // slab rows per iteration
            const int dy = (it * RPI) / 32, dx = (it * RPI) % 32;
            const unsigned o = oyb + dy < p.OH ? vo + (unsigned)((dy * p.OW + dx) * p.rpitch * 2) : 0x80000000u;
            rvv[it] = __builtin_amdgcn_raw_buffer_load_b128(rr, o, 0, 0);
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int idx = lane + 64 * it;
        const int row = idx / CPR, ch = idx % CPR;
        const int oy = oy0 + wm * MT + half * HT + row / 32, ox = ox0 + (row & 31);
        const int co = ct * TN + wn * NT * 32 + ch * 8;
        if (oy < p.OH && ox < p.OW && co < p.Cout) {
          u32x4 v = *reinterpret_cast<const u32x4*>(slab + row * ROWB + ch * 16);
          const size_t pix = KS == 2 ? ((size_t)b * (2 * p.OH) + 2 * oy + pa) * (size_t)(2 * p.OW) + 2 * ox + pb
                                     : ((size_t)b * p.OH + oy) * p.OW + ox;
          if (p.res) {
            u32x4 rv;
            if (res_buf) rv = rvv[RES_AHEAD ? it : 0];
            else rv = *reinterpret_cast<const u32x4*>(p.res + pix * p.rpitch + p.roff + co);
            if (asel.expk) {            // one uniform branch per 8-element group
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = pack_a2(act_exp(alo(v[e]) + alo(rv[e]), asel.expk), act_exp(ahi(v[e]) + ahi(rv[e]), asel.expk));
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = pack_a2(act_cheap(alo(v[e]) + alo(rv[e]), asel.relu), act_cheap(ahi(v[e]) + ahi(rv[e]), asel.relu));
            }
          }
          *reinterpret_cast<u32x4*>(reinterpret_cast<a16_t*>(p.out) + pix * p.opitch + p.ooff + co) = v;
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
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    });
    if (p.gn_part) {  // lanes with the same channel chunk are CPR apart: fold them, one lane per chunk writes
#pragma unroll
      for (int o = CPR; o < 64; o <<= 1) {
        gs0 += __shfl_xor(gs0, o, 64); gq0 += __shfl_xor(gq0, o, 64);
        gs1 += __shfl_xor(gs1, o, 64); gq1 += __shfl_xor(gq1, o, 64);
      }
      const int co = ct * TN + wn * NT * 32 + lane * 8;
      if (lane < CPR && co < p.Cout) {
        // parts live on the 8-row x 32-col grid whatever the tile: (8-row block, column tile, wave row within the block)
        const int row0 = oy0 + wm * MT;
        const int part = ((phase * ((p.OH + 7) / 8) + row0 / 8) * p.tiles_x + tx) * 2 + ((row0 / 4) & 1);
        if (part < p.gn_nparts) {
          float* dst = p.gn_part + (((size_t)b * p.gn_nparts + part) * (p.Cout / 4) + co / 4) * 2;
          dst[0] = gs0; dst[1] = gq0; dst[2] = gs1; dst[3] = gq1;
        }
      }
    }
    return;
  }

  // ---- planar fp32 output through LDS (the DCN's offset / mask-logit planes, conv_offset: deform_conv.py:357-364).  In the C/D
  // layout a lane holds ONE output channel, so a direct planar store is 64 scattered 4-B words per instruction (32 planes x 2); via
  // a wave-private fp32 slab (one 32-pixel tile row at a time) each lane stores 4 consecutive pixels of one plane and 8 consecutive
  // lanes cover the 128 contiguous bytes of the row.  128 -> 108 at 8 x 420 x 620: 1.31 -> 0.83 ms; 256 -> 108 at half size: 0.45 -> 0.32 ms.
  if constexpr (EPI == EPI_PLANAR) {
    static_assert(!HILO && KS == 3 && STRIDE == 1 && NW == 4, "the planar slab epilogue exists for the 3x3 stride-1 4-wave kernels");
    {
      constexpr int ROWD = NT * 32 + 1;                 // slab row pitch in dwords (odd: column reads spread over the banks)
      static_assert(NW * 32 * ROWD * 4 <= (2 * A_SLOTS + 2 * B_CHUNKS) * 16, "planar epilogue slab fits the pipeline LDS");
      typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
      __syncthreads();  // every wave is done reading the pipeline buffers
      float* slab = reinterpret_cast<float*>(smem) + wave * (32 * ROWD);
      const int lane = fresh_lane_id();
      const int ncol = lane & 31, rhalf = lane >> 5;
      const ActSel asel = act_sel(p.act);
      static_for<MT>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        static_for<NT>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          const int co = ct * TN + (wn * NT + j) * 32 + ncol;
          const float bv = (p.bias && co < p.Cout) ? p.bias[co] : 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = (r & 3) + 8 * (r >> 2) + 4 * rhalf;
            slab[m * ROWD + j * 32 + ncol] = act_cheap(acc[i][j][r] + bv, asel.relu);
          }
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int oy = oy0 + wm * MT + i;
#pragma unroll
        for (int it = 0; it < 8 * NT * 32 / 64; ++it) {
          const int idx = lane + 64 * it;
          const int q = idx & 7, cl = idx >> 3;                  // 4-pixel group of the tile row, channel within the wave's NT*32
          const int co = ct * TN + wn * NT * 32 + cl, x0 = ox0 + 4 * q;
          if (oy < p.OH && co < p.Cout && x0 < p.OW) {
            const f32x4u v = {slab[(4 * q) * ROWD + cl], slab[(4 * q + 1) * ROWD + cl], slab[(4 * q + 2) * ROWD + cl],
                              slab[(4 * q + 3) * ROWD + cl]};
            float* dst = reinterpret_cast<float*>(p.out) + ((size_t)b * p.opitch + p.ooff + co) * (size_t)p.plane_pitch + (size_t)oy * p.OW + x0;
            if (x0 + 3 < p.OW) {
              *reinterpret_cast<f32x4u*>(dst) = v;
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (x0 + e < p.OW) dst[e] = v[e];
            }
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      });
      return;
    }
  }

  // ---- general epilogue: C/D layout of the 32x32 tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  if constexpr (EPI == EPI_GENERAL) {
  const int lane = fresh_lane_id();
  const int ncol = lane & 31, rhalf = lane >> 5;
  // static_for: the accumulator indices must be compile-time constants (a runtime-indexed
  // ext_vector array is demoted to scratch memory)
  const ActSel asel = act_sel(p.act);   // the general epilogue (odd pitches, fp32 NHWC, 16-bit planes): not a hot path, a branch per element is fine
  static_for<NT>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    const int co = ct * TN + (wn * NT + j) * 32 + ncol;
    const bool co_ok = co < p.Cout;
    const float bv = (co_ok && p.bias) ? p.bias[co] : 0.f;
    static_for<MT>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const int oy = oy0 + wm * MT + i;
      const bool row_ok = oy < p.OH;
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int xb = ox0 + 8 * rq + 4 * rhalf;  // 4 consecutive x: xb .. xb+3
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][rq * 4 + e] + bv;
        const size_t pix0 = ((size_t)b * p.OH + oy) * p.OW + xb;
        if (p.out_mode == GLARE_OUT_NHWC_BF16 || p.out_mode == GLARE_OUT_NHWC_F32) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (row_ok && co_ok && xb + e < p.OW) {
              float y = v[e];
              if (p.res) y += a2f(p.res[(pix0 + e) * p.rpitch + p.roff + co]);
              y = act_any(y, asel);
              if (p.out_mode == GLARE_OUT_NHWC_BF16)
                reinterpret_cast<a16_t*>(p.out)[(pix0 + e) * p.opitch + p.ooff + co] = f2a(y);
              else
                reinterpret_cast<float*>(p.out)[(pix0 + e) * p.opitch + p.ooff + co] = y;
            }
          }
        } else {  // planar: [b][co][plane_pitch], pixel index y*OW + x
          if (co_ok && row_ok) {
            const size_t base = ((size_t)b * p.opitch + p.ooff + co) * (size_t)p.plane_pitch + (size_t)oy * p.OW + xb;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if (xb + e < p.OW) {
                const float y = act_any(v[e], asel);
                if (p.out_mode == GLARE_OUT_PLANAR_F32)
                  reinterpret_cast<float*>(p.out)[base + e] = y;
                else
                  reinterpret_cast<a16_t*>(p.out)[base + e] = f2a(y);
              }
            }
          }
        }
      }
    });
  });
  }
}

template <int KS, int STRIDE, int MT, int NT, int WM, int WN, int KSTEPS, bool HILO = false, bool GNP = false, int EPI = EPI_FAST>
int launch(const ConvParams& p_in, hipStream_t stream) {
  using G = TileGeom<KS, STRIDE, WM * MT>;
  constexpr int TN = WN * NT * 32;
  ConvParams p = p_in;
  p.tiles_y = cdiv(p.OH, WM * MT);
  const long long nb = (long long)p.B * p.tiles_x * p.tiles_y * p.co_tiles * (KS == 2 ? 4 : 1);
  if (nb > 0x7fffffffLL) return GLARE_ERR_INVALID;
  p.n_blocks = (int)nb;
  p.gn_nparts = p.tiles_x * cdiv(p.OH, 8) * 2 * (KS == 2 ? 4 : 1);  // the 8 x 32 grid, 2 wave rows (4 rows each) per block
  if (p.gn_part && !(p.fast_epilogue && p.Cout % 32 == 0)) return GLARE_ERR_UNSUPPORTED;
  if (p.gn_part && !HILO && EPI != EPI_FAST) return GLARE_ERR_UNSUPPORTED;   // the fused statistics live in the slab epilogues
  const size_t lds = (size_t)(2 * (((KSTEPS * 2 * G::NPOS + 63) / 64) * 64) + 2 * KS * KSTEPS * 2 * TN) * 16 + (GNP ? (size_t)p.Cin0 * 8 : 0);
  constexpr bool PLANAR_KERNEL = !HILO && KS == 3 && STRIDE == 1 && WM * WN == 4;
  if (!HILO && conv_pick_epilogue(p, PLANAR_KERNEL) != EPI) return GLARE_ERR_INVALID;   // the dispatcher picked the wrong instantiation
  auto kern = conv_igemm_kernel<KS, STRIDE, MT, NT, WM, WN, KSTEPS, HILO, GNP, EPI>;
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return GLARE_ERR_LAUNCH;
  hipLaunchKernelGGL(kern, dim3(p.n_blocks, p.groups > 1 ? p.groups : 1), dim3(64 * WM * WN), lds, stream, p);
  return glare_launch_status();
}

}  // namespace
