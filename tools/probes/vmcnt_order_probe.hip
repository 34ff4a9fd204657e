// Do vector-memory LOADS of one wave always return in issue order on gfx950?  s_waitcnt vmcnt(N) ("all but the N youngest
// have landed"), as the compiler inserts it, is only a valid wait for an OLDER load if they do.
// Each lane issues load A (older) and load B (younger) back to back, waits vmcnt(1) and reads A's destination register,
// which was preset to a sentinel.  A sentinel still there = B completed before A.
//   mode 0: A scattered over a large cold buffer (one descriptor), B dense from a small hot buffer (another descriptor)
//   mode 1: A dense hot, B scattered cold          mode 2: both scattered cold          mode 3: like 0, through global_load
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/probes/vmcnt_order_probe.hip -o /tmp/vp && /tmp/vp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32x4 make_rsrc(const void* p, unsigned bytes) {
  const unsigned long long a = (unsigned long long)p;
  u32x4 r = {(unsigned)a, (unsigned)(a >> 32) & 0xffffu, bytes, 0x00020000u};
  r[0] = __builtin_amdgcn_readfirstlane(r[0]); r[1] = __builtin_amdgcn_readfirstlane(r[1]);
  r[2] = __builtin_amdgcn_readfirstlane(r[2]); r[3] = __builtin_amdgcn_readfirstlane(r[3]);
  return r;
}

template <int MODE>
__global__ __launch_bounds__(256) void probe(const unsigned* cold, unsigned cold_bytes, const unsigned* hot, unsigned hot_bytes,
                                             unsigned long long* viol, int iters, unsigned seed, unsigned* sink) {
  const u32x4 rc = make_rsrc(cold, cold_bytes), rh = make_rsrc(hot, hot_bytes);
  const unsigned lane = threadIdx.x & 63, gid = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned state = gid * 2654435761u + seed, bad = 0, acc = 0;
  for (int it = 0; it < iters; ++it) {
    state = state * 1664525u + 1013904223u;
    const unsigned sc = ((state >> 3) % (cold_bytes / 4)) * 4u;             // scattered, cold
    const unsigned sc2 = (((state * 40503u) >> 3) % (cold_bytes / 4)) * 4u;
    const unsigned dh = (((unsigned)it * 64u + lane) * 4u) % hot_bytes;     // dense, hot
    unsigned a, b, chk;
    if (MODE == 3) {
      const unsigned* pa = cold + sc / 4;
      const unsigned* pb = hot + dh / 4;
      asm volatile("v_mov_b32 %0, 0xdeadbeef\n\tglobal_load_dword %0, %3, off\n\tglobal_load_dword %1, %4, off\n\t"
                   "s_waitcnt vmcnt(1)\n\tv_mov_b32 %2, %0\n\ts_waitcnt vmcnt(0)"
                   : "=&v"(a), "=&v"(b), "=&v"(chk) : "v"(pa), "v"(pb) : "memory");
    } else {
      const unsigned va = MODE == 1 ? dh : sc, vb = MODE == 0 ? dh : (MODE == 1 ? sc : sc2);
      const u32x4 ra = MODE == 1 ? rh : rc, rb = MODE == 0 ? rh : rc;
      asm volatile("v_mov_b32 %0, 0xdeadbeef\n\tbuffer_load_dword %0, %3, %5, 0 offen\n\tbuffer_load_dword %1, %4, %6, 0 offen\n\t"
                   "s_waitcnt vmcnt(1)\n\tv_mov_b32 %2, %0\n\ts_waitcnt vmcnt(0)"
                   : "=&v"(a), "=&v"(b), "=&v"(chk) : "v"(va), "v"(vb), "s"(ra), "s"(rb) : "memory");
    }
    bad += chk == 0xdeadbeefu;
    acc ^= a ^ b;
  }
  if (bad) atomicAdd(viol, (unsigned long long)bad);
  if (acc == 0x12345u) sink[0] = acc;
}

// Compiler-scheduled variant with 16-B loads and position-dependent data (cold[i] = i, hot[i] = i | 2^31), so a register
// consumed before its load landed shows the PREVIOUS iteration's value.  Four loads A (older) then four loads B (younger);
// A is checked first (the compiler waits vmcnt(4) for it), B feeds MFMAs afterwards.
//   mode 0: A scattered/cold, B dense/hot      mode 1: A dense/hot, B scattered/cold
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void fill(unsigned* p, size_t n, unsigned orv) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (unsigned)i | orv;
}
template <int MODE>
__global__ __launch_bounds__(256) void probe4(const unsigned* cold, unsigned cold_bytes, const unsigned* hot, unsigned hot_bytes,
                                              unsigned long long* viol, int iters, unsigned seed, float* sink) {
  const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(cold), 0, (int)cold_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(hot), 0, (int)hot_bytes, 0x00020000);
  const unsigned lane = threadIdx.x & 63, gid = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned state = gid * 2654435761u + seed, bad = 0;
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int it = 0; it < iters; ++it) {
    unsigned sc[4], dh[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      state = state * 1664525u + 1013904223u;
      sc[k] = ((state >> 4) % (cold_bytes / 16)) * 16u;
      dh[k] = ((((unsigned)it * 4u + k) * 64u + lane) * 16u) % hot_bytes;
    }
    u32x4 a[4], b[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] = MODE == 0 ? __builtin_amdgcn_raw_buffer_load_b128(rc, sc[k], 0, 0) : __builtin_amdgcn_raw_buffer_load_b128(rh, dh[k], 0, 0);
#pragma unroll
    for (int k = 0; k < 4; ++k) b[k] = MODE == 0 ? __builtin_amdgcn_raw_buffer_load_b128(rh, dh[k], 0, 0) : __builtin_amdgcn_raw_buffer_load_b128(rc, sc[k], 0, 0);
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned want = MODE == 0 ? (sc[k] / 4 + j) : ((dh[k] / 4 + j) | 0x80000000u);
        bad += a[k][j] != want;
      }
#pragma unroll
    for (int k = 0; k < 4; k += 2)
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b[k]), __builtin_bit_cast(bf16x8, b[k + 1]), acc, 0, 0, 0);
  }
  if (bad) atomicAdd(viol, (unsigned long long)bad);
  if (acc[0] == 12345.f) sink[0] = acc[1];
}

// Destination register == address register (what the compiler emits for the DCN corner gathers): scattered 4-B and 16-B
// loads whose vdata overlaps vaddr, data = dword index, full wait, every lane checked.
__global__ __launch_bounds__(256) void probe_overlap(const unsigned* cold, unsigned cold_bytes, unsigned long long* viol, int iters,
                                                     unsigned seed) {
  const u32x4 rc = make_rsrc(cold, cold_bytes);
  const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned state = gid * 2654435761u + seed, bad = 0;
  for (int it = 0; it < iters; ++it) {
    state = state * 1664525u + 1013904223u;
    const unsigned off = ((state >> 4) % (cold_bytes / 16)) * 16u;
    unsigned a = off;
    asm volatile("buffer_load_dword %0, %0, %1, 0 offen\n\ts_waitcnt vmcnt(0)" : "+v"(a) : "s"(rc) : "memory");
    bad += a != off / 4;
    u32x4 q = {off, 0u, 0u, 0u};
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen\n\ts_waitcnt vmcnt(0)" : "+v"(q) : "v"(off), "s"(rc) : "memory");
    bad += (q[0] != off / 4) + (q[3] != off / 4 + 3);
  }
  if (bad) atomicAdd(viol, (unsigned long long)bad);
}

int main() {
  const unsigned cold_bytes = 1u << 30, hot_bytes = 64u << 10;
  unsigned *cold, *hot, *sink;
  unsigned long long* viol;
  hipMalloc(&cold, cold_bytes); hipMalloc(&hot, hot_bytes); hipMalloc(&viol, 8); hipMalloc(&sink, 4);
  hipMemset(cold, 0x01, cold_bytes); hipMemset(hot, 0x02, hot_bytes);
  const int blocks = 256 * 8, iters = 2000;
  for (int mode = 0; mode < 4; ++mode) {
    hipMemset(viol, 0, 8);
    switch (mode) {
      case 0: hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 0, 0, cold, cold_bytes, hot, hot_bytes, viol, iters, 17u, sink); break;
      case 1: hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, cold, cold_bytes, hot, hot_bytes, viol, iters, 17u, sink); break;
      case 2: hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(256), 0, 0, cold, cold_bytes, hot, hot_bytes, viol, iters, 17u, sink); break;
      default: hipLaunchKernelGGL(probe<3>, dim3(blocks), dim3(256), 0, 0, cold, cold_bytes, hot, hot_bytes, viol, iters, 17u, sink); break;
    }
    unsigned long long v = 0;
    hipMemcpy(&v, viol, 8, hipMemcpyDeviceToHost);
    printf("mode %d: %llu of %llu lane-loads saw the OLDER load's register unwritten after vmcnt(1)  (%s)\n", mode, v,
           (unsigned long long)blocks * 256 * iters, hipGetErrorString(hipGetLastError()));
  }
  hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, cold, (size_t)cold_bytes / 4, 0u);
  hipLaunchKernelGGL(fill, dim3(64), dim3(256), 0, 0, hot, (size_t)hot_bytes / 4, 0x80000000u);
  for (int mode = 0; mode < 2; ++mode) {
    hipMemset(viol, 0, 8);
    if (mode == 0) hipLaunchKernelGGL(probe4<0>, dim3(blocks), dim3(256), 0, 0, cold, cold_bytes, hot, hot_bytes, viol, iters, 29u, (float*)sink);
    else hipLaunchKernelGGL(probe4<1>, dim3(blocks), dim3(256), 0, 0, cold, cold_bytes, hot, hot_bytes, viol, iters, 29u, (float*)sink);
    unsigned long long v = 0;
    hipMemcpy(&v, viol, 8, hipMemcpyDeviceToHost);
    printf("x4 mode %d: %llu of %llu dwords of the OLDER loads were wrong when read behind the compiler's counted wait  (%s)\n", mode, v,
           (unsigned long long)blocks * 256 * iters * 16, hipGetErrorString(hipGetLastError()));
  }
  hipMemset(viol, 0, 8);
  hipLaunchKernelGGL(probe_overlap, dim3(blocks), dim3(256), 0, 0, cold, cold_bytes, viol, iters, 31u);
  {
    unsigned long long v = 0;
    hipMemcpy(&v, viol, 8, hipMemcpyDeviceToHost);
    printf("vdata overlapping vaddr: %llu mismatches  (%s)\n", v, hipGetErrorString(hipGetLastError()));
  }
  return 0;
}
