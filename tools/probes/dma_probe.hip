// Micro-benchmark: sustained global->LDS throughput per CU of (a) LDS-DMA (global_load_lds_dwordx4), (b) plain
// global_load_dwordx4 + ds_write_b128, from an L2-resident source, at 4 / 8 / 12 waves per CU (1 / 2 / 3 workgroups).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/dma_probe.hip -o /tmp/dma_probe && /tmp/dma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

template <int MODE>
__global__ __launch_bounds__(256) void probe(const char* __restrict__ src, size_t src_bytes, int iters, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // each wave streams its own 48 KB window of the (L2-resident) source, 1 KB per instruction
  const char* base = src + (size_t)((blockIdx.x * 4 + wave) % 64) * 49152;   // 3 MB footprint: resident in every XCD L2
  unsigned acc = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      const char* g = base + j * 4096 + lane * 16 + (it & 3) * 1024;
      char* l = smem + wave * 12288 + j * 1024;
      if (MODE == 0) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
      } else {
        const u32x4 v = *reinterpret_cast<const u32x4*>(g);
        *reinterpret_cast<u32x4*>(l + lane * 16) = v;
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    acc += *reinterpret_cast<unsigned*>(smem + wave * 12288 + lane * 4);
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
  const size_t src_bytes = 64u << 20;  // 64 MB: L2 / MALL resident after the first pass
  char* src; unsigned* sink;
  hipMalloc(&src, src_bytes); hipMemset(src, 1, src_bytes); hipMalloc(&sink, 4);
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  const double clk = prop.clockRate * 1e3;
  for (int mode = 0; mode < 2; ++mode)
    for (int wgs = 1; wgs <= 3; ++wgs) {
      const int iters = 2000, blocks = cus * wgs;
      const size_t lds = 49152;  // 3 workgroups of 48 KB fit one CU
      auto k = mode == 0 ? probe<0> : probe<1>;
      hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, 0, src, src_bytes, 50, sink);
      hipEventRecord(a);
      hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, 0, src, src_bytes, iters, sink);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      const double bytes = (double)blocks * 4 * 12 * 1024 * iters;
      printf("%s  %d wg/CU (%2d waves): %.2f TB/s aggregate, %.1f B/clk/CU, %.0f cycles per 1-KB piece per CU\n",
             mode == 0 ? "LDS-DMA        " : "load + ds_write", wgs, wgs * 4, bytes / ms / 1e9, bytes / (ms * 1e-3) / clk / cus,
             1024.0 / (bytes / (ms * 1e-3) / clk / cus));
    }
  return 0;
}
