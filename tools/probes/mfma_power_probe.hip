// What dense MFMA rate can this chip SUSTAIN, as a function of the operand values?  A register-only loop (no LDS, no memory): every
// wave holds 2 A and 4 B fragments and 8 independent 32x32 accumulators and issues acc[i][j] += A[i] . B[j] back to back, so that
// consecutive MFMAs see different operand pairs (as in the real kernels).  Each configuration runs for a few seconds while a host
// thread samples the shader clock and socket power from the amdgpu hwmon files.  Configurations: fp16 / bf16, 32x32x16 / 16x16x32,
// operands random-normal / all zero / one side zero, 1-4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/mfma_power_probe.hip -o /tmp/mfma_power_probe -lpthread && /tmp/mfma_power_probe [seconds]
// The answer (MI355X, round 4) is in DESIGN.md section 3 ("the power wall") and profiles/r04_mfma_power.txt.
#include <hip/hip_runtime.h>
#include <dirent.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

// KIND 0: v_mfma_f32_32x32x16_f16   1: v_mfma_f32_32x32x16_bf16   2: v_mfma_f32_16x16x32_f16
// ORDER (32x32 kinds): 0 = i outer / j inner (B changes every MFMA, A every 4th); 1 = snake (exactly one operand changes per MFMA);
// 2 = both operands change on every MFMA.  Same 8 products per iteration in all three.
template <int KIND, int ORDER = 0>
__global__ __launch_bounds__(256) void mfma_loop(const u32x4* __restrict__ frag, int iters, float* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // 6 fragments per wave, different per wave and lane: [block % 61][wave][6][64 lanes]
  const u32x4* f = frag + ((size_t)((blockIdx.x % 61) * 4 + wave) * 6) * 64 + lane;
  u32x4 a[2], b[4];
  a[0] = f[0]; a[1] = f[64];
#pragma unroll
  for (int j = 0; j < 4; ++j) b[j] = f[(2 + j) * 64];
  if constexpr (KIND == 2) {
    f32x4 acc[2][4][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h) acc[i][j][h] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int h = 0; h < 2; ++h)   // two 16x16x32 MFMAs = the FLOPs of one 32x32x16
            acc[i][j][h] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[i]), __builtin_bit_cast(f16x8, b[j]), acc[i][j][h], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h) s += acc[i][j][h][0] + acc[i][j][h][3];
    if (s == 12345.678f) sink[0] = s;
  } else {
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    constexpr int SEQ[3][8][2] = {{{0, 0}, {0, 1}, {0, 2}, {0, 3}, {1, 0}, {1, 1}, {1, 2}, {1, 3}},
                                  {{0, 0}, {0, 1}, {0, 2}, {0, 3}, {1, 3}, {1, 2}, {1, 1}, {1, 0}},
                                  {{0, 0}, {1, 1}, {0, 2}, {1, 3}, {0, 1}, {1, 0}, {0, 3}, {1, 2}}};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
          constexpr int dummy = 0; (void)dummy;
          const int i = SEQ[ORDER][q][0], j = SEQ[ORDER][q][1];
          {
          if constexpr (KIND == 0)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[i]), __builtin_bit_cast(f16x8, b[j]), acc[i][j], 0, 0, 0);
          else
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]), acc[i][j], 0, 0, 0);
          }
      }
      __builtin_amdgcn_sched_barrier(0);     // the issue order is the experiment
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][15];
    if (s == 12345.678f) sink[0] = s;
  }
}

// ---- host -------------------------------------------------------------------------------------------------------------------------
static std::vector<std::string> hwmon_dirs() {
  std::vector<std::string> out;
  for (int c = 0; c < 128; ++c) {
    const std::string base = "/sys/class/drm/card" + std::to_string(c) + "/device/hwmon";
    DIR* d = opendir(base.c_str());
    if (!d) continue;
    while (dirent* e = readdir(d)) {
      if (strncmp(e->d_name, "hwmon", 5)) continue;
      const std::string h = base + "/" + e->d_name;
      FILE* f = fopen((h + "/freq1_input").c_str(), "r");
      if (f) { fclose(f); out.push_back(h); }
    }
    closedir(d);
  }
  return out;
}
static double read_num(const std::string& path) {
  FILE* f = fopen(path.c_str(), "r");
  if (!f) return 0;
  double v = 0;
  if (fscanf(f, "%lf", &v) != 1) v = 0;
  fclose(f);
  return v;
}
static double median(std::vector<double> v) {
  if (v.empty()) return 0;
  std::sort(v.begin(), v.end());
  return v[v.size() / 2];
}

static uint16_t f2h(float x) { _Float16 h = (_Float16)x; uint16_t u; memcpy(&u, &h, 2); return u; }
static uint16_t f2b(float x) { uint32_t u; memcpy(&u, &x, 4); return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1)) >> 16); }
static float randn() {   // Box-Muller on rand()
  const float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = (rand() + 1.0f) / (RAND_MAX + 2.0f);
  return sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
}

int main(int argc, char** argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 3.0;
  auto dirs = hwmon_dirs();
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  {   // a box may show the hwmon files of GPUs that belong to other tenants: keep the device this process computes on
    char bdf[64] = {0};
    if (hipDeviceGetPCIBusId(bdf, sizeof(bdf), 0) == hipSuccess) {
      std::vector<std::string> mine;
      for (const auto& d : dirs) {
        char real[4096];
        if (realpath((d + "/../..").c_str(), real) && strcasestr(real, bdf)) mine.push_back(d);
      }
      if (!mine.empty()) dirs = mine;
    }
  }
  const int cus = prop.multiProcessorCount;
  const size_t n_frag = (size_t)61 * 4 * 6 * 64;              // u32x4 elements
  u32x4* d_frag; float* sink;
  hipMalloc(&d_frag, n_frag * 16); hipMalloc(&sink, 4);
  std::vector<uint16_t> host(n_frag * 8);
  printf("%d CUs, %zu hwmon dir(s); %.1f s per configuration (first 40 %% of the samples dropped)\n", cus, dirs.size(), secs);
  printf("%-34s %-10s %5s  %9s %7s  %9s %8s\n", "instruction", "operands", "w/SIMD", "TFLOP/s", "of 2500", "sclk MHz", "socket W");
  struct Cfg { int kind; const char* name; };
  const Cfg kinds[3] = {{0, "v_mfma_f32_32x32x16_f16"}, {1, "v_mfma_f32_32x32x16_bf16"}, {2, "v_mfma_f32_16x16x32_f16 (x2)"}};
  const char* dnames[4] = {"random", "all zero", "A zero", "B zero"};
  for (int kind = 0; kind < 3; ++kind)
    for (int data = 0; data < 4; ++data)
      for (int wps = 4; wps >= 1; wps >>= 1) {
        if ((data >= 2 || wps == 2) && kind != 0) continue;     // the one-sided cases and the 2-wave point only for the fp16 32x32 form
        if (data == 1 && wps != 4) continue;
        srand(1234);
        for (size_t e = 0; e < n_frag * 8; ++e) {
          const size_t fi = (e / (8 * 64)) % 6;                  // fragment index within the wave: 0, 1 = A; 2..5 = B
          const bool zero = data == 1 || (data == 2 && fi < 2) || (data == 3 && fi >= 2);
          const float v = zero ? 0.f : randn();
          host[e] = kind == 1 ? f2b(v) : f2h(v);
        }
        hipMemcpy(d_frag, host.data(), n_frag * 16, hipMemcpyHostToDevice);
        const int blocks = cus * wps, iters = 20000;             // 8 MFMAs (32x32x16-equivalents) per iteration and wave
        const double flop = (double)blocks * 4 * iters * 8 * 32768.0;
        auto launch = [&]() {
          if (kind == 0) hipLaunchKernelGGL(mfma_loop<0>, dim3(blocks), dim3(256), 0, 0, d_frag, iters, sink);
          else if (kind == 1) hipLaunchKernelGGL(mfma_loop<1>, dim3(blocks), dim3(256), 0, 0, d_frag, iters, sink);
          else hipLaunchKernelGGL(mfma_loop<2>, dim3(blocks), dim3(256), 0, 0, d_frag, iters, sink);
        };
        launch(); hipDeviceSynchronize();
        std::atomic<bool> stop{false};
        std::vector<std::vector<double>> clk(dirs.size()), pw(dirs.size());
        std::thread th([&]() {
          while (!stop.load()) {
            for (size_t i = 0; i < dirs.size(); ++i) {
              clk[i].push_back(read_num(dirs[i] + "/freq1_input") / 1e6);
              pw[i].push_back(read_num(dirs[i] + "/power1_input") / 1e6);
            }
            std::this_thread::sleep_for(std::chrono::milliseconds(20));
          }
        });
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const auto t0 = std::chrono::steady_clock::now();
        int n = 0; float ms_tail = 0; int n_tail = 0;
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
          hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
          float ms; hipEventElapsedTime(&ms, e0, e1);
          ++n;
          if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 0.4 * secs) { ms_tail += ms; ++n_tail; }
        }
        stop.store(true); th.join();
        double best_c = 0, best_p = 0;
        for (size_t i = 0; i < dirs.size(); ++i) {
          std::vector<double> c(clk[i].begin() + clk[i].size() * 2 / 5, clk[i].end()), w(pw[i].begin() + pw[i].size() * 2 / 5, pw[i].end());
          if (median(w) > best_p) { best_p = median(w); best_c = median(c); }
        }
        const double tf = flop / (ms_tail / std::max(n_tail, 1)) / 1e9;
        printf("%-34s %-10s %5d  %9.0f %7.3f  %9.0f %8.0f\n", kinds[kind].name, dnames[data], wps, tf, tf / 2500.0, best_c, best_p);
        fflush(stdout);
        std::this_thread::sleep_for(std::chrono::milliseconds(800));
      }
  // operand ORDER at fixed data (fp16 32x32x16, random, 1 wave / SIMD): how much of the power is operand delivery?
  for (int order = 0; order < 3; ++order) {
    srand(1234);
    for (size_t e = 0; e < n_frag * 8; ++e) host[e] = f2h(randn());
    hipMemcpy(d_frag, host.data(), n_frag * 16, hipMemcpyHostToDevice);
    const int blocks = cus, iters = 20000;
    const double flop = (double)blocks * 4 * iters * 8 * 32768.0;
    auto launch = [&]() {
      if (order == 0) hipLaunchKernelGGL((mfma_loop<0, 0>), dim3(blocks), dim3(256), 0, 0, d_frag, iters, sink);
      else if (order == 1) hipLaunchKernelGGL((mfma_loop<0, 1>), dim3(blocks), dim3(256), 0, 0, d_frag, iters, sink);
      else hipLaunchKernelGGL((mfma_loop<0, 2>), dim3(blocks), dim3(256), 0, 0, d_frag, iters, sink);
    };
    launch(); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const auto t0 = std::chrono::steady_clock::now();
    float ms_tail = 0; int n_tail = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 0.4 * secs) { ms_tail += ms; ++n_tail; }
    }
    const char* on[3] = {"i outer, j inner (10 operand changes / 8)", "snake (8 / 8: one operand per MFMA)", "both change every MFMA (16 / 8)"};
    printf("order: %-44s %9.0f TFLOP/s\n", on[order], flop / (ms_tail / std::max(n_tail, 1)) / 1e9);
    fflush(stdout);
    std::this_thread::sleep_for(std::chrono::milliseconds(800));
  }
  return 0;
}
