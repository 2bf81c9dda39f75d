// Where do the cycles of a key tile go in the ping-pong attention loop?  Built once per diagnostic variant with the kernel
// source compiled in (not against the shipped library):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DLL_ATTN_SCHED=4 -DLL_ATTN_DIAG=<bits> -Iinclude \
//         tools/attn_diag.hip longlive_amd/csrc/attention.hip longlive_amd/csrc/api.hip -o tools/attn_diag_<bits>
// bits: 1 stamps only | 2 no K/V staging in the loop | 4 no softmax phase | 8 no matrix phase | 16 no wait for the staged tile
// (see attention.hip).  Runs the steady-state self-attention launch (Lq 4680, 18720 keys, 12 heads) back to back for
// <seconds>, then prints the in-kernel clock and the cycles per key tile (median over workgroups) of the LAST launch.
//   usage: attn_diag_<bits> [seconds=2] [lq=4680] [lk=18720]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "longlive_hip.h"

extern "C" int ll_attn_diag_read(unsigned long long* host, int n);

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define LL(x) do { int r_ = (x); if (r_) { printf("ll error %d: %s\n", r_, ll_last_error()); exit(1); } } while (0)

static void* dalloc(size_t bytes, float scale) {
  std::vector<unsigned short> h(bytes / 2);
  unsigned long long s = 0x9E3779B97F4A7C15ull;
  for (auto& x : h) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    float f = ((int)((s >> 40) & 0xFFFF) - 32768) / 32768.0f * scale;
    unsigned u; memcpy(&u, &f, 4);
    x = (unsigned short)(u >> 16);
  }
  void* d; CK(hipMalloc(&d, bytes)); CK(hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice));
  return d;
}

int main(int argc, char** argv) {
  double secs = argc > 1 ? atof(argv[1]) : 2.0;
  int M = argc > 2 ? atoi(argv[2]) : 4680, Lk = argc > 3 ? atoi(argv[3]) : 18720, H = 12;
  auto* q = (ll_bf16*)dalloc((size_t)M * H * 128 * 2, 1.0f);
  auto* k = (ll_bf16*)dalloc((size_t)Lk * H * 128 * 2, 1.0f);
  auto* v = (ll_bf16*)dalloc((size_t)Lk * H * 128 * 2, 0.7f);
  auto* o = (ll_bf16*)dalloc((size_t)M * H * 128 * 2, 0.f);
  float scale = 1.0f / sqrtf(128.f);
  hipStream_t s = 0;
  auto fn = [&]() { LL(ll_flash_attn(q, k, v, o, 1, M, H, H * 128, H * 128, H * 128, (long long)Lk * H * 128, 0, Lk, 0, 0, scale, nullptr, 0, s)); };
  for (int i = 0; i < 5; ++i) fn();
  CK(hipDeviceSynchronize());
  auto t0 = std::chrono::steady_clock::now();
  long launches = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
    for (int i = 0; i < 20; ++i) fn();
    CK(hipDeviceSynchronize());
    launches += 20;
  }
  double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  int nwg = ((M + 255) / 256) * H;
  std::vector<unsigned long long> d(4 * 2048);
  LL(ll_attn_diag_read(d.data(), 4 * 2048));
  std::vector<double> cyc, clk;
  for (int w = 0; w < nwg && w < 2048; ++w) {
    double c = (double)d[4 * w], r = (double)d[4 * w + 1], nt = (double)d[4 * w + 2];
    if (nt < 1 || r < 1) continue;
    cyc.push_back(c / nt);
    clk.push_back(c / r * 0.1);     // s_memrealtime ticks at 100 MHz
  }
  std::sort(cyc.begin(), cyc.end());
  std::sort(clk.begin(), clk.end());
  double us = el / launches * 1e6;
  printf("{\"diag\": %d, \"us_per_launch\": %.1f, \"tflops\": %.1f, \"cycles_per_tile_median\": %.0f, \"cycles_per_tile_min\": %.0f, "
         "\"cycles_per_tile_max\": %.0f, \"clock_ghz_median\": %.3f, \"workgroups\": %zu}\n",
         LL_ATTN_DIAG, us, 4.0 * M * Lk * H * 128 / us * 1e-6, cyc[cyc.size() / 2], cyc.front(), cyc.back(), clk[clk.size() / 2], cyc.size());
  return 0;
}
