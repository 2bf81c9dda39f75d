// Diagnostic build of flash_attn_asm_kernel (never part of the library): the generated body with s_memtime stamps around the two
// phases and the wait + barrier of every tile (gen/attn_asm_gen.py --diag).  Prints, per wave position, the average cycles per tile
// spent in phase A (S = K Q^T || exp / sum / pack), phase B (O += V^T P^T || row max, LDS reads, LDS-DMA) and at the tile's
// counted wait + barrier.  Read SHARES, not lengths: every stamp drains the LDS queue.
//   python3 longlive_amd/csrc/gen/attn_asm_gen.py --diag tools/attn_asm_body_d.inc
//   hipcc --offload-arch=gfx950 -O2 -Iinclude -Ilonglive_amd/csrc tools/attn_asm_diag.hip -o tools/attn_asm_diag
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include "common.h"

int ll_check_launch(const char*) { return hipGetLastError() == hipSuccess ? 0 : 1; }
#define ASM_KT 64
#define LL_ASM_DIAG 1
#define LL_ASM_NAME flash_attn_asm_d_kernel
#define LL_ASM_INC "../../tools/attn_asm_body_d.inc"
#include "attention_asm_kernel.inl"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static void* dalloc(size_t bytes, float scale) {
  std::vector<unsigned short> h(bytes / 2);
  unsigned long long s = 0x9E3779B97F4A7C15ull;
  for (auto& x : h) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    float f = ((int)((s >> 40) & 0xFFFF) - 32768) / 32768.0f * scale * 1.7f;
    unsigned u; memcpy(&u, &f, 4);
    x = (unsigned short)(u >> 16);
  }
  void* d; CK(hipMalloc(&d, bytes)); CK(hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice));
  return d;
}

int main(int argc, char** argv) {
  const int Lq = argc > 1 ? atoi(argv[1]) : 4680, H = 12, Lk = argc > 2 ? atoi(argv[2]) : 18720, C = H * 128;
  const int iters = argc > 3 ? atoi(argv[3]) : 20;
  bf16* q = (bf16*)dalloc((size_t)Lq * C * 2, 1.f);
  bf16* k = (bf16*)dalloc((size_t)Lk * C * 2, 1.f);
  bf16* v = (bf16*)dalloc((size_t)Lk * C * 2, 0.7f);
  bf16* o; CK(hipMalloc(&o, (size_t)Lq * C * 2));
  const int nqt = (Lq + 255) / 256, nwg = nqt * H;
  unsigned long long* dbg; CK(hipMalloc(&dbg, (size_t)nwg * 4 * 64)); CK(hipMemset(dbg, 0, (size_t)nwg * 4 * 64));
  CK(hipFuncSetAttribute((const void*)flash_attn_asm_d_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  const float c = 0.08838834764831845f * 1.4426950408889634f;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int it = 0; it < iters + 3; ++it) {
    if (it == 3) CK(hipEventRecord(e0));
    hipLaunchKernelGGL(flash_attn_asm_d_kernel, dim3(nwg), dim3(256), 128 * 1024, 0, q, k, v, o, Lq, C, C, C, (long long)Lk * C, 0, Lk, c, nqt, 1, dbg);
  }
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const int nt = (Lk + 63) / 64;
#ifdef LL_DIAG_PRO
  // timeline build (ASM_DIAG_PRO=1): eight s_memrealtime stamps (100 MHz) per wave -> where a launch's time goes outside the tile loop
  {
    std::vector<unsigned long long> t((size_t)nwg * 4 * 8);
    CK(hipMemcpy(t.data(), dbg, t.size() * 8, hipMemcpyDeviceToHost));
    const char* names[7] = {"setup + staging issue", "Q loads issued -> Q and first tiles landed", "Q convert (+ norm)", "tile 0 + K(1) + barriers",
                            "tile loop (tiles 1 .. nt-1) + tail", "epilogue (normalise, pack, store issue)", "stores acknowledged"};
    unsigned long long first = ~0ull, last = 0;
    std::vector<double> d[7];
    for (int g = 0; g < nwg * 4; ++g) {
      unsigned long long* r = &t[(size_t)g * 8];
      if (r[7] == 0) continue;                       // idle wave
      if (r[0] < first) first = r[0];
      if (r[7] > last) last = r[7];
      for (int k = 0; k < 7; ++k) d[k].push_back((double)(r[k + 1] - r[k]) * 0.01);
    }
    auto med = [](std::vector<double> x) { std::sort(x.begin(), x.end()); return x[x.size() / 2]; };
    printf("timeline build: %.1f us per launch by events; first wave entry -> last wave done %.1f us; %d workgroups, %d key tiles; per wave (median, us):\n",
           ms * 1e3 / iters, (double)(last - first) * 0.01, nwg, nt);
    double sum = 0;
    for (int k = 0; k < 7; ++k) { printf("  %-46s %6.2f\n", names[k], med(d[k])); sum += med(d[k]); }
    printf("  %-46s %6.2f\n", "sum (one wave's life)", sum);
    return 0;
  }
#endif
  std::vector<unsigned long long> h((size_t)nwg * 16);
  CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
  printf("diag build: %.1f us per launch (stamped: not the kernel's real time), %d workgroups, %d tiles\n", ms * 1e3 / iters, nwg, nt);
  for (int w = 0; w < 4; ++w) {
    std::vector<double> a, b, s, tot;
    for (int g = 0; g < nwg; ++g) {
      unsigned long long* r = &h[(size_t)(g * 4 + w) * 4];
      if (r[3] == 0) continue;                      // idle wave
      a.push_back((double)r[0] / (nt - 1)); b.push_back((double)r[1] / (nt - 1)); s.push_back((double)r[2] / (nt - 1)); tot.push_back((double)r[3] / (nt - 1));
    }
    if (a.empty()) continue;
    auto med = [](std::vector<double> x) { std::sort(x.begin(), x.end()); return x[x.size() / 2]; };
    printf("wave %d (%zu active): cycles per tile  A %.0f  B %.0f  wait+barrier %.0f  | loop %.0f (median over workgroups; 2048 = the MFMAs alone)\n",
           w, a.size(), med(a), med(b), med(s), med(tot));
  }
  return 0;
}
