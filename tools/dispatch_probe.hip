// How fast does the hardware start the workgroups of a one-wave-per-SIMD kernel?  (every MFMA launch of the
// layer pays ~4.5 us between its first and its last workgroup's entry.)  A kernel of 228 workgroups x 256 threads stamps
// s_memrealtime (100 MHz) at entry, then idles ~20 us so that no workgroup leaves before the last one has started; variants differ in
// the resources a workgroup claims: dynamic LDS bytes and the register file (512 = VGPRs + AGPRs of a whole SIMD, as the generated
// kernels; 128).  Prints first-to-last entry spread (median of the launches) per variant.
//   hipcc --offload-arch=gfx950 -O2 -o tools/dispatch_probe tools/dispatch_probe.hip && ./tools/dispatch_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x)                                                                              \
  do {                                                                                     \
    hipError_t e_ = (x);                                                                   \
    if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; }     \
  } while (0)

template <bool BIG>
__global__ __launch_bounds__(256) void probe(unsigned long long* t, int spin, int use_lds) {
  extern __shared__ char lds[];
  unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (BIG) asm volatile("v_mov_b32 v255, 0\n v_accvgpr_write_b32 a255, 0" ::: "v255", "a255");
  if (threadIdx.x == 0) {
    if (use_lds) lds[0] = 1;
    t[blockIdx.x] = t0;
  }
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(8);
}

int main() {
  const int G = 228, reps = 21;
  unsigned long long* d;
  CK(hipMalloc(&d, G * sizeof(unsigned long long)));
  CK(hipFuncSetAttribute((const void*)probe<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void*)probe<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  std::vector<unsigned long long> h(G);
  for (int big = 0; big < 2; ++big)
    for (int kb : {0, 32, 64, 96, 128, 148, 160}) {
      std::vector<double> spreads, p50;
      for (int r = 0; r < reps; ++r) {
        if (big) hipLaunchKernelGGL(probe<true>, dim3(G), dim3(256), kb * 1024, 0, d, 2000, kb > 0);
        else hipLaunchKernelGGL(probe<false>, dim3(G), dim3(256), kb * 1024, 0, d, 2000, kb > 0);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), d, G * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        std::vector<unsigned long long> s(h);
        std::sort(s.begin(), s.end());
        spreads.push_back((s.back() - s.front()) * 0.01);
        p50.push_back((s[G / 2] - s.front()) * 0.01);
      }
      std::sort(spreads.begin(), spreads.end());
      std::sort(p50.begin(), p50.end());
      printf("regs %3d  LDS %3d KiB : first -> last entry %5.2f us (median of %d launches; min %5.2f), first -> median workgroup %5.2f us\n",
             big ? 512 : 128, kb, spreads[reps / 2], reps, spreads.front(), p50[reps / 2]);
    }
  return 0;
}
