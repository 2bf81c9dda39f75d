// Does an XCD run faster when fewer of its CUs are busy?  (profiles/r04_xcd_clocks.md: the eight XCDs of a device hold clocks 6-10 %
// apart, the odd ones lowest, and every single-round launch of this path -- 228 workgroups on 256 CUs -- ends with the slowest XCD.)
// A synthetic one-workgroup-per-CU MFMA kernel (512 registers, 128 KiB LDS, the shape of the generated kernels) gives every ACTIVE
// workgroup the same fixed number of MFMAs and stamps s_memrealtime at both ends, with its XCC id.  Variants differ in WHERE the 28
// idle CUs sit: spread evenly (what the hardware's round-robin does to a 228-workgroup grid), or concentrated on chosen XCDs
// (grid 256, workgroup w runs on XCD w % 8; it is active iff w / 8 < n[w % 8]).  Each variant is looped for ~1.5 s; printed: the
// launch time by HIP events and, per XCD, the active workgroups and their mean duration.
//   hipcc --offload-arch=gfx950 -O2 -o tools/xcd_balance_probe tools/xcd_balance_probe.hip && ./tools/xcd_balance_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

#define CK(x)                                                                              \
  do {                                                                                     \
    hipError_t e_ = (x);                                                                   \
    if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; }     \
  } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

struct Stamp {
  unsigned long long t0, t1;
  unsigned xcc, active;
};

__global__ __launch_bounds__(256) void probe(Stamp* st, const int* n_per_xcd, int iters, float* sink) {
  extern __shared__ char lds[];
  const int w = blockIdx.x;
  unsigned xcc = __builtin_amdgcn_s_getreg((20 /* HW_REG_XCC_ID */) | (0 << 6) | (3 << 11)) & 0xf;
  const bool active = (w >> 3) < n_per_xcd[w & 7];
  unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  asm volatile("v_mov_b32 v255, 0\n v_accvgpr_write_b32 a255, 0" ::: "v255", "a255");      // claim the whole register file
  f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  if (active) {
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * (threadIdx.x + j)); b[j] = (__bf16)(0.002f * (threadIdx.x ^ j)); }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    lds[0] = 1;
    st[w] = Stamp{t0, t1, xcc, active ? 1u : 0u};
  }
  if (active && c0[0] + c1[1] + c2[2] + c3[3] == 12345.678f) sink[threadIdx.x] = c0[3];      // keep the accumulators alive
}

// ---- round 5: what DYNAMIC work stealing between XCDs could buy, measured before building it ----
// Every unit (one per owner workgroup, 228 of them) has `iters` steps of 32 MFMAs.  Its first (1 - f) share is the owner's; the
// tail share f of ALL units sits in one device-wide pool of chunks of `chunk` steps, handed out by an atomic counter to whichever
// workgroup asks next -- owners that have finished their own share (fast XCDs ask earlier and more often) and, with G = 256,
// 28 helper workgroups that own nothing.  No partial results, no merge: the UPPER bound of any such scheme.  Total MFMA work is
// the same in every variant.
__global__ __launch_bounds__(256) void probe_steal(Stamp* st, int owners, int main_iters, int chunk, int nchunks, unsigned* counter,
                                                    unsigned* taken, float* sink) {
  extern __shared__ char lds[];
  const int w = blockIdx.x;
  unsigned xcc = __builtin_amdgcn_s_getreg((20 /* HW_REG_XCC_ID */) | (0 << 6) | (3 << 11)) & 0xf;
  unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  asm volatile("v_mov_b32 v255, 0\n v_accvgpr_write_b32 a255, 0" ::: "v255", "a255");
  f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * (threadIdx.x + j)); b[j] = (__bf16)(0.002f * (threadIdx.x ^ j)); }
  auto steps = [&](int n) {
    for (int i = 0; i < n; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
      }
    }
  };
  if (w < owners) steps(main_iters);
  unsigned got = 0;
  volatile unsigned* slot = (volatile unsigned*)lds;
  for (;;) {                                          // every wave of the workgroup works on the chunk thread 0 claimed
    __syncthreads();
    if (threadIdx.x == 0) *slot = atomicAdd(counter, 1u);
    __syncthreads();
    if (*slot >= (unsigned)nchunks) break;            // pool empty: every workgroup reaches this
    steps(chunk);
    ++got;
  }
  unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    st[w] = Stamp{t0, t1, xcc, 1u};
    taken[w] = got;
  }
  if (c0[0] + c1[1] + c2[2] + c3[3] == 12345.678f) sink[threadIdx.x] = c0[3];
}

static int steal_section(Stamp* d, float* sink) {
  const int iters = 2200, owners = 228, launches = 600;
  unsigned *counters, *taken;
  CK(hipMalloc(&counters, (launches + 200) * sizeof(unsigned)));
  CK(hipMalloc(&taken, 256 * sizeof(unsigned)));
  CK(hipFuncSetAttribute((const void*)probe_steal, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  struct V { const char* name; int G; int tail_pct; int chunk; };
  const V vs[] = {
      {"static: 228 owners, nothing pooled", 228, 0, 32},
      {"228 workgroups, 6 % of every unit pooled in chunks of 1.5 %", 228, 6, 33},
      {"228 workgroups, 12 % pooled in chunks of 1.5 %", 228, 12, 33},
      {"256 workgroups (28 helpers), 12 % pooled in chunks of 1.5 %", 256, 12, 33},
      {"256 workgroups (28 helpers), 18 % pooled in chunks of 1.5 %", 256, 18, 33},
      {"256 workgroups (28 helpers), 12 % pooled in chunks of 3 %", 256, 12, 66},
      {"static again", 228, 0, 32},
  };
  printf("\nround 5 -- dynamic stealing, upper bound (no merge, no partial results): %d steps of 32 MFMAs per unit, 228 units\n", iters);
  std::vector<Stamp> h(256);
  std::vector<unsigned> tk(256);
  for (int rep = 0; rep < 2; ++rep)
    for (const V& v : vs) {
      int per_unit_tail = iters * v.tail_pct / 100 / v.chunk * v.chunk;         // whole chunks
      int main_iters = iters - per_unit_tail, nchunks = owners * (per_unit_tail / v.chunk);
      CK(hipMemset(counters, 0, (launches + 200) * sizeof(unsigned)));
      for (int i = 0; i < 200; ++i)
        hipLaunchKernelGGL(probe_steal, dim3(v.G), dim3(256), 128 * 1024, 0, d, owners, main_iters, v.chunk, nchunks, counters + launches + i, taken, sink);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < launches; ++i)
        hipLaunchKernelGGL(probe_steal, dim3(v.G), dim3(256), 128 * 1024, 0, d, owners, main_iters, v.chunk, nchunks, counters + i, taken, sink);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      CK(hipMemcpy(h.data(), d, v.G * sizeof(Stamp), hipMemcpyDeviceToHost));
      CK(hipMemcpy(tk.data(), taken, v.G * sizeof(unsigned), hipMemcpyDeviceToHost));
      double sum[16] = {0}, tks[16] = {0};
      int cnt[16] = {0};
      unsigned long long first = ~0ull, last = 0;
      for (int w = 0; w < v.G; ++w) {
        sum[h[w].xcc] += (h[w].t1 - h[w].t0) * 0.01, tks[h[w].xcc] += tk[w], cnt[h[w].xcc]++;
        first = std::min(first, h[w].t0), last = std::max(last, h[w].t1);
      }
      printf("%-64s: launch %7.1f us (events, mean of %d); last launch first-entry -> last-exit %7.1f us\n   per XCD  mean us / chunks taken per workgroup:",
             v.name, 1e3 * ms / launches, launches, (last - first) * 0.01);
      for (int x = 0; x < 8; ++x) printf("  %d: %6.1f / %4.1f", x, cnt[x] ? sum[x] / cnt[x] : 0.0, cnt[x] ? tks[x] / cnt[x] : 0.0);
      printf("\n");
    }
  return 0;
}

int main(int argc, char** argv) {
  const int G = 256, iters = 2200;      // 2200 x 32 MFMAs x 32 cycles ~ 2.25 M cycles ~ 1.2 ms per launch at 1.9 GHz
  Stamp* d;
  int* dn;
  float* sink;
  CK(hipMalloc(&d, G * sizeof(Stamp)));
  CK(hipMalloc(&dn, 8 * sizeof(int)));
  CK(hipMalloc(&sink, 256 * sizeof(float)));
  CK(hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  if (argc > 1 && !strcmp(argv[1], "steal")) return steal_section(d, sink);
  struct Var { const char* name; int n[8]; };
  const Var vars[] = {
      {"even spread (29,29,29,29,28,28,28,28): a 228-workgroup grid", {29, 29, 29, 29, 28, 28, 28, 28}},
      {"idle CUs on XCD 1 and 3 (18 each)", {32, 18, 32, 18, 32, 32, 32, 32}},
      {"idle CUs on the odd XCDs (25 each)", {32, 25, 32, 25, 32, 25, 32, 25}},
      {"idle CUs on XCD 1 (4 active)", {32, 4, 32, 32, 32, 32, 32, 32}},
      {"all 256", {32, 32, 32, 32, 32, 32, 32, 32}},
      {"even spread again", {29, 29, 29, 29, 28, 28, 28, 28}},
  };
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<Stamp> h(G);
  for (const Var& v : vars) {
    CK(hipMemcpy(dn, v.n, sizeof v.n, hipMemcpyHostToDevice));
    const int launches = 1200;
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(probe, dim3(G), dim3(256), 128 * 1024, 0, d, dn, iters, sink);      // settle the clocks
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(probe, dim3(G), dim3(256), 128 * 1024, 0, d, dn, iters, sink);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(h.data(), d, G * sizeof(Stamp), hipMemcpyDeviceToHost));
    double sum[16] = {0}, mx[16] = {0};
    int cnt[16] = {0}, total = 0, mism = 0;
    for (int w = 0; w < G; ++w) {
      if ((int)h[w].xcc != (w & 7)) ++mism;
      if (!h[w].active) continue;
      double us = (h[w].t1 - h[w].t0) * 0.01;
      sum[h[w].xcc] += us, cnt[h[w].xcc]++, total++;
      mx[h[w].xcc] = std::max(mx[h[w].xcc], us);
    }
    printf("%-62s: %4d active, launch %7.1f us (events, mean of %d)%s\n   per XCD  active / mean us of a workgroup:", v.name, total,
           1e3 * ms / launches, launches, mism ? "  [XCC id != blockIdx % 8 for some workgroups]" : "");
    for (int x = 0; x < 8; ++x) printf("  %d: %2d / %6.1f", x, cnt[x], cnt[x] ? sum[x] / cnt[x] : 0.0);
    printf("\n");
  }
  // greedy levelling: move one active CU from the XCD whose workgroups take longest to the one whose take least, 30 times
  printf("\ngreedy levelling from the even spread (300 launches per step):\n");
  int n[8] = {29, 29, 29, 29, 28, 28, 28, 28};
  double best = 1e30;
  int bestn[8];
  for (int step = 0; step < 30; ++step) {
    CK(hipMemcpy(dn, n, sizeof n, hipMemcpyHostToDevice));
    for (int i = 0; i < 60; ++i) hipLaunchKernelGGL(probe, dim3(G), dim3(256), 128 * 1024, 0, d, dn, iters, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 300; ++i) hipLaunchKernelGGL(probe, dim3(G), dim3(256), 128 * 1024, 0, d, dn, iters, sink);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(h.data(), d, G * sizeof(Stamp), hipMemcpyDeviceToHost));
    double sum[8] = {0};
    int cnt[8] = {0};
    for (int w = 0; w < G; ++w)
      if (h[w].active) sum[w & 7] += (h[w].t1 - h[w].t0) * 0.01, cnt[w & 7]++;
    double us = 1e3 * ms / 300;
    if (us < best) { best = us; for (int x = 0; x < 8; ++x) bestn[x] = n[x]; }
    printf("  step %2d  n = %2d %2d %2d %2d %2d %2d %2d %2d  launch %7.1f us   per-XCD mean:", step, n[0], n[1], n[2], n[3], n[4], n[5], n[6], n[7], us);
    int hi = -1, lo = -1;
    for (int x = 0; x < 8; ++x) {
      double m = cnt[x] ? sum[x] / cnt[x] : 0;
      printf(" %6.1f", m);
      if (cnt[x] > 1 && (hi < 0 || m > sum[hi] / cnt[hi])) hi = x;
      if (n[x] < 32 && cnt[x] && (lo < 0 || m < sum[lo] / cnt[lo])) lo = x;
    }
    printf("\n");
    if (hi < 0 || lo < 0 || hi == lo) break;
    n[hi]--, n[lo]++;
  }
  printf("best: %.1f us with n = %d %d %d %d %d %d %d %d\n", best, bestn[0], bestn[1], bestn[2], bestn[3], bestn[4], bestn[5], bestn[6], bestn[7]);
  return 0;
}
