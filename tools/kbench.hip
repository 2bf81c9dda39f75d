// Standalone kernel micro-benchmark for liblonglive_hip.so (no Python, no torch): times the MFMA kernels at the real
// LongLive-1.3B shapes with HIP events and spot-checks a sample of outputs against a host fp64 reference.
//   hipcc --offload-arch=gfx950 -O2 tools/kbench.hip -Iinclude -Llonglive_amd -llonglive_hip -Wl,-rpath,$PWD/longlive_amd -o tools/kbench
//   tools/kbench [gemm|attn|all] [iters]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <functional>
#include <string>
#include <vector>

#include "longlive_hip.h"

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));    \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)
#define LL(x)                                                       \
  do {                                                              \
    int r_ = (x);                                                   \
    if (r_ != 0) {                                                  \
      fprintf(stderr, "%s -> %d: %s\n", #x, r_, ll_last_error());   \
      exit(1);                                                      \
    }                                                               \
  } while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static inline uint64_t rnd() {
  uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static inline float gauss() {  // Irwin-Hall(12)
  float s = 0;
  for (int i = 0; i < 3; ++i) {
    uint64_t h = rnd();
    for (int k = 0; k < 4; ++k) s += (float)((h >> (16 * k)) & 0xFFFF);
  }
  return (s - 393210.f) / 65536.f;
}
static inline uint16_t f2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7FFF + ((u >> 16) & 1);
  return (uint16_t)(u >> 16);
}
static inline float bf2f(uint16_t b) {
  uint32_t u = (uint32_t)b << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

struct Buf {
  std::vector<uint16_t> h;
  uint16_t* d = nullptr;
  Buf(size_t n, float scale) : h(n) {
    for (size_t i = 0; i < n; ++i) h[i] = f2bf(gauss() * scale);
    CK(hipMalloc(&d, n * 2));
    CK(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice));
  }
  void pull() { CK(hipMemcpy(h.data(), d, h.size() * 2, hipMemcpyDeviceToHost)); }
  ~Buf() { hipFree(d); }
};

static double time_ms(hipStream_t s, int iters, const std::function<void()>& fn) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) fn();
  CK(hipStreamSynchronize(s));
  CK(hipEventRecord(a, s));
  for (int i = 0; i < iters; ++i) fn();
  CK(hipEventRecord(b, s));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms / iters;
}

static void bench_gemm(const char* name, int M, int N, int K, int epi, int iters) {
  Buf x((size_t)M * K, 1.0f), w((size_t)N * K, 1.0f / sqrtf((float)K)), bias(N, 0.1f), out((size_t)M * N, 0.f);
  Buf res((size_t)M * N, 1.0f), e((size_t)3 * 6 * N, 0.5f), mod((size_t)6 * N, 0.1f);
  hipStream_t s = 0;
  auto fn = [&]() {
    LL(ll_gemm_bf16(x.d, w.d, bias.d, out.d, M, N, K, K, N, epi, res.d, e.d, getenv("KBENCH_MOD") ? mod.d : nullptr, 6, 2, M, M / 3, s));   // as shipped: e is the modulation table (mod = NULL)
  };
  double ms = time_ms(s, iters, fn);
  // spot check (epilogue 0 and 1 only: plain value check)
  out.pull();
  double worst = 0;
  if (epi == LL_EPI_BIAS) {
    for (int t = 0; t < 64; ++t) {
      int m = (int)(rnd() % M), n = (int)(rnd() % N);
      if (t < 4) { m = M - 1 - t; n = N - 1 - t; }
      double acc = bf2f(bias.h[n]);
      for (int k = 0; k < K; ++k) acc += (double)bf2f(x.h[(size_t)m * K + k]) * bf2f(w.h[(size_t)n * K + k]);
      double err = fabs(acc - bf2f(out.h[(size_t)m * N + n])) / (fabs(acc) * 0.0079 + 2e-3);
      if (err > worst) worst = err;
    }
  }
  printf("gemm %-10s M=%5d N=%5d K=%5d epi=%d : %8.1f us  %7.1f TFLOP/s   check(ulp-ish)=%.2f %s\n", name, M, N, K, epi,
         ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) / 1e12, worst, worst > 1.0 ? "**FAIL**" : "");
}

static void bench_gemm_i8(const char* name, int M, int N, int K, int iters) {
  Buf x((size_t)M * K, 1.0f), w((size_t)N * K, 1.0f / sqrtf((float)K)), bias(N, 0.1f), out((size_t)M * N, 0.f);
  int8_t *xq, *wq; float *sx, *sw;
  CK(hipMalloc(&xq, (size_t)M * K)); CK(hipMalloc(&wq, (size_t)N * K)); CK(hipMalloc(&sx, M * 4)); CK(hipMalloc(&sw, N * 4));
  hipStream_t s = 0;
  LL(ll_quantize_rows(x.d, xq, sx, M, K, K, s));
  LL(ll_quantize_rows(w.d, wq, sw, N, K, K, s));
  double qms = time_ms(s, iters, [&]() { LL(ll_quantize_rows(x.d, xq, sx, M, K, K, s)); });
  auto fn = [&]() { LL(ll_gemm_w8a8(xq, sx, wq, sw, bias.d, out.d, M, N, K, N, LL_EPI_BIAS, nullptr, nullptr, nullptr, 0, 0, 0, 0, s)); };
  double ms = time_ms(s, iters, fn);
  out.pull();
  std::vector<int8_t> hx((size_t)M * K), hw((size_t)N * K);
  std::vector<float> hsx(M), hsw(N);
  CK(hipMemcpy(hx.data(), xq, hx.size(), hipMemcpyDeviceToHost)); CK(hipMemcpy(hw.data(), wq, hw.size(), hipMemcpyDeviceToHost));
  CK(hipMemcpy(hsx.data(), sx, M * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hsw.data(), sw, N * 4, hipMemcpyDeviceToHost));
  double worst = 0, worst_vs_bf16 = 0;
  for (int t = 0; t < 48; ++t) {
    int m = (int)(rnd() % M), n = (int)(rnd() % N);
    if (t < 4) { m = M - 1 - t; n = N - 1 - t; }
    long long acc = 0;
    double ref = bf2f(bias.h[n]);
    for (int k = 0; k < K; ++k) {
      acc += (long long)hx[(size_t)m * K + k] * hw[(size_t)n * K + k];
      ref += (double)bf2f(x.h[(size_t)m * K + k]) * bf2f(w.h[(size_t)n * K + k]);
    }
    double want = (double)acc * ((double)hsx[m] * hsw[n]) + bf2f(bias.h[n]);
    double got = bf2f(out.h[(size_t)m * N + n]);
    double err = fabs(want - got) / (fabs(want) * 0.0079 + 2e-3);
    if (err > worst) worst = err;
    double e2 = fabs(ref - got);
    if (e2 > worst_vs_bf16) worst_vs_bf16 = e2;
  }
  printf("w8a8 %-10s M=%5d N=%5d K=%5d       : %8.1f us  %7.1f TOP/s  (quantize x: %.1f us)  exact-int check=%.2f %s  max|err vs fp|=%.3f\n",
         name, M, N, K, ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) / 1e12, qms * 1e3, worst, worst > 1.0 ? "**FAIL**" : "", worst_vs_bf16);
  hipFree(xq); hipFree(wq); hipFree(sx); hipFree(sw);
}

static void bench_attn(const char* name, int Lq, int H, int Sk, int n0, int iters) {
  Buf q((size_t)Lq * H * 128, 1.0f), k((size_t)Sk * H * 128, 1.0f), v((size_t)Sk * H * 128, 0.7f), o((size_t)Lq * H * 128, 0.f);
  hipStream_t s = 0;
  float scale = 1.0f / sqrtf(128.f);
  auto fn = [&]() {
    LL(ll_flash_attn(q.d, k.d, v.d, o.d, 1, Lq, H, H * 128, H * 128, H * 128, (long long)Sk * H * 128, 0, n0, 0, 0, scale, s));
  };
  double ms = time_ms(s, iters, fn);
  o.pull();
  double worst = 0;
  for (int t = 0; t < 6; ++t) {
    int qi = (t < 2) ? Lq - 1 - t : (int)(rnd() % Lq), hh = (int)(rnd() % H);
    std::vector<double> sc(n0);
    double mx = -1e300;
    for (int j = 0; j < n0; ++j) {
      double a = 0;
      for (int d = 0; d < 128; ++d) a += (double)bf2f(q.h[((size_t)qi * H + hh) * 128 + d]) * bf2f(k.h[((size_t)j * H + hh) * 128 + d]);
      sc[j] = a * scale;
      if (sc[j] > mx) mx = sc[j];
    }
    double l = 0;
    for (int j = 0; j < n0; ++j) { sc[j] = exp(sc[j] - mx); l += sc[j]; }
    for (int d = 0; d < 128; d += (getenv("KBENCH_DEBUG") ? 1 : 17)) {
      double a = 0;
      for (int j = 0; j < n0; ++j) a += sc[j] * bf2f(v.h[((size_t)j * H + hh) * 128 + d]);
      a /= l;
      double got = bf2f(o.h[((size_t)qi * H + hh) * 128 + d]);
      double err = fabs(a - got);
      if (err > worst) worst = err;
      if (getenv("KBENCH_DEBUG") && t < 3) printf("   q=%d d=%3d want %+.4f got %+.4f %s\n", qi, d, a, got, err > 1e-2 ? "<<<" : "");
    }
  }
  printf("attn %-10s Lq=%5d H=%2d Lk=%5d      : %8.1f us  %7.1f TFLOP/s   max|err|=%.2e %s\n", name, Lq, H, n0, ms * 1e3,
         4.0 * Lq * (double)n0 * 128 * H / (ms * 1e-3) / 1e12, worst, worst > 1e-2 ? "**FAIL**" : "");
}


// ---- "shipped": exactly the launches of ONE steady-state DiT layer under the default tuning, one timing line each plus a
// machine-readable "#WORK" record (kernel-name substring, grid, algorithmic FLOPs / bytes, us) that tools/pmc_summary.py
// joins with the rocprofv3 --pmc counters of the same command.
static void work_line(const char* tag, const char* match, double flop, double bytes, double us, const char* bound) {
  printf("#WORK {\"tag\": \"%s\", \"match\": \"%s\", \"flop\": %.6g, \"bytes\": %.6g, \"us\": %.3f, \"bound\": \"%s\"}\n", tag, match,
         flop, bytes, us, bound);
}

static void bench_shipped(int iters) {
  const int L = 4680, C = 1536, F1 = 8960, S = 18720, H = 12, FS = 1560;
  hipStream_t s = 0;
  char plan[256];
  // ---- MFMA kernels
  struct G { const char* tag; int N, K, epi; } gs[] = {{"gemm_qkv", 3 * C, C, LL_EPI_BIAS}, {"gemm_o", C, C, LL_EPI_BIAS_GATE_RES},
      {"gemm_cq", C, C, LL_EPI_BIAS}, {"gemm_co", C, C, LL_EPI_BIAS_RES}, {"gemm_f1", F1, C, LL_EPI_BIAS_GELU},
      {"gemm_f2", C, F1, LL_EPI_BIAS_GATE_RES}};
  Buf vcache((size_t)S * C, 1.0f);
  for (auto& g : gs) {
    Buf x((size_t)L * g.K, 1.0f), w((size_t)g.N * g.K, 1.0f / sqrtf((float)g.K)), bias(g.N, 0.1f), out((size_t)L * g.N, 0.f);
    Buf res((size_t)L * g.N, 1.0f), e((size_t)3 * 6 * g.N, 0.5f);
    const bool qkv = !strcmp(g.tag, "gemm_qkv");
    double ms = time_ms(s, iters, [&]() {
      if (qkv)      // as shipped: the V third goes straight into the KV cache
        LL(ll_gemm_bf16_qkv(x.d, w.d, bias.d, out.d, L, g.N, g.K, g.K, g.N, vcache.d, 1, L, S, S - L, 0, L, s));
      else          // as shipped: `e` is a layer's slice of ll_modulation_table (mod = NULL)
        LL(ll_gemm_bf16(x.d, w.d, bias.d, out.d, L, g.N, g.K, g.K, g.N, g.epi, res.d, e.d, nullptr, 6, 2, L, FS, s));
    });
    LL(ll_gemm_plan_epi(L, g.N, g.K, 0, g.epi, qkv ? 2 : 1, plan, sizeof plan));
    double fl = 2.0 * L * g.N * g.K, by = 2.0 * ((double)L * g.K + (double)g.N * g.K + (double)L * g.N * (g.epi >= 2 ? 2 : 1));
    printf("%-10s %-60s %8.1f us %7.1f TFLOP/s\n", g.tag, plan, ms * 1e3, fl / (ms * 1e-3) / 1e12);
    char match[64];
    snprintf(match, sizeof match, "%.14s<%d", plan, g.epi);      // gemm_kernel_vN<EPI
    work_line(g.tag, match, fl, by, ms * 1e3, "mfma");
  }
  {
    Buf q((size_t)L * C, 1.0f), k((size_t)S * C, 1.0f), v((size_t)S * C, 0.7f), o((size_t)L * C, 0.f);
    float scale = 1.0f / sqrtf(128.f);
      double ms = time_ms(s, iters, [&]() { LL(ll_flash_attn(q.d, k.d, v.d, o.d, 1, L, H, C, C, C, (long long)S * C, 0, S, 0, 0, scale, s)); });
    LL(ll_flash_attn_plan(L, H, 1, S, 0, 1, plan, sizeof plan));
    double fl = 4.0 * L * (double)S * 128 * H;
    printf("%-10s %-60.60s %8.1f us %7.1f TFLOP/s\n", "attn_self", plan, ms * 1e3, fl / (ms * 1e-3) / 1e12);
    work_line("flash_attn_self", strstr(plan, "flash_attn_asm") ? "flash_attn_asm_kernel" : "flash_attn_pipe_kernel<8, 1>", fl,
              2.0 * (2.0 * L * C + 2.0 * S * C), ms * 1e3, "mfma");
    Buf kc((size_t)512 * C, 1.0f), vc((size_t)512 * C, 0.7f);
    ms = time_ms(s, iters, [&]() { LL(ll_flash_attn(q.d, kc.d, vc.d, o.d, 1, L, H, C, C, C, (long long)512 * C, 0, 512, 0, 0, scale, s)); });
    fl = 4.0 * L * 512.0 * 128 * H;
    printf("%-10s %-60.60s %8.1f us %7.1f TFLOP/s\n", "attn_cross", "flash_attn_pipe_kernel<8, 0> (512 text keys)", ms * 1e3, fl / (ms * 1e-3) / 1e12);
    work_line("flash_attn_cross", "flash_attn_pipe_kernel<8, 0>", fl, 2.0 * (2.0 * L * C + 2.0 * 512 * C), ms * 1e3, "mfma");
  }
  // ---- row kernels (HBM-bound)
  {
    Buf x((size_t)L * C, 1.7f), out((size_t)L * C, 0.f), e((size_t)3 * 6 * C, 0.5f), mod((size_t)6 * C, 0.1f), w(C, 0.1f), b(C, 0.1f);
    double ms = time_ms(s, iters, [&]() { LL(ll_ln_modulate(x.d, out.d, e.d, nullptr, 6, 0, 1, 1, L, C, 3, 1e-6f, s)); });
    printf("%-10s %-60s %8.1f us %7.1f GB/s\n", "ln_mod", "ln_modulate_kernel", ms * 1e3, 4.0 * L * C / (ms * 1e-3) / 1e9);
    work_line("ln_modulate", "ln_modulate_kernel", 0, 4.0 * L * C, ms * 1e3, "hbm");
    ms = time_ms(s, iters, [&]() { LL(ll_layernorm_affine(x.d, w.d, b.d, out.d, L, C, 1e-6f, s)); });
    printf("%-10s %-60s %8.1f us %7.1f GB/s\n", "ln_affine", "layernorm_affine_kernel", ms * 1e3, 4.0 * L * C / (ms * 1e-3) / 1e9);
    work_line("layernorm_affine", "layernorm_affine_kernel", 0, 4.0 * L * C, ms * 1e3, "hbm");
    ms = time_ms(s, iters, [&]() { LL(ll_rmsnorm(x.d, w.d, out.d, L, C, C, C, 1e-6f, s)); });
    printf("%-10s %-60s %8.1f us %7.1f GB/s\n", "rmsnorm", "rmsnorm_kernel", ms * 1e3, 4.0 * L * C / (ms * 1e-3) / 1e9);
    work_line("rmsnorm", "rmsnorm_kernel", 0, 4.0 * L * C, ms * 1e3, "hbm");
  }
  {
    Buf qkv((size_t)L * 3 * C, 1.3f), wq(C, 0.1f), wk(C, 0.1f), qo((size_t)L * C, 0.f), ck((size_t)S * C, 1.0f), cv((size_t)S * C, 1.0f);
    float *rf, *rhw;
    std::vector<float> hrf((size_t)1024 * 22 * 2), hrhw((size_t)FS * 42 * 2);
    for (auto& t : hrf) t = 0.5f;
    for (auto& t : hrhw) t = 0.5f;
    CK(hipMalloc(&rf, hrf.size() * 4)); CK(hipMalloc(&rhw, hrhw.size() * 4));
    CK(hipMemcpy(rf, hrf.data(), hrf.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(rhw, hrhw.data(), hrhw.size() * 4, hipMemcpyHostToDevice));
    double ms = time_ms(s, iters, [&]() {
      LL(ll_qk_norm_rope_kv_store(qkv.d, wq.d, wk.d, rf, rhw, qo.d, ck.d, nullptr, 1, L, C, 128, FS, 12, S, S - L, 0, L, 1e-6f, s));
    });     // as shipped: cache_v = NULL (V was inserted by the projection): q, k in; q out; k -> cache
    printf("%-10s %-60s %8.1f us %7.1f GB/s\n", "qk_rope_kv", "qk_norm_rope_kv_kernel", ms * 1e3, 8.0 * L * C / (ms * 1e-3) / 1e9);
    work_line("qk_norm_rope_kv_store", "qk_norm_rope_kv_kernel", 0, 8.0 * L * C, ms * 1e3, "hbm");
    ms = time_ms(s, iters, [&]() { LL(ll_kv_roll(ck.d, cv.d, 1, S, C, 3 * FS, 6 * FS, 6 * FS, s)); });
    printf("%-10s %-60s %8.1f us %7.1f GB/s\n", "kv_roll", "copy_rows_kernel (2 launches)", ms * 1e3, 8.0 * 6 * FS * C / (ms * 1e-3) / 1e9);
    work_line("kv_roll", "copy_rows_kernel", 0, 8.0 * 6 * FS * C, ms * 1e3, "hbm");
    hipFree(rf); hipFree(rhw);
  }
}

// ---- "layerseq": the launches of one steady-state DiT layer IN THE MODEL'S ORDER, back to back, `layers` times (buffers
// allocated once).  This is the regime the headline is measured in -- different kernels following each other at the clock the
// pipeline holds -- without Python: rocprofv3 --pmc over bench.py itself dies in the profiler's dispatch hook (profiles/r03_pmc_*),
// so the in-pipeline counter passes (tools/pmc_inpipe.sh) run over this.
static void bench_layerseq(int layers) {
  const int L = 4680, C = 1536, F1 = 8960, S = 18720, H = 12, FS = 1560;
  hipStream_t s = 0;
  Buf xs((size_t)L * C, 1.0f), h((size_t)L * C, 0.f), qkv((size_t)L * 3 * C, 0.f), q((size_t)L * C, 0.f), att((size_t)L * C, 0.f);
  Buf ffh((size_t)L * F1, 0.f), kc((size_t)S * C, 1.0f), vc((size_t)S * C, 0.7f), ck((size_t)512 * C, 1.0f), cv((size_t)512 * C, 0.7f);
  Buf wqkv((size_t)3 * C * C, 0.0255f), bqkv(3 * C, 0.1f), wo((size_t)C * C, 0.0255f), bo(C, 0.1f), wcq((size_t)C * C, 0.0255f), wco((size_t)C * C, 0.0255f);
  Buf w1((size_t)F1 * C, 0.0255f), b1(F1, 0.1f), w2((size_t)C * F1, 0.0106f), b2(C, 0.1f), e((size_t)3 * 6 * C, 0.5f), nw(C, 0.1f), nb(C, 0.1f);
  float *rf, *rhw;
  std::vector<float> hrf((size_t)1024 * 22 * 2, 0.5f), hrhw((size_t)FS * 42 * 2, 0.5f);
  CK(hipMalloc(&rf, hrf.size() * 4)); CK(hipMalloc(&rhw, hrhw.size() * 4));
  CK(hipMemcpy(rf, hrf.data(), hrf.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(rhw, hrhw.data(), hrhw.size() * 4, hipMemcpyHostToDevice));
  const float scale = 1.0f / sqrtf(128.f);
  // TIMING-ONLY experiment switches (what a fusion or a balanced grid could buy, before building it):
  //   KB_SKIP=i,j,...   leave launches i, j, ... (0-based, model order) out of the layer
  //   KB_ATTN_LQ / KB_ATTN_S   geometry of the self-attention launch (e.g. 5376 rows x 17024 keys = 252 workgroups x 266 key tiles:
  //                     the grid a key-split that balances 228 workgroups x 293 tiles over 256 CUs would run, without its merge)
  unsigned skip = 0;
  if (const char* sk = getenv("KB_SKIP")) for (const char* p = sk; *p; ++p) { if (*p >= '0' && *p <= '9') { int v = atoi(p); skip |= 1u << v; while (*p >= '0' && *p <= '9') ++p; if (!*p) break; } }
  const int aLq = getenv("KB_ATTN_LQ") ? atoi(getenv("KB_ATTN_LQ")) : L, aS = getenv("KB_ATTN_S") ? atoi(getenv("KB_ATTN_S")) : S;
  Buf qbig((size_t)(aLq > L ? aLq : 1) * C, 1.0f), obig((size_t)(aLq > L ? aLq : 1) * C, 0.f);
  uint16_t* aq = aLq > L ? qbig.d : q.d;
  uint16_t* ao = aLq > L ? obig.d : att.d;
  if (skip || aLq != L || aS != S) printf("layerseq: TIMING-ONLY experiment (skip mask 0x%x, self-attention %d rows x %d keys)\n", skip, aLq, aS);
  // KB_FUSE_QN=0: the three-launch form of cross-attention's q path (projection, RMSNorm, attention) instead of the shipped two
  const bool fuse_qn = !(getenv("KB_FUSE_QN") && atoi(getenv("KB_FUSE_QN")) == 0) && ll_gemm_ssq_planes(L, C, C) == H && ll_flash_attn_qnorm_ok(H, 512);
  float* ssq; CK(hipMalloc(&ssq, (size_t)H * L * 4)); CK(hipMemset(ssq, 0, (size_t)H * L * 4));
  // KB_TAB32=0: LN + modulate from the bf16 table (ll_ln_modulate) instead of the shipped fp32 one (ll_ln_modulate_tab)
  const bool tab32 = !(getenv("KB_TAB32") && atoi(getenv("KB_TAB32")) == 0);
  float* e32; CK(hipMalloc(&e32, (size_t)3 * 6 * C * 4));
  {
    std::vector<float> he32((size_t)3 * 6 * C);            // what ll_modulation_table_f32 would make of the bf16 table `e`: the two forms
    for (int f = 0; f < 3; ++f)                             // of the launch then produce the same h (same data for the kernels behind them)
      for (int i = 0; i < 6; ++i)
        for (int c = 0; c < C; ++c) {
          size_t k = ((size_t)f * 6 + i) * C + c;
          float v = bf2f(e.h[k]);
          he32[k] = (i == 1 || i == 4) ? bf2f(f2bf(1.0f + v)) : v;
        }
    CK(hipMemcpy(e32, he32.data(), he32.size() * 4, hipMemcpyHostToDevice));
  }
#define ON(i) (!(skip & (1u << (i))))
  auto layer = [&]() {
    if (ON(0)) { if (tab32) LL(ll_ln_modulate_tab(xs.d, h.d, nullptr, nullptr, e32, 6, 0, 1, 1, L, C, 3, 1e-6f, s)); else LL(ll_ln_modulate(xs.d, h.d, e.d, nullptr, 6, 0, 1, 1, L, C, 3, 1e-6f, s)); }
    if (ON(1)) LL(ll_gemm_bf16_qkv(h.d, wqkv.d, bqkv.d, qkv.d, L, 3 * C, C, C, 3 * C, vc.d, 1, L, S, S - L, 0, L, s));
    if (ON(2)) LL(ll_qk_norm_rope_kv_store(qkv.d, nw.d, nw.d, rf, rhw, q.d, kc.d, nullptr, 1, L, C, 128, FS, 12, S, S - L, 0, L, 1e-6f, s));
    if (ON(3)) LL(ll_flash_attn(aq, kc.d, vc.d, ao, 1, aLq, H, C, C, C, (long long)S * C, 0, aS, 0, 0, scale, s));
    if (ON(4)) LL(ll_gemm_bf16(att.d, wo.d, bo.d, xs.d, L, C, C, C, C, LL_EPI_BIAS_GATE_RES, xs.d, e.d, nullptr, 6, 2, L, FS, s));
    if (ON(5)) LL(ll_layernorm_affine(xs.d, nw.d, nb.d, h.d, L, C, 1e-6f, s));
    if (fuse_qn) {      // cross-attention q: RMSNorm statistics from the projection's epilogue, applied in the attention prologue (2 launches)
      if (ON(6)) LL(ll_gemm_bf16_ssq(h.d, wcq.d, bo.d, q.d, ssq, L, C, C, C, C, s));
      if (ON(8)) LL(ll_flash_attn_qnorm(q.d, ssq, nw.d, 1e-6f, ck.d, cv.d, att.d, 1, L, H, C, C, C, (long long)512 * C, 0, 512, scale, s));
    } else {
      if (ON(6)) LL(ll_gemm_bf16(h.d, wcq.d, bo.d, q.d, L, C, C, C, C, LL_EPI_BIAS, nullptr, nullptr, nullptr, 0, 0, 0, 0, s));
      if (ON(7)) LL(ll_rmsnorm(q.d, nw.d, q.d, L, C, C, C, 1e-6f, s));
      if (ON(8)) LL(ll_flash_attn(q.d, ck.d, cv.d, att.d, 1, L, H, C, C, C, (long long)512 * C, 0, 512, 0, 0, scale, s));
    }
    if (ON(9)) LL(ll_gemm_bf16(att.d, wco.d, bo.d, xs.d, L, C, C, C, C, LL_EPI_BIAS_RES, xs.d, nullptr, nullptr, 0, 0, 0, 0, s));
    if (ON(10)) { if (tab32) LL(ll_ln_modulate_tab(xs.d, h.d, nullptr, nullptr, e32, 6, 3, 4, 1, L, C, 3, 1e-6f, s)); else LL(ll_ln_modulate(xs.d, h.d, e.d, nullptr, 6, 3, 4, 1, L, C, 3, 1e-6f, s)); }
    if (ON(11)) LL(ll_gemm_bf16(h.d, w1.d, b1.d, ffh.d, L, F1, C, C, F1, LL_EPI_BIAS_GELU, nullptr, nullptr, nullptr, 0, 0, 0, 0, s));
    if (ON(12)) LL(ll_gemm_bf16(ffh.d, w2.d, b2.d, xs.d, L, C, F1, F1, C, LL_EPI_BIAS_GATE_RES, xs.d, e.d, nullptr, 6, 5, L, FS, s));
  };
#undef ON
  // KB_SPLIT=<rows> (TIMING-ONLY): the layer as TWO row ranges [0, rows) / [rows, L) on two HIP streams that meet only at the
  // self-attention (each half's attention reads the K / V both halves inserted) -- would a token split of ONE prompt stream let one
  // half's row kernels / epilogues overlap the other half's dense kernels?  (frame indices are not kept: results invalid)
  if (const char* sp = getenv("KB_SPLIT")) {
    const int R0 = atoi(sp);
    hipStream_t st[2]; CK(hipStreamCreate(&st[0])); CK(hipStreamCreate(&st[1]));
    hipEvent_t ev[2]; CK(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
    auto half = [&](int hf, int phase) {
      hipStream_t t = st[hf];
      const int r0 = hf ? R0 : 0, M = hf ? L - R0 : R0, fl = M / 2;
      uint16_t *x_ = xs.d + (size_t)r0 * C, *h_ = h.d + (size_t)r0 * C, *qkv_ = qkv.d + (size_t)r0 * 3 * C, *q_ = q.d + (size_t)r0 * C, *att_ = att.d + (size_t)r0 * C;
      uint16_t* ffh_ = ffh.d + (size_t)r0 * F1;
      if (phase == 0) {
        LL(ll_ln_modulate(x_, h_, e.d, nullptr, 6, 0, 1, 1, M, C, 1, 1e-6f, t));
        LL(ll_gemm_bf16_qkv(h_, wqkv.d, bqkv.d, qkv_, M, 3 * C, C, C, 3 * C, vc.d, 1, M, S, S - L + r0, 0, M, t));
        LL(ll_qk_norm_rope_kv_store(qkv_, nw.d, nw.d, rf, rhw, q_, kc.d, nullptr, 1, M, C, 128, fl, 12, S, S - L + r0, 0, M, 1e-6f, t));
        CK(hipEventRecord(ev[hf], t));
        return;
      }
      CK(hipStreamWaitEvent(t, ev[1 - hf], 0));
      LL(ll_flash_attn(q_, kc.d, vc.d, att_, 1, M, H, C, C, C, (long long)S * C, 0, S, 0, 0, scale, t));
      LL(ll_gemm_bf16(att_, wo.d, bo.d, x_, M, C, C, C, C, LL_EPI_BIAS_GATE_RES, x_, e.d, nullptr, 6, 2, M, M, t));
      LL(ll_layernorm_affine(x_, nw.d, nb.d, h_, M, C, 1e-6f, t));
      LL(ll_gemm_bf16_ssq(h_, wcq.d, bo.d, q_, ssq + r0, M, C, C, C, C, t));      // (plane stride M here: timing only)
      LL(ll_flash_attn_qnorm(q_, ssq + r0, nw.d, 1e-6f, ck.d, cv.d, att_, 1, M, H, C, C, C, (long long)512 * C, 0, 512, scale, t));
      LL(ll_gemm_bf16(att_, wco.d, bo.d, x_, M, C, C, C, C, LL_EPI_BIAS_RES, x_, nullptr, nullptr, 0, 0, 0, 0, t));
      LL(ll_ln_modulate(x_, h_, e.d, nullptr, 6, 3, 4, 1, M, C, 1, 1e-6f, t));
      LL(ll_gemm_bf16(h_, w1.d, b1.d, ffh_, M, F1, C, C, F1, LL_EPI_BIAS_GELU, nullptr, nullptr, nullptr, 0, 0, 0, 0, t));
      LL(ll_gemm_bf16(ffh_, w2.d, b2.d, x_, M, C, F1, F1, C, LL_EPI_BIAS_GATE_RES, x_, e.d, nullptr, 6, 5, M, M, t));
    };
    auto layer2 = [&]() { half(0, 0); half(1, 0); half(0, 1); half(1, 1); };
    for (int i = 0; i < 3; ++i) layer2();
    CK(hipDeviceSynchronize());
    hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    CK(hipEventRecord(t0, st[0])); CK(hipStreamWaitEvent(st[1], t0, 0));
    for (int i = 0; i < layers; ++i) layer2();
    CK(hipEventRecord(ev[1], st[1])); CK(hipStreamWaitEvent(st[0], ev[1], 0));
    CK(hipEventRecord(t1, st[0])); CK(hipEventSynchronize(t1));
    float ms2; CK(hipEventElapsedTime(&ms2, t0, t1));
    printf("layerseq: TIMING-ONLY token split %d + %d rows on two streams: %d layers, %.1f us per layer\n", R0, L - R0, layers, ms2 * 1e3 / layers);
    hipFree(rf); hipFree(rhw); hipFree(ssq);
    return;
  }
  for (int i = 0; i < 3; ++i) layer();
  CK(hipStreamSynchronize(s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, s));
  for (int i = 0; i < layers; ++i) layer();
  CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("layerseq: %d layers, %.1f us per layer (%d launches in model order; 150 layers = one AR block: %.1f ms)\n", layers, ms * 1e3 / layers, fuse_qn ? 12 : 13, ms * 150.0 / layers);
  hipFree(rf); hipFree(rhw); hipFree(ssq);
}

int main(int argc, char** argv) {
  if (const char* tun = getenv("LL_TUNING")) {      // LL_TUNING=key=value,...: the kernel A/B switches bench.py accepts
    std::string t(tun);
    size_t pos = 0;
    while (pos < t.size()) {
      size_t c = t.find(',', pos);
      std::string kv = t.substr(pos, c == std::string::npos ? std::string::npos : c - pos);
      size_t eq = kv.find('=');
      if (eq != std::string::npos) LL(ll_set_tuning(kv.substr(0, eq).c_str(), atoi(kv.c_str() + eq + 1)));
      if (c == std::string::npos) break;
      pos = c + 1;
    }
  }
  const char* what = argc > 1 ? argv[1] : "all";
  int iters = argc > 2 ? atoi(argv[2]) : 20;
  bool all = !strcmp(what, "all");
  if (!strcmp(what, "shipped")) {
    bench_shipped(iters);
    return 0;
  }
  if (!strcmp(what, "layerseq")) {
    bench_layerseq(iters);
    return 0;
  }
  if (!strcmp(what, "gemmx")) {      // kbench gemmx <iters> M N K epi [gemm_variant]: one custom shape
    if (argc < 7) { printf("usage: kbench gemmx iters M N K epi [variant]\n"); return 1; }
    if (argc > 7) LL(ll_set_tuning("gemm_variant", atoi(argv[7])));
    bench_gemm("custom", atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), atoi(argv[6]), iters);
    return 0;
  }
  if (all || !strcmp(what, "gemm")) {
   for (int variant = 2; variant <= 6; ++variant) {
    LL(ll_set_tuning("gemm_variant", variant));
    printf("-- gemm_variant %d\n", variant);
    bench_gemm("qkv", 4680, 4608, 1536, LL_EPI_BIAS, iters);
    bench_gemm("o/q/co", 4680, 1536, 1536, LL_EPI_BIAS, iters);
    bench_gemm("o+gate", 4680, 1536, 1536, LL_EPI_BIAS_GATE_RES, iters);
    bench_gemm("ffn1", 4680, 8960, 1536, LL_EPI_BIAS_GELU, iters);
    bench_gemm("ffn2", 4680, 1536, 8960, LL_EPI_BIAS_GATE_RES, iters);
    bench_gemm("ffn2-plain", 4680, 1536, 8960, LL_EPI_BIAS, iters);
    bench_gemm("recache-qkv", 18720, 4608, 1536, LL_EPI_BIAS, iters > 5 ? 5 : iters);
    bench_gemm("sq4096", 4096, 4096, 4096, LL_EPI_BIAS, iters);
    bench_gemm("edge", 300, 136, 128, LL_EPI_BIAS, 2);
   }
   LL(ll_set_tuning("gemm_variant", 0));
   bench_gemm_i8("qkv", 4680, 4608, 1536, iters);
   bench_gemm_i8("o/q/co", 4680, 1536, 1536, iters);
   bench_gemm_i8("ffn1", 4680, 8960, 1536, iters);
   bench_gemm_i8("ffn2", 4680, 1536, 8960, iters);
   bench_gemm_i8("edge", 300, 136, 256, 2);
  }
  if (!strcmp(what, "attnx")) {      // kbench attnx <iters> [rounds]: steady-state self-attention, shipped kernel vs the attn_asm forms, interleaved
    const int rounds = argc > 3 ? atoi(argv[3]) : 3;
    for (int r = 0; r < rounds; ++r)
      for (int form = 0; form <= 2; ++form) {
        LL(ll_set_tuning("attn_asm", form));
        bench_attn(form == 0 ? "self/pipe" : form == 1 ? "self/asm-b" : "self/asm-g", 4680, 12, 18720, 18720, iters);
      }
    LL(ll_set_tuning("attn_asm", 1));
    bench_attn("recache/asm-b", 18720, 12, 18720, 18720, iters > 5 ? 5 : iters);
    LL(ll_set_tuning("attn_asm", 0));
    bench_attn("recache/pipe", 18720, 12, 18720, 18720, iters > 5 ? 5 : iters);
    return 0;
  }
  if (all || !strcmp(what, "attn")) {
   for (int variant = 0; variant <= 2; ++variant) {
    LL(ll_set_tuning("attn_variant", variant));
    if (getenv("KB_ATTN_XCD")) LL(ll_set_tuning("attn_xcd", atoi(getenv("KB_ATTN_XCD"))));
    printf("-- attn_variant %d\n", variant);
    bench_attn("self", 4680, 12, 18720, 18720, iters);
    if (getenv("KBENCH_SELF_ONLY")) continue;          // PMC passes: only the steady-state self-attention launch
    bench_attn("self-b1", 4680, 12, 18720, 9360, iters);
    bench_attn("cross", 4680, 12, 512, 512, iters);
    bench_attn("ragged", 200, 2, 300, 157, 3);
    bench_attn("ragged-long", 700, 3, 1500, 1437, 3);
    bench_attn("long-1tile+", 300, 2, 1100, 1025, 3);
    bench_attn("tiny", 33, 1, 7, 7, 3);
    bench_attn("onetile", 64, 1, 64, 64, 3);
   }
  }
  return 0;
}
