// Energy per launch of one kernel configuration: back-to-back launches for a few seconds while a host thread samples the
// card's hwmon power1_input.  The chip runs the real pipeline AT its 1400 W cap, so energy per launch -- not time per
// launch -- is what a kernel variant contributes to end-to-end speed.
//   usage: kenergy <qkv|o|ffn1|ffn2|attn|cross> <variant> [seconds=3]
//   build: hipcc --offload-arch=gfx950 -O2 tools/kenergy.hip -Iinclude -Llonglive_amd -llonglive_hip -Wl,-rpath,'$ORIGIN/../longlive_amd' -lpthread -o tools/kenergy
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <glob.h>
#include <string>
#include <thread>
#include <vector>
#include "longlive_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define LL(x) do { int r_ = (x); if (r_) { printf("ll error %d: %s\n", r_, ll_last_error()); exit(1); } } while (0)

static std::vector<std::string> power_files() {
  std::vector<std::string> out;
  glob_t g;
  if (!glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input", 0, nullptr, &g))
    for (size_t i = 0; i < g.gl_pathc; ++i) out.push_back(g.gl_pathv[i]);
  globfree(&g);
  return out;
}
static double read_w(const std::string& f) {
  FILE* fp = fopen(f.c_str(), "r");
  if (!fp) return 0;
  double v = 0;
  if (fscanf(fp, "%lf", &v) != 1) v = 0;
  fclose(fp);
  return v * 1e-6;
}

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
  if (argc < 3) { printf("usage: kenergy <qkv|o|ffn1|ffn2|attn|cross> <variant> [seconds]\n"); return 1; }
  const char* what = argv[1];
  int variant = atoi(argv[2]);
  double secs = argc > 3 ? atof(argv[3]) : 3.0;
  auto files = power_files();
  std::vector<double> idle;
  for (auto& f : files) idle.push_back(read_w(f));

  // LL_TUNING=key=value,...: the same kernel A/B switches bench.py accepts
  if (const char* tun = getenv("LL_TUNING")) {
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
  int M = 4680, N = 0, K = 0, epi = 0;
  bool attn = !strcmp(what, "attn") || !strcmp(what, "cross");
  if (!strcmp(what, "qkv")) { N = 4608; K = 1536; }
  else if (!strcmp(what, "o")) { N = 1536; K = 1536; epi = 2; }
  else if (!strcmp(what, "ffn1")) { N = 8960; K = 1536; epi = 1; }
  else if (!strcmp(what, "ffn2")) { N = 1536; K = 8960; epi = 2; }
  else if (!attn) { printf("unknown kernel %s\n", what); return 1; }
  hipStream_t s = 0;
  std::function<void()> fn;
  double flops = 0;
  if (attn) {
    int Lk = !strcmp(what, "attn") ? 18720 : 512, H = 12;
    LL(ll_set_tuning("attn_variant", variant));
    auto* q = (ll_bf16*)dalloc((size_t)M * H * 128 * 2, 1.0f);
    auto* k = (ll_bf16*)dalloc((size_t)Lk * H * 128 * 2, 1.0f);
    auto* v = (ll_bf16*)dalloc((size_t)Lk * H * 128 * 2, 0.7f);
    auto* o = (ll_bf16*)dalloc((size_t)M * H * 128 * 2, 0.f);
    float scale = 1.0f / sqrtf(128.f);
    fn = [=]() { LL(ll_flash_attn(q, k, v, o, 1, M, H, H * 128, H * 128, H * 128, (long long)Lk * H * 128, 0, Lk, 0, 0, scale, s)); };
    flops = 4.0 * M * Lk * H * 128;
  } else {
    if (variant > 0) LL(ll_set_tuning("gemm_variant", variant));
    auto* x = (ll_bf16*)dalloc((size_t)M * K * 2, 1.0f);
    auto* w = (ll_bf16*)dalloc((size_t)N * K * 2, 1.0f / sqrtf((float)K));
    auto* b = (ll_bf16*)dalloc((size_t)N * 2, 0.1f);
    auto* out = (ll_bf16*)dalloc((size_t)M * N * 2, 0.f);
    auto* res = (ll_bf16*)dalloc((size_t)M * N * 2, 1.0f);
    auto* e = (ll_bf16*)dalloc((size_t)3 * 6 * N * 2, 0.5f);
    auto* mod = (ll_bf16*)dalloc((size_t)6 * N * 2, 0.1f);
    fn = [=]() { LL(ll_gemm_bf16(x, w, b, out, M, N, K, K, N, epi, res, e, getenv("KENERGY_MOD") ? mod : nullptr, 6, 2, M, M / 3, s)); };
    flops = 2.0 * M * N * K;
  }
  for (int i = 0; i < 10; ++i) fn();
  CK(hipDeviceSynchronize());
  std::atomic<bool> stop{false};
  std::vector<std::vector<double>> samples(files.size());
  std::thread sampler([&]() {
    while (!stop.load()) {
      for (size_t i = 0; i < files.size(); ++i) samples[i].push_back(read_w(files[i]));
      std::this_thread::sleep_for(std::chrono::milliseconds(20));
    }
  });
  auto t0 = std::chrono::steady_clock::now();
  long launches = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
    for (int i = 0; i < 50; ++i) fn();
    CK(hipDeviceSynchronize());
    launches += 50;
  }
  double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  stop.store(true);
  sampler.join();
  // our card = the one whose power rose most; average its samples over the second half of the run
  size_t best = 0; double rise = -1e9;
  std::vector<double> avg(files.size(), 0);
  for (size_t i = 0; i < files.size(); ++i) {
    size_t n = samples[i].size(), lo = n / 2;
    double a = 0; for (size_t j = lo; j < n; ++j) a += samples[i][j];
    avg[i] = n > lo ? a / (n - lo) : 0;
    if (avg[i] - idle[i] > rise) { rise = avg[i] - idle[i]; best = i; }
  }
  double us = el / launches * 1e6;
  printf("{\"kernel\": \"%s\", \"variant\": %d, \"us_per_launch\": %.1f, \"tflops\": %.1f, \"avg_power_w\": %.0f, \"idle_w\": %.0f, "
         "\"mj_per_launch\": %.2f, \"pj_per_flop\": %.3f}\n", what, variant, us, flops / us * 1e-6, avg[best], idle[best],
         avg[best] * us * 1e-3, avg[best] * us * 1e-6 / flops * 1e12);
  return 0;
}
