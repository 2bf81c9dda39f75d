// Energy per FLOP of bare bf16 MFMA loops (operands in registers, random data) for the two shapes, at 1 and 2 waves/SIMD.
//   build: hipcc --offload-arch=gfx950 -O2 tools/mfma_energy.hip -lpthread -o tools/mfma_energy ; usage: mfma_energy [seconds]
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <glob.h>
#include <string>
#include <thread>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int SHAPE>
__global__ __launch_bounds__(512) void mfma_loop(const bf16x8* __restrict__ in, float* __restrict__ out, int iters) {
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = in[(threadIdx.x * 8 + i) & 4095]; b[i] = in[(threadIdx.x * 8 + 4 + i) & 4095]; }
  float r = 0.f;
  if (SHAPE == 32) {
    f32x16 c[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) c[i][j] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[i], c[i], 0, 0, 0);
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) r += c[i][j];
  } else {
    f32x4 c[8];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) c[i][j] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[(i + (i >> 2)) & 3], c[i], 0, 0, 0);
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) r += c[i][j];
  }
  if (r == 123.456f) out[threadIdx.x] = r;
}

static std::vector<std::string> power_files() {
  std::vector<std::string> o; glob_t g;
  if (!glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input", 0, nullptr, &g)) for (size_t i = 0; i < g.gl_pathc; ++i) o.push_back(g.gl_pathv[i]);
  globfree(&g); return o;
}
static double read_w(const std::string& f) { FILE* fp = fopen(f.c_str(), "r"); if (!fp) return 0; double v = 0; if (fscanf(fp, "%lf", &v) != 1) v = 0; fclose(fp); return v * 1e-6; }

template <int SHAPE>
static void run(const char* name, int wg_per_cu_threads, double secs, bf16x8* in, float* out) {
  auto files = power_files();
  std::vector<double> idle; for (auto& f : files) idle.push_back(read_w(f));
  const int iters = 20000, grid = 256;
  auto launch = [&]() { hipLaunchKernelGGL(mfma_loop<SHAPE>, dim3(grid), dim3(wg_per_cu_threads), 0, 0, in, out, iters); };
  launch(); CK(hipDeviceSynchronize());
  std::atomic<bool> stop{false}; std::vector<std::vector<double>> smp(files.size());
  std::thread th([&]() { while (!stop.load()) { for (size_t i = 0; i < files.size(); ++i) smp[i].push_back(read_w(files[i])); std::this_thread::sleep_for(std::chrono::milliseconds(20)); } });
  auto t0 = std::chrono::steady_clock::now(); long n = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) { for (int i = 0; i < 4; ++i) launch(); CK(hipDeviceSynchronize()); n += 4; }
  double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  stop.store(true); th.join();
  size_t best = 0; double rise = -1e9, pw = 0;
  for (size_t i = 0; i < files.size(); ++i) { size_t m = smp[i].size(), lo = m / 2; double a = 0; for (size_t j = lo; j < m; ++j) a += smp[i][j]; a /= (m - lo); if (a - idle[i] > rise) { rise = a - idle[i]; best = i; pw = a; } }
  double waves = grid * (wg_per_cu_threads / 64.0);
  double mf = (SHAPE == 32 ? 4.0 : 8.0) * iters, fl = mf * (SHAPE == 32 ? 32768.0 : 16384.0) * waves * n;
  printf("%-28s %7.1f TFLOP/s  %6.0f W  %.3f pJ/FLOP  (%.2f cycles/MFMA at 2.4 GHz)\n", name, fl / el * 1e-12, pw, pw * el / fl * 1e12,
         el / n / mf * 2.4e9);
}

int main(int argc, char** argv) {
  double secs = argc > 1 ? atof(argv[1]) : 3.0;
  std::vector<unsigned short> h(4096 * 8); unsigned long long s = 88172645463325252ull;
  for (auto& x : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; float f = ((int)(s & 0xFFFF) - 32768) / 32768.0f; unsigned u; memcpy(&u, &f, 4); x = (unsigned short)(u >> 16); }
  bf16x8* in; float* out; CK(hipMalloc(&in, h.size() * 2)); CK(hipMalloc(&out, 4096)); CK(hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  run<32>("32x32x16, 1 wave/SIMD", 256, secs, in, out);
  run<16>("16x16x32, 1 wave/SIMD", 256, secs, in, out);
  run<32>("32x32x16, 2 waves/SIMD", 512, secs, in, out);
  run<16>("16x16x32, 2 waves/SIMD", 512, secs, in, out);
  return 0;
}
