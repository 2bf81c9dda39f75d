// umT5 text-encoder kernels (wan/modules/t5.py).  The linears are ll_gemm_bf16; here: the
// T5 RMS norm at C = 4096, the bias-added un-scaled attention over the 512 text positions, the python-GELU gate and the
// token-embedding gather.  Runs once per prompt (~5 TFLOP), so these kernels are written for exactness of the
// reference's bf16 rounding points first; the GEMMs carry the time.
#include "gemm_common.h"

// T5LayerNorm.forward (t5.py:57-63): y = bf16(w * bf16(x * rsqrt(mean(x^2) + eps))), statistics in fp32.  One 256-thread workgroup
// per row (512 rows: a wave per row left three quarters of the device without a wave): thread t owns columns 8 t + 2048 i; the sum
// of squares is taken per thread in column order, per wave by shuffles, then ((w0 + w1) + w2) + w3 -- t5_block_sum, shared with
// gemm_ksplit_reduce_norm_kernel (gemm_asm.hip) so that the fused pass and this kernel give the same bits.
__global__ __launch_bounds__(256) void t5_rmsnorm_kernel(const bf16* __restrict__ x, const bf16* __restrict__ w,
                                                         bf16* __restrict__ out, int rows, int C, float eps) {
  __shared__ float sh[4];
  const int row = blockIdx.x;
  const bf16* xr = x + (size_t)row * C;
  float ss = 0.f;
  for (int c = threadIdx.x * 8; c < C; c += 2048) {
    bf16x8 v = *reinterpret_cast<const bf16x8*>(xr + c);
#pragma unroll
    for (int j = 0; j < 8; ++j) ss += (float)v[j] * (float)v[j];
  }
  ss = t5_block_sum(ss, sh);
  const float r = rsqrtf(ss / (float)C + eps);
  bf16* orow = out + (size_t)row * C;
  for (int c = threadIdx.x * 8; c < C; c += 2048) {
    bf16x8 v = *reinterpret_cast<const bf16x8*>(xr + c);
    bf16x8 g = *reinterpret_cast<const bf16x8*>(w + c);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16)((float)g[j] * rbf((float)v[j] * r));
    *reinterpret_cast<bf16x8*>(orow + c) = o;
  }
}

// T5FeedForward's gate (t5.py:46-50,134-139): out = bf16(fc1 * GELU(gate)) with the reference's op-by-op bf16 rounding of
// 0.5 * x * (1 + tanh(sqrt(2/pi) * (x + 0.044715 * x^3))).  h [M, 2F] = [gate | fc1] (one fused GEMM), out [M, F].
__device__ __forceinline__ float gelu_py_bf16(float x) {
  float x3 = rbf(x * x * x);
  float t = rbf(x + rbf(0.044715f * x3));
  t = rbf(tanhf(rbf(0.7978845608028654f * t)));
  return rbf(rbf(0.5f * x) * rbf(1.0f + t));
}

__global__ __launch_bounds__(256) void t5_gated_gelu_kernel(const bf16* __restrict__ h, bf16* __restrict__ out, long long M,
                                                            int F) {
  long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 8;
  if (i >= M * F) return;
  long long m = i / F;
  int f = (int)(i - m * F);
  bf16x8 g = *reinterpret_cast<const bf16x8*>(h + m * 2 * F + f);
  bf16x8 u = *reinterpret_cast<const bf16x8*>(h + m * 2 * F + F + f);
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (bf16)((float)u[j] * gelu_py_bf16((float)g[j]));
  *reinterpret_cast<bf16x8*>(out + i) = o;
}

// token_embedding(ids) (t5.py:297): one wave per token row.
__global__ __launch_bounds__(256) void gather_rows_kernel(const bf16* __restrict__ table, const long long* __restrict__ ids,
                                                          bf16* __restrict__ out, int n, int C, long long vocab) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  long long id = ids[row];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);      // host validates; never read out of bounds
  const bf16* src = table + (size_t)id * C;
  bf16* dst = out + (size_t)row * C;
  for (int c = lane * 8; c < C; c += 512) *reinterpret_cast<bf16x8*>(dst + c) = *reinterpret_cast<const bf16x8*>(src + c);
}

// T5Attention.forward (t5.py:85-117) for one (64-query tile, head): S = bf16(q k^T) (no 1/sqrt(d)), S' = bf16(S + bias),
// masked keys = finfo(bf16).min, P = bf16(softmax_fp32(S')), O = bf16(P v).  head_dim 64, L = 64 * LT keys all resident:
// K [L][64] (XOR-swizzled 128-byte rows) and V^T [64][L + 8] in LDS; the whole score row of a query lives in registers.
// MFMA 16x16x32 with the keys as the A operand, so lane (fr, fg) holds, for query fr, keys 16a + 4fg + {0..3} of tile a;
// the P operand of the second contraction is that accumulator tile pair re-packed in place (k permuted identically on
// the V^T side: two 8-byte LDS reads per operand).
#define T5_BF16_MIN (-3.3895313892515355e38f)

template <int LT>
__global__ __launch_bounds__(256) void t5_attn_kernel(const bf16* __restrict__ q, const bf16* __restrict__ k, int ldqk,
                                                      const bf16* __restrict__ vt, const bf16* __restrict__ bias_tab,
                                                      bf16* __restrict__ out, int ldo, int seq_len) {
  constexpr int L = 64 * LT, NA = L / 16, VROW = L + 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ks = smem;                                        // [L][128 B]
  bf16* vs = reinterpret_cast<bf16*>(smem + L * 128);     // [64][VROW]
  bf16* bs = vs + 64 * VROW;                              // [2L - 1] (+ pad)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int head = blockIdx.y, q0 = blockIdx.x * 64 + wave * 16;
  const int fr = lane & 15, fg = lane >> 4;

  for (int c = tid; c < L * 8; c += 256) {                // K rows of this head
    int key = c >> 3, ch = c & 7;
    i32x4 v = *reinterpret_cast<const i32x4*>(k + (size_t)key * ldqk + head * 64 + ch * 8);
    *reinterpret_cast<i32x4*>(ks + key * 128 + ((ch ^ (key & 7)) << 4)) = v;
  }
  for (int c = tid; c < 64 * (L / 8); c += 256) {         // V^T rows of this head
    int d = c / (L / 8), ch = c - d * (L / 8);
    i32x4 v = *reinterpret_cast<const i32x4*>(vt + (size_t)(head * 64 + d) * L + ch * 8);
    *reinterpret_cast<i32x4*>(vs + d * VROW + ch * 8) = v;
  }
  for (int c = tid; c < 2 * L - 1; c += 256) bs[c] = bias_tab[(size_t)head * (2 * L - 1) + c];
  bf16x8 qf[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(q + (size_t)(q0 + fr) * ldqk + head * 64 + s * 32 + fg * 8);
  __syncthreads();

  f32x4 acc[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a) {
    acc[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      int key = a * 16 + fr;
      bf16x8 kf = *reinterpret_cast<const bf16x8*>(ks + key * 128 + (((s * 4 + fg) ^ (key & 7)) << 4));
      acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[s], acc[a], 0, 0, 0);
    }
  }
  // bias, mask, fp32 softmax over the row (4 * NA values in this lane, the rest in lanes fr + 16 * {0..3})
  const int qi = q0 + fr;
  float mx = -INFINITY;
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int key = a * 16 + fg * 4 + j;
      float s = rbf(rbf(acc[a][j]) + (float)bs[key - qi + L - 1]);
      s = key < seq_len ? s : T5_BF16_MIN;
      acc[a][j] = s;
      mx = fmaxf(mx, s);
    }
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float e = __expf(acc[a][j] - mx);
      acc[a][j] = e;
      sum += e;
    }
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);

  f32x4 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < NA / 2; ++s) {
    bf16x8 pf;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      pf[j] = (bf16)(acc[2 * s][j] / sum);
      pf[4 + j] = (bf16)(acc[2 * s + 1][j] / sum);
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const bf16* vr = vs + (dt * 16 + fr) * VROW + s * 32 + fg * 4;
      bf16x4 lo = *reinterpret_cast<const bf16x4*>(vr), hi = *reinterpret_cast<const bf16x4*>(vr + 16);
      bf16x8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o[dt], 0, 0, 0);
    }
  }
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    bf16x4 ov = {(bf16)o[dt][0], (bf16)o[dt][1], (bf16)o[dt][2], (bf16)o[dt][3]};
    *reinterpret_cast<bf16x4*>(out + (size_t)qi * ldo + head * 64 + dt * 16 + fg * 4) = ov;
  }
}

// ===============================================================================================================
extern "C" int ll_t5_rmsnorm(const ll_bf16* x, const ll_bf16* w, ll_bf16* out, int rows, int C, float eps, ll_stream stream) {
  LL_REQUIRE(C > 0 && C % 8 == 0, "ll_t5_rmsnorm: C=%d must be a multiple of 8", C);
  if (rows == 0) return LL_OK;
  hipLaunchKernelGGL(t5_rmsnorm_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (const bf16*)w,
                     (bf16*)out, rows, C, eps);
  return ll_check_launch("ll_t5_rmsnorm");
}

extern "C" int ll_t5_gated_gelu(const ll_bf16* h, ll_bf16* out, long long M, int F, ll_stream stream) {
  LL_REQUIRE(F > 0 && F % 8 == 0, "ll_t5_gated_gelu: F=%d must be a multiple of 8", F);
  if (M == 0) return LL_OK;
  long long n8 = M * F / 8;
  hipLaunchKernelGGL(t5_gated_gelu_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16*)h,
                     (bf16*)out, M, F);
  return ll_check_launch("ll_t5_gated_gelu");
}

extern "C" int ll_gather_rows(const ll_bf16* table, const long long* ids, ll_bf16* out, int n, int C, long long vocab,
                              ll_stream stream) {
  LL_REQUIRE(C > 0 && C % 8 == 0 && vocab > 0, "ll_gather_rows: C=%d must be a multiple of 8, vocab > 0", C);
  if (n == 0) return LL_OK;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16*)table, ids, (bf16*)out,
                     n, C, vocab);
  return ll_check_launch("ll_gather_rows");
}

extern "C" int ll_t5_attention(const ll_bf16* q, const ll_bf16* k, const ll_bf16* vt, const ll_bf16* bias_tab, ll_bf16* out,
                               int L, int H, int ldqk, int ldo, int seq_len, ll_stream stream) {
  LL_REQUIRE(L == 64 || L == 128 || L == 256 || L == 512, "ll_t5_attention: L=%d must be 64, 128, 256 or 512", L);
  LL_REQUIRE(H > 0 && ldqk >= H * 64 && ldqk % 8 == 0 && ldo >= H * 64 && ldo % 4 == 0, "ll_t5_attention: bad strides");
  LL_REQUIRE(seq_len >= 1 && seq_len <= L, "ll_t5_attention: seq_len=%d outside [1, %d]", seq_len, L);
  dim3 grid(L / 64, H), block(256);
  hipStream_t s = (hipStream_t)stream;
#define T5_LAUNCH(LT)                                                                                                  \
  do {                                                                                                                 \
    size_t lds = (size_t)(64 * LT) * 128 + 64 * (64 * LT + 8) * 2 + (2 * 64 * LT) * 2;                                 \
    {                                                                                          \
      (void)ll_lds_attr((const void*)t5_attn_kernel<LT>, (int)lds); \
    }                                                                                                                  \
    hipLaunchKernelGGL((t5_attn_kernel<LT>), grid, block, lds, s, (const bf16*)q, (const bf16*)k, ldqk, (const bf16*)vt, \
                       (const bf16*)bias_tab, (bf16*)out, ldo, seq_len);                                               \
  } while (0)
  if (L == 64) T5_LAUNCH(1); else if (L == 128) T5_LAUNCH(2); else if (L == 256) T5_LAUNCH(4); else T5_LAUNCH(8);
#undef T5_LAUNCH
  return ll_check_launch("ll_t5_attention");
}
