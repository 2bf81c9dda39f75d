// HBM-bound row kernels: LayerNorm(+modulate), RMSNorm, fused q/k RMSNorm + RoPE + KV-cache insert, KV roll,
// patchify, sinusoid, unpatchify + flow->x0, add_noise.
//
// Layout: one 64-lane wave owns one token row of C (<= 2048, C % 8 == 0) bf16 values: lane l holds the 16-byte
// chunks l, l+64, l+128, ... so every global access is a fully coalesced 1 KiB wave transaction; row statistics are
// reduced with wave shuffles only (no LDS, no barrier).  4 waves (rows) per 256-thread workgroup.
#include "common.h"

#define MAXCH 4  // 4 * 64 lanes * 8 elements = 2048 columns max

// These kernels are VALU-bound as much as HBM-bound (4.6 waves per SIMD, ~600 vector instructions per row before this form), so the
// arithmetic runs on PAIRS of elements: a 32-bit word of the row is two bf16 values, unpack2 makes them an fp32 pair (shift / mask),
// the fp32 steps are v_pk_mul_f32 / v_pk_add_f32 (two elements per instruction), and every bf16 rounding point of the reference is
// one v_cvt_pk_bf16_f32 per pair (+ the shift / mask when the value is used again).  Same operations in the same order per element
// as the scalar form: bit-identical results.
struct RowRegs {
  f32x2 p[MAXCH][4];      // chunk i of the lane: elements 2j, 2j + 1
};
struct RowWords {
  unsigned w[MAXCH][4];   // the same as packed bf16 pairs
};

// FULL: C is a whole number of 512-column chunks (C == 512 NCH; the 1.3B model's 1536): every `chunk inside the row` test is true
// at compile time, so a row's loads are issued together and waited for once -- with the test in place the compiler keeps each
// load inside its own exec-masked branch and waits for it there.
template <bool FULL>
__device__ __forceinline__ bool in_row(int c, int C) {
  return FULL || c < C;
}

template <int NCH, bool FULL>
__device__ __forceinline__ void load_words(const bf16* __restrict__ p, int C, int lane, RowWords& r) {
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    int c = (lane + 64 * i) * 8;
    uint4 t = make_uint4(0u, 0u, 0u, 0u);
    if (in_row<FULL>(c, C)) t = *reinterpret_cast<const uint4*>(p + c);
    r.w[i][0] = t.x, r.w[i][1] = t.y, r.w[i][2] = t.z, r.w[i][3] = t.w;
  }
}

template <int NCH, bool FULL>
__device__ __forceinline__ void load_row(const bf16* __restrict__ p, int C, int lane, RowRegs& r) {
  RowWords w;
  load_words<NCH, FULL>(p, C, lane, w);
#pragma unroll
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) r.p[i][j] = unpack2(w.w[i][j]);
}

// mean and 1/sqrt(var + eps); on return r holds x - mean (what every consumer needs next).  Sums in element order, as before.
template <int NCH, bool FULL>
__device__ __forceinline__ void layernorm_center(RowRegs& r, int C, int lane, float eps, float& rstd) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s += r.p[i][j].x;
      s += r.p[i][j].y;
    }
  float mean = wave_sum(s) / (float)C;
  f32x2 m2 = splat2(mean);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    if (in_row<FULL>((lane + 64 * i) * 8, C)) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x2 d = r.p[i][j] - m2;
        r.p[i][j] = d;
        f32x2 dd = d * d;
        q += dd.x;
        q += dd.y;
      }
    }
  }
  float var = wave_sum(q) / (float)C;
  rstd = 1.0f / sqrtf(var + eps);
}

// Row output from packed bf16 pairs: stored as they are, or (int8 mode) symmetric per-row int8 + scale computed from the SAME
// bf16-rounded values the bf16 path stores, so fused and unfused quantisation are bit-identical (scale = max|y| / 127,
// q = rint(y / scale)).
template <int NCH, bool FULL>
__device__ __forceinline__ void emit_row(const RowWords& y, int C, int lane, int row, bf16* __restrict__ out,
                                         int8_t* __restrict__ q, float* __restrict__ qscale) {
  if (q == nullptr) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      int c = (lane + 64 * i) * 8;
      if (in_row<FULL>(c, C))
        *reinterpret_cast<uint4*>(out + (size_t)row * C + c) = make_uint4(y.w[i][0], y.w[i][1], y.w[i][2], y.w[i][3]);
    }
    return;
  }
  float mx = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i)
    if (in_row<FULL>((lane + 64 * i) * 8, C)) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x2 v = unpack2(y.w[i][j]);
        mx = fmaxf(mx, fabsf(v.x));
        mx = fmaxf(mx, fabsf(v.y));
      }
    }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  float sc = mx > 0.f ? mx / 127.0f : 1.0f;
  float inv = 1.0f / sc;
  if (lane == 0) qscale[row] = sc;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    int c = (lane + 64 * i) * 8;
    if (in_row<FULL>(c, C)) {
      unsigned lo = 0, hi = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {      // element j of the chunk lives in pair j >> 1; elements 0..3 -> lo, 4..7 -> hi
        f32x2 va = unpack2(y.w[i][j >> 1]), vb = unpack2(y.w[i][2 + (j >> 1)]);
        int a = __float2int_rn(((j & 1) ? va.y : va.x) * inv), b = __float2int_rn(((j & 1) ? vb.y : vb.x) * inv);
        a = a < -127 ? -127 : (a > 127 ? 127 : a);
        b = b < -127 ? -127 : (b > 127 ? 127 : b);
        lo |= (unsigned)(a & 0xFF) << (8 * j);
        hi |= (unsigned)(b & 0xFF) << (8 * j);
      }
      *reinterpret_cast<uint2*>(q + (size_t)row * C + c) = make_uint2(lo, hi);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// LN (no affine) + per-frame modulation.  Rounding points of the reference (bf16 tensors, causal_model.py:445):
//   y = bf16(LN(x)); s1 = bf16(1 + bf16(mod_s + e_s)); out = bf16(bf16(y * s1) + bf16(mod_t + e_t))
// PRE: `e` already holds bf16(mod + e) for every chunk (ll_modulation_table, once per forward for all layers): two vector
// loads and three operations per element less; the values are the ones the unfused form computes, bit for bit.
template <int NCH, bool FULL, bool PRE>
__global__ __launch_bounds__(256) void ln_modulate_kernel(const bf16* __restrict__ x, bf16* __restrict__ out,
                                                          const bf16* __restrict__ e, const bf16* __restrict__ mod,
                                                          int nmod, int shift_idx, int scale_idx, int rows, int L,
                                                          int C, int frame_len, int F, float eps,
                                                          int8_t* __restrict__ q, float* __restrict__ qscale) {
  int lane = threadIdx.x & 63;
  int row = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));      // one row per wave: the row decode runs on the scalar unit
  if (row >= rows) return;
  RowRegs r;
  load_row<NCH, FULL>(x + (size_t)row * C, C, lane, r);
  int b = row / L, f = (row % L) / frame_len;
  const bf16* eb = e + ((size_t)(b * F + f) * nmod) * C;
  RowWords es, et, ms, mt;                               // the modulation chunks travel while the row statistics are reduced
  load_words<NCH, FULL>(eb + (size_t)scale_idx * C, C, lane, es);
  load_words<NCH, FULL>(eb + (size_t)shift_idx * C, C, lane, et);
  if (!PRE) {
    load_words<NCH, FULL>(mod + (size_t)scale_idx * C, C, lane, ms);
    load_words<NCH, FULL>(mod + (size_t)shift_idx * C, C, lane, mt);
  }
  asm volatile("" ::: "memory");                          // (the compiler would sink these loads below the reductions again)
  __builtin_amdgcn_sched_barrier(0);
  float rstd;
  layernorm_center<NCH, FULL>(r, C, lane, eps, rstd);
  const f32x2 r2 = splat2(rstd), one = splat2(1.0f);
  RowWords o;
#pragma unroll
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x2 y = rbf2(r.p[i][j] * r2);
      f32x2 sc = PRE ? unpack2(es.w[i][j]) : rbf2(unpack2(ms.w[i][j]) + unpack2(es.w[i][j]));
      f32x2 t = PRE ? unpack2(et.w[i][j]) : rbf2(unpack2(mt.w[i][j]) + unpack2(et.w[i][j]));
      f32x2 s1 = rbf2(one + sc);
      o.w[i][j] = pack2(rbf2(y * s1) + t);
    }
  emit_row<NCH, FULL>(o, C, lane, row, out, q, qscale);
}

// The same from an fp32 table (ll_modulation_table_f32): tab[b, f, scale_idx] = 1 + scale and tab[b, f, shift_idx] = shift, each already
// rounded to bf16 where the reference rounds (s1 and t above, bit for bit) and widened to fp32 -- per pair of elements no unpacking
// and no `1 +` / rounding of the scale: 10 vector instructions instead of 18 in a kernel bound by their count.
template <int NCH, bool FULL>
__global__ __launch_bounds__(256) void ln_modulate_tab_kernel(const bf16* __restrict__ x, bf16* __restrict__ out,
                                                              const float* __restrict__ tab, int nmod, int shift_idx, int scale_idx,
                                                              int rows, int L, int C, int frame_len, int F, float eps,
                                                              int8_t* __restrict__ q, float* __restrict__ qscale) {
  int lane = threadIdx.x & 63;
  int row = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (row >= rows) return;
  RowRegs r;
  load_row<NCH, FULL>(x + (size_t)row * C, C, lane, r);
  int b = row / L, f = (row % L) / frame_len;
  const float* tb = tab + ((size_t)(b * F + f) * nmod) * C;
  f32x2 s1[NCH][4], t[NCH][4];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    int c = (lane + 64 * i) * 8;
    if (in_row<FULL>(c, C)) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        f32x4 a = *reinterpret_cast<const f32x4*>(tb + (size_t)scale_idx * C + c + 4 * h);
        f32x4 d = *reinterpret_cast<const f32x4*>(tb + (size_t)shift_idx * C + c + 4 * h);
        s1[i][2 * h].x = a[0], s1[i][2 * h].y = a[1], s1[i][2 * h + 1].x = a[2], s1[i][2 * h + 1].y = a[3];
        t[i][2 * h].x = d[0], t[i][2 * h].y = d[1], t[i][2 * h + 1].x = d[2], t[i][2 * h + 1].y = d[3];
      }
    }
  }
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  float rstd;
  layernorm_center<NCH, FULL>(r, C, lane, eps, rstd);
  const f32x2 r2 = splat2(rstd);
  RowWords o;
#pragma unroll
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) o.w[i][j] = pack2(rbf2(rbf2(r.p[i][j] * r2) * s1[i][j]) + t[i][j]);
  emit_row<NCH, FULL>(o, C, lane, row, out, q, qscale);
}

// LN with affine (norm3): F.layer_norm computes (x-mean)*rstd*w + b in fp32 and rounds once.
template <int NCH, bool FULL>
__global__ __launch_bounds__(256) void layernorm_affine_kernel(const bf16* __restrict__ x, const bf16* __restrict__ w,
                                                               const bf16* __restrict__ bb, bf16* __restrict__ out,
                                                               int rows, int C, float eps, int8_t* __restrict__ q,
                                                               float* __restrict__ qscale) {
  int lane = threadIdx.x & 63;
  int row = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));      // one row per wave: row arithmetic on the scalar unit
  if (row >= rows) return;
  RowRegs r;
  load_row<NCH, FULL>(x + (size_t)row * C, C, lane, r);
  RowWords wv, bv;
  load_words<NCH, FULL>(w, C, lane, wv);
  load_words<NCH, FULL>(bb, C, lane, bv);
  asm volatile("" ::: "memory");           // every load of the row is in flight before the reductions (see ln_modulate_kernel)
  __builtin_amdgcn_sched_barrier(0);
  float rstd;
  layernorm_center<NCH, FULL>(r, C, lane, eps, rstd);
  const f32x2 r2 = splat2(rstd);
  RowWords o;
#pragma unroll
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) o.w[i][j] = pack2(r.p[i][j] * r2 * unpack2(wv.w[i][j]) + unpack2(bv.w[i][j]));
  emit_row<NCH, FULL>(o, C, lane, row, out, q, qscale);
}

// ---------------------------------------------------------------------------------------------------------------
// WanRMSNorm (model.py:78-86): bf16(x * rsqrt(mean(x^2)+eps)) * w, second product rounded to bf16 again.
template <int NCH>
__device__ __forceinline__ float rms_rinv(const RowRegs& r, int C, float eps) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x2 xx = r.p[i][j] * r.p[i][j];
      s += xx.x;
      s += xx.y;
    }
  return 1.0f / sqrtf(wave_sum(s) / (float)C + eps);
}

template <int NCH, bool FULL>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const bf16* __restrict__ x, const bf16* __restrict__ w,
                                                      bf16* __restrict__ out, int rows, int C, int ldx, int ldo,
                                                      float eps) {
  int lane = threadIdx.x & 63;
  int row = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));      // one row per wave: row arithmetic on the scalar unit
  if (row >= rows) return;
  RowRegs r;
  load_row<NCH, FULL>(x + (size_t)row * ldx, C, lane, r);
  RowWords wv;
  load_words<NCH, FULL>(w, C, lane, wv);
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  const f32x2 ri = splat2(rms_rinv<NCH>(r, C, eps));
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    int c = (lane + 64 * i) * 8;
    if (in_row<FULL>(c, C)) {
      unsigned o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = pack2(rbf2(r.p[i][j] * ri) * unpack2(wv.w[i][j]));
      *reinterpret_cast<uint4*>(out + (size_t)row * ldo + c) = make_uint4(o[0], o[1], o[2], o[3]);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Fused q/k RMSNorm + RoPE + KV insert.  One wave per token: reads the q, k and v thirds of the fused QKV row once,
// writes roped q to q_out and roped k / raw v straight into their KV-cache slots (no clone / cat / second pass).
// RoPE pairs are (2i, 2i+1) inside each 128-wide head (causal_model.py:45-55); pair p < nf rotates by the frame
// angle, the rest by the (h, w) angles; angles come from fp32 (cos, sin) tables made from the fp64 table on the host.
// (This kernel keeps the scalar one-element form: the pair form of the kernels above -- with the row's loads issued together -- was
// measured at 22.5 - 24 us against 17.5 us for this one in the model-order replay although it issues a third fewer vector
// instructions; profiles/r04_row_kernels.md.)
struct RowF {
  float v[MAXCH][8];
};

template <int NCH>
__device__ __forceinline__ void load_row_f(const bf16* __restrict__ p, int C, int lane, RowF& r) {
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    int c = (lane + 64 * i) * 8;
    if (c < C) {
      bf16x8 t = *reinterpret_cast<const bf16x8*>(p + c);
#pragma unroll
      for (int j = 0; j < 8; ++j) r.v[i][j] = (float)t[j];
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) r.v[i][j] = 0.f;
    }
  }
}


template <int NCH>
__device__ __forceinline__ float rms_rinv_f(const RowF& r, int C, float eps) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) s += r.v[i][j] * r.v[i][j];
  return 1.0f / sqrtf(wave_sum(s) / (float)C + eps);
}

template <int NCH>
__device__ __forceinline__ void norm_rope_store(RowF& r, const bf16* __restrict__ w, int C, int lane, float eps,
                                                const float2* __restrict__ rf, const float2* __restrict__ rhw, int nf,
                                                int half_hd, bf16* __restrict__ dst) {
  float rinv = rms_rinv_f<NCH>(r, C, eps);
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    int c = (lane + 64 * i) * 8;
    if (c < C) {
      bf16x8 wv = *reinterpret_cast<const bf16x8*>(w + c);
      int p0 = (c >> 1) % half_hd;  // first pair index inside the head
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float a = rbf(rbf(r.v[i][2 * j] * rinv) * (float)wv[2 * j]);
        float bq = rbf(rbf(r.v[i][2 * j + 1] * rinv) * (float)wv[2 * j + 1]);
        int p = p0 + j;
        float2 cs = (p < nf) ? rf[p] : rhw[p - nf];
        o[2 * j] = (bf16)(a * cs.x - bq * cs.y);
        o[2 * j + 1] = (bf16)(a * cs.y + bq * cs.x);
      }
      if (dst) *reinterpret_cast<bf16x8*>(dst + c) = o;
    }
  }
}

template <int NCH>
__global__ __launch_bounds__(256) void qk_norm_rope_kv_kernel(
    const bf16* __restrict__ qkv, const bf16* __restrict__ wq, const bf16* __restrict__ wk,
    const float2* __restrict__ rope_f, const float2* __restrict__ rope_hw, bf16* __restrict__ q_out,
    bf16* __restrict__ cache_k, bf16* __restrict__ cache_v, int rows, int L, int C, int half_hd, int nf,
    int frame_len, int start_frame, int S, int write_start, int roped_offset, int write_len, float eps) {
  int lane = threadIdx.x & 63;
  int row = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));      // one row per wave: the row decode runs on the scalar unit
  if (row >= rows) return;
  int b = row / L, t = row % L;
  int f = t / frame_len + start_frame, sp = t % frame_len;
  const float2* rf = rope_f + (size_t)f * nf;
  const float2* rhw = rope_hw + (size_t)sp * (half_hd - nf);
  const bf16* src = qkv + (size_t)row * 3 * C;
  RowF r;
  load_row_f<NCH>(src, C, lane, r);
  norm_rope_store<NCH>(r, wq, C, lane, eps, rf, rhw, nf, half_hd, q_out + (size_t)row * C);
  int wi = t - roped_offset;
  bool wr = (wi >= 0) && (wi < write_len);
  size_t slot = ((size_t)b * S + write_start + wi) * C;
  load_row_f<NCH>(src + C, C, lane, r);
  norm_rope_store<NCH>(r, wk, C, lane, eps, rf, rhw, nf, half_hd, wr ? cache_k + slot : nullptr);
  if (wr && cache_v != nullptr) {        // cache_v == NULL: the QKV projection's epilogue has already inserted V (ll_gemm_*_qkv)
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      int c = (lane + 64 * i) * 8;
      if (c < C) *reinterpret_cast<uint4*>(cache_v + slot + c) = *reinterpret_cast<const uint4*>(src + 2 * C + c);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// KV roll: plain 16-byte copy of n rows; the host issues it in chunks of (src - dst) rows so that source and
// destination of one launch never overlap (stream order serialises the chunks).
__global__ __launch_bounds__(256) void copy_rows_kernel(uint4* __restrict__ kbase, uint4* __restrict__ vbase,
                                                        size_t batch_stride16, size_t dst16, size_t src16,
                                                        size_t n16) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  size_t stride = (size_t)gridDim.x * 256;
  uint4* base = (blockIdx.z ? vbase : kbase) + (size_t)blockIdx.y * batch_stride16;
  for (; i < n16; i += stride) base[dst16 + i] = base[src16 + i];
}

// ---------------------------------------------------------------------------------------------------------------
// patchify: patches[b, (f,h,w), c*4 + p*2 + q] = x[b, f, c, 2h+p, 2w+q]   (Conv3d weight.flatten(1) column order)
__global__ __launch_bounds__(256) void patchify_kernel(const bf16* __restrict__ x, bf16* __restrict__ out, int B, int F,
                                                       int Cin, int H, int W) {
  int hp = H / 2, wp = W / 2;
  size_t total = (size_t)B * F * hp * wp * Cin * 2;  // one thread per (token, c, p): writes the q = 0,1 pair
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  int p = i & 1;
  size_t r = i >> 1;
  int c = r % Cin;
  r /= Cin;
  int w = r % wp;
  r /= wp;
  int h = r % hp;
  r /= hp;  // r = b*F + f
  const bf16* s = x + ((r * Cin + c) * H + (2 * h + p)) * (size_t)W + 2 * w;
  bf16x2 v = *reinterpret_cast<const bf16x2*>(s);
  size_t tok = (r * hp + h) * wp + w;
  *reinterpret_cast<bf16x2*>(out + tok * (Cin * 4) + c * 4 + p * 2) = v;
}

// sinusoidal_embedding_1d in fp64 (model.py:15-25): [cos(t w_j), sin(t w_j)], w_j = 10000^(-j/half)
__global__ void sinusoid_kernel(const float* __restrict__ t, bf16* __restrict__ out, int n, int dim) {
  int half = dim / 2;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * half) return;
  int r = i / half, j = i % half;
  double w = pow(10000.0, -((double)j / (double)half));
  double a = (double)t[r] * w;
  out[(size_t)r * dim + j] = (bf16)(float)cos(a);
  out[(size_t)r * dim + half + j] = (bf16)(float)sin(a);
}

// unpatchify + flow->x0.  head[b, (f,h,w), (q*2+r)*Cout + c] -> flow[b,f,c,2h+q,2w+r];
// x0 = bf16(float(double(xt) - double(sigma) * double(flow)))   (wan_wrapper.py:189-199: fp64, cast back via float)
__global__ __launch_bounds__(256) void unpatchify_x0_kernel(const bf16* __restrict__ head, const bf16* __restrict__ xt,
                                                            const float* __restrict__ sigma, bf16* __restrict__ flow,
                                                            bf16* __restrict__ x0, int B, int F, int Cout, int H,
                                                            int W) {
  size_t total = (size_t)B * F * Cout * H * (W / 2);
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  int wp = W / 2, hp = H / 2;
  size_t r = i;
  int w2 = r % wp;
  r /= wp;
  int y = r % H;
  r /= H;
  int c = r % Cout;
  r /= Cout;  // r = b*F + f
  int h = y >> 1, q = y & 1;
  size_t tok = (r * hp + h) * wp + w2;
  const bf16* hs = head + tok * (4 * Cout) + (q * 2) * Cout + c;
  float f0 = (float)hs[0], f1 = (float)hs[Cout];
  size_t o = ((r * Cout + c) * H + y) * (size_t)W + 2 * w2;
  bf16x2 xv = *reinterpret_cast<const bf16x2*>(xt + o);
  double sg = (double)sigma[r];
  bf16x2 fo, xo;
  fo[0] = (bf16)f0;
  fo[1] = (bf16)f1;
  xo[0] = (bf16)(float)__dsub_rn((double)(float)xv[0], __dmul_rn(sg, (double)f0));
  xo[1] = (bf16)(float)__dsub_rn((double)(float)xv[1], __dmul_rn(sg, (double)f1));
  *reinterpret_cast<bf16x2*>(flow + o) = fo;
  *reinterpret_cast<bf16x2*>(x0 + o) = xo;
}

// add_noise: fp32, no contraction: bf16( (1-s)*x0 + s*n )
__global__ __launch_bounds__(256) void add_noise_kernel(const bf16* __restrict__ x0, const bf16* __restrict__ nz,
                                                        const float* __restrict__ sigma, bf16* __restrict__ out,
                                                        long long inner8) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= inner8) return;
  int n = blockIdx.y;
  float s = sigma[n];
  float oms = __fsub_rn(1.0f, s);
  size_t o = ((size_t)n * inner8 + i) * 8;
  bf16x8 a = *reinterpret_cast<const bf16x8*>(x0 + o);
  bf16x8 b = *reinterpret_cast<const bf16x8*>(nz + o);
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (bf16)__fadd_rn(__fmul_rn(oms, (float)a[j]), __fmul_rn(s, (float)b[j]));
  *reinterpret_cast<bf16x8*>(out + o) = r;
}


// sigma_t = sigmas[argmin_i |timesteps[i] - t|]  (utils/wan_wrapper.py:195-197, utils/scheduler.py:172-174).
// One wave per query; fp64 differences; ties resolve to the lowest index like torch.argmin.
__global__ __launch_bounds__(64) void sigma_lookup_kernel(const float* __restrict__ t, const float* __restrict__ timesteps,
                                                          const float* __restrict__ sigmas, float* __restrict__ out,
                                                          int n_table) {
  int lane = threadIdx.x;
  double tv = (double)t[blockIdx.x];
  double best = INFINITY;
  int bi = 0x7fffffff;
  for (int i = lane; i < n_table; i += 64) {
    double d = fabs((double)timesteps[i] - tv);
    if (d < best) { best = d; bi = i; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    double ob = __shfl_xor(best, o, 64);
    int oi = __shfl_xor(bi, o, 64);
    if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  if (lane == 0) out[blockIdx.x] = sigmas[bi];
}

// out[l][bf][i][:] = bf16(mods[l][i][:] + e[bf][i][:]): the per-frame modulation vectors of ALL layers of one forward
// (causal_model.py:440: `e = (self.modulation.unsqueeze(1) + e).chunk(6, dim=2)`), computed once instead of once per token row.
__global__ __launch_bounds__(256) void modulation_table_kernel(const bf16* __restrict__ e, const bf16* __restrict__ mods,
                                                               bf16* __restrict__ out, int NL, int BF, int nmodC8) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;           // one 16-byte chunk of the output
  long long total = (long long)NL * BF * nmodC8;
  if (i >= total) return;
  int c = (int)(i % nmodC8);
  int bf = (int)((i / nmodC8) % BF);
  int l = (int)(i / ((long long)nmodC8 * BF));
  bf16x8 a = *reinterpret_cast<const bf16x8*>(mods + ((size_t)l * nmodC8 + c) * 8);
  bf16x8 b = *reinterpret_cast<const bf16x8*>(e + ((size_t)bf * nmodC8 + c) * 8);
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (bf16)((float)a[j] + (float)b[j]);
  *reinterpret_cast<bf16x8*>(out + (size_t)i * 8) = o;
}

// The fp32 form for ll_ln_modulate_tab: out[l][bf][i][:] = float(bf16(mods + e)) -- and float(bf16(1 + bf16(mods + e))) for the
// chunks whose bit is set in one_plus (the scale chunks: what ln_modulate forms per element as s1).
__global__ __launch_bounds__(256) void modulation_table_f32_kernel(const bf16* __restrict__ e, const bf16* __restrict__ mods,
                                                                   float* __restrict__ out, int NL, int BF, int nmod, int C8,
                                                                   unsigned one_plus) {
  const int nmodC8 = nmod * C8;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;           // one 8-element chunk of the output
  long long total = (long long)NL * BF * nmodC8;
  if (i >= total) return;
  int c = (int)(i % nmodC8);
  int bf = (int)((i / nmodC8) % BF);
  int l = (int)(i / ((long long)nmodC8 * BF));
  bool plus = (one_plus >> (c / C8)) & 1u;
  bf16x8 a = *reinterpret_cast<const bf16x8*>(mods + ((size_t)l * nmodC8 + c) * 8);
  bf16x8 b = *reinterpret_cast<const bf16x8*>(e + ((size_t)bf * nmodC8 + c) * 8);
  f32x4 o0, o1;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float v = rbf((float)a[j] + (float)b[j]);
    if (plus) v = rbf(1.0f + v);
    if (j < 4) o0[j] = v; else o1[j - 4] = v;
  }
  *reinterpret_cast<f32x4*>(out + (size_t)i * 8) = o0;
  *reinterpret_cast<f32x4*>(out + (size_t)i * 8 + 4) = o1;
}

// ===============================================================================================================
// host launchers
#define DISPATCH_NCH(C, CALL)                                              \
  do {                                                                     \
    int nch_ = ((C) + 511) / 512;                                          \
    bool full_ = ((C) % 512) == 0;                                         \
    if (nch_ == 1) { if (full_) CALL(1, true); else CALL(1, false); }      \
    else if (nch_ == 2) { if (full_) CALL(2, true); else CALL(2, false); } \
    else if (nch_ == 3) { if (full_) CALL(3, true); else CALL(3, false); } \
    else { if (full_) CALL(4, true); else CALL(4, false); }                \
  } while (0)

static inline bool row_ok(int C) { return C > 0 && C <= 2048 && (C % 8) == 0; }

static int ln_modulate_launch(const ll_bf16* x, ll_bf16* out, int8_t* q, float* qscale, const ll_bf16* e,
                              const ll_bf16* mod, int nmod, int shift_idx, int scale_idx, int B, int L, int C, int F,
                              float eps, ll_stream stream) {
  LL_REQUIRE(row_ok(C), "ll_ln_modulate: C=%d must be a multiple of 8 and <= 2048", C);
  LL_REQUIRE(F > 0 && L % F == 0, "ll_ln_modulate: L=%d not divisible by F=%d", L, F);
  LL_REQUIRE(shift_idx >= 0 && shift_idx < nmod && scale_idx >= 0 && scale_idx < nmod, "ll_ln_modulate: bad mod index");
  int rows = B * L;
  if (rows == 0) return LL_OK;
  dim3 grid((rows + 3) / 4);
#define CALL(N, FL)                                                                                                          \
  do {                                                                                                                   \
    if (mod)                                                                                                             \
      hipLaunchKernelGGL((ln_modulate_kernel<N, FL, false>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (bf16*)out, \
                         (const bf16*)e, (const bf16*)mod, nmod, shift_idx, scale_idx, rows, L, C, L / F, F, eps, q, qscale); \
    else                                                                                                                 \
      hipLaunchKernelGGL((ln_modulate_kernel<N, FL, true>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (bf16*)out,  \
                         (const bf16*)e, (const bf16*)mod, nmod, shift_idx, scale_idx, rows, L, C, L / F, F, eps, q, qscale); \
  } while (0)
  DISPATCH_NCH(C, CALL);
#undef CALL
  return ll_check_launch("ll_ln_modulate");
}

extern "C" int ll_modulation_table(const ll_bf16* e, const ll_bf16* mods, ll_bf16* out, int num_layers, int BF, int nmod, int C,
                                   ll_stream stream) {
  LL_REQUIRE(C > 0 && C % 8 == 0 && nmod > 0 && num_layers >= 0 && BF >= 0, "ll_modulation_table: bad shape");
  long long chunks = (long long)num_layers * BF * nmod * (C / 8);
  if (chunks == 0) return LL_OK;
  LL_REQUIRE(chunks < (1LL << 31) * 256, "ll_modulation_table: too large");
  hipLaunchKernelGGL(modulation_table_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)e, (const bf16*)mods, (bf16*)out, num_layers, BF, nmod * (C / 8));
  return ll_check_launch("ll_modulation_table");
}

extern "C" int ll_ln_modulate(const ll_bf16* x, ll_bf16* out, const ll_bf16* e, const ll_bf16* mod, int nmod,
                              int shift_idx, int scale_idx, int B, int L, int C, int F, float eps, ll_stream stream) {
  return ln_modulate_launch(x, out, nullptr, nullptr, e, mod, nmod, shift_idx, scale_idx, B, L, C, F, eps, stream);
}

extern "C" int ll_ln_modulate_q8(const ll_bf16* x, int8_t* q, float* qscale, const ll_bf16* e, const ll_bf16* mod,
                                 int nmod, int shift_idx, int scale_idx, int B, int L, int C, int F, float eps,
                                 ll_stream stream) {
  LL_REQUIRE(q && qscale, "ll_ln_modulate_q8: q and qscale are required");
  return ln_modulate_launch(x, nullptr, q, qscale, e, mod, nmod, shift_idx, scale_idx, B, L, C, F, eps, stream);
}

extern "C" int ll_modulation_table_f32(const ll_bf16* e, const ll_bf16* mods, float* out, int num_layers, int BF, int nmod, int C,
                                       unsigned one_plus_mask, ll_stream stream) {
  LL_REQUIRE(C > 0 && C % 8 == 0 && nmod > 0 && nmod <= 32 && num_layers >= 0 && BF >= 0, "ll_modulation_table_f32: bad shape");
  LL_REQUIRE(nmod == 32 || (one_plus_mask >> nmod) == 0, "ll_modulation_table_f32: one_plus_mask names a chunk >= nmod=%d", nmod);
  long long chunks = (long long)num_layers * BF * nmod * (C / 8);
  if (chunks == 0) return LL_OK;
  LL_REQUIRE(chunks < (1LL << 31) * 256, "ll_modulation_table_f32: too large");
  hipLaunchKernelGGL(modulation_table_f32_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)e, (const bf16*)mods, out, num_layers, BF, nmod, C / 8, one_plus_mask);
  return ll_check_launch("ll_modulation_table_f32");
}

extern "C" int ll_ln_modulate_tab(const ll_bf16* x, ll_bf16* out, int8_t* q, float* qscale, const float* tab, int nmod,
                                  int shift_idx, int scale_idx, int B, int L, int C, int F, float eps, ll_stream stream) {
  LL_REQUIRE(row_ok(C), "ll_ln_modulate_tab: C=%d must be a multiple of 8 and <= 2048", C);
  LL_REQUIRE(F > 0 && L % F == 0, "ll_ln_modulate_tab: L=%d not divisible by F=%d", L, F);
  LL_REQUIRE(shift_idx >= 0 && shift_idx < nmod && scale_idx >= 0 && scale_idx < nmod, "ll_ln_modulate_tab: bad mod index");
  LL_REQUIRE((out != nullptr) != (q != nullptr), "ll_ln_modulate_tab: exactly one of out (bf16) and q (int8, with qscale) is required");
  LL_REQUIRE(q == nullptr || qscale != nullptr, "ll_ln_modulate_tab: q needs qscale");
  int rows = B * L;
  if (rows == 0) return LL_OK;
  dim3 grid((rows + 3) / 4);
#define CALL(N, FL)                                                                                                      \
  hipLaunchKernelGGL((ln_modulate_tab_kernel<N, FL>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (bf16*)out, tab, \
                     nmod, shift_idx, scale_idx, rows, L, C, L / F, F, eps, q, qscale)
  DISPATCH_NCH(C, CALL);
#undef CALL
  return ll_check_launch("ll_ln_modulate_tab");
}

static int layernorm_affine_launch(const ll_bf16* x, const ll_bf16* w, const ll_bf16* b, ll_bf16* out, int8_t* q,
                                   float* qscale, int rows, int C, float eps, ll_stream stream) {
  LL_REQUIRE(row_ok(C), "ll_layernorm_affine: C=%d must be a multiple of 8 and <= 2048", C);
  if (rows == 0) return LL_OK;
  dim3 grid((rows + 3) / 4);
#define CALL(N, FL)                                                                                              \
  hipLaunchKernelGGL((layernorm_affine_kernel<N, FL>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)x,   \
                     (const bf16*)w, (const bf16*)b, (bf16*)out, rows, C, eps, q, qscale)
  DISPATCH_NCH(C, CALL);
#undef CALL
  return ll_check_launch("ll_layernorm_affine");
}

extern "C" int ll_layernorm_affine(const ll_bf16* x, const ll_bf16* w, const ll_bf16* b, ll_bf16* out, int rows, int C,
                                   float eps, ll_stream stream) {
  return layernorm_affine_launch(x, w, b, out, nullptr, nullptr, rows, C, eps, stream);
}

extern "C" int ll_layernorm_affine_q8(const ll_bf16* x, const ll_bf16* w, const ll_bf16* b, int8_t* q, float* qscale,
                                      int rows, int C, float eps, ll_stream stream) {
  LL_REQUIRE(q && qscale, "ll_layernorm_affine_q8: q and qscale are required");
  return layernorm_affine_launch(x, w, b, nullptr, q, qscale, rows, C, eps, stream);
}

extern "C" int ll_rmsnorm(const ll_bf16* x, const ll_bf16* w, ll_bf16* out, int rows, int C, int ldx, int ldo,
                          float eps, ll_stream stream) {
  LL_REQUIRE(row_ok(C), "ll_rmsnorm: C=%d must be a multiple of 8 and <= 2048", C);
  LL_REQUIRE(ldx % 8 == 0 && ldo % 8 == 0 && ldx >= C && ldo >= C, "ll_rmsnorm: bad row strides %d %d", ldx, ldo);
  if (rows == 0) return LL_OK;
  dim3 grid((rows + 3) / 4);
#define CALL(N, FL)                                                                                                  \
  hipLaunchKernelGGL((rmsnorm_kernel<N, FL>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (const bf16*)w, \
                     (bf16*)out, rows, C, ldx, ldo, eps)
  DISPATCH_NCH(C, CALL);
#undef CALL
  return ll_check_launch("ll_rmsnorm");
}

extern "C" int ll_qk_norm_rope_kv_store(const ll_bf16* qkv, const ll_bf16* wq, const ll_bf16* wk, const float* rope_f,
                                        const float* rope_hw, ll_bf16* q_out, ll_bf16* cache_k, ll_bf16* cache_v,
                                        int B, int L, int C, int head_dim, int frame_len, int start_frame, int S,
                                        int write_start, int roped_offset, int write_len, float eps,
                                        ll_stream stream) {
  LL_REQUIRE(row_ok(C), "ll_qk_norm_rope_kv_store: C=%d must be a multiple of 8 and <= 2048", C);
  LL_REQUIRE(head_dim > 0 && head_dim % 8 == 0 && C % head_dim == 0, "ll_qk_norm_rope_kv_store: bad head_dim %d", head_dim);
  LL_REQUIRE(frame_len > 0 && L % frame_len == 0, "ll_qk_norm_rope_kv_store: L=%d not a whole number of frames (%d)", L, frame_len);
  LL_REQUIRE(start_frame >= 0 && start_frame + L / frame_len <= 1024, "ll_qk_norm_rope_kv_store: frame index beyond the 1024-entry RoPE table");
  // write_len == 0 with roped_offset > L is legal: a recompute pass whose tokens all lie inside the protected sink writes nothing
  // (wan/modules/causal_model.py:302-311 with write_start = sink_tokens > local_end: roped_offset = sink - local_start, write_len = 0)
  LL_REQUIRE(write_len >= 0 && roped_offset >= 0 && (write_len == 0 || roped_offset + write_len <= L), "ll_qk_norm_rope_kv_store: write window outside the new tokens");
  LL_REQUIRE(write_len == 0 || (write_start >= 0 && write_start + write_len <= S), "ll_qk_norm_rope_kv_store: write [%d,+%d) outside cache of %d slots", write_start, write_len, S);
  int rows = B * L;
  if (rows == 0) return LL_OK;
  int half = head_dim / 2;
  int c3 = half / 3;
  int nf = half - 2 * c3;
  dim3 grid((rows + 3) / 4);
#define CALL(N, FL)                                                                                                        \
  hipLaunchKernelGGL((qk_norm_rope_kv_kernel<N>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)qkv,            \
                     (const bf16*)wq, (const bf16*)wk, (const float2*)rope_f, (const float2*)rope_hw, (bf16*)q_out,    \
                     (bf16*)cache_k, (bf16*)cache_v, rows, L, C, half, nf, frame_len, start_frame, S, write_start,     \
                     roped_offset, write_len, eps)
  DISPATCH_NCH(C, CALL);
#undef CALL
  return ll_check_launch("ll_qk_norm_rope_kv_store");
}

extern "C" int ll_kv_roll(ll_bf16* cache_k, ll_bf16* cache_v, int B, int S, int C, int dst, int src, int n,
                          ll_stream stream) {
  LL_REQUIRE(C % 8 == 0, "ll_kv_roll: C=%d must be a multiple of 8", C);
  LL_REQUIRE(src > dst && dst >= 0, "ll_kv_roll: needs src > dst >= 0 (got dst=%d src=%d)", dst, src);
  LL_REQUIRE(n >= 0 && src + n <= S, "ll_kv_roll: source rows [%d,+%d) outside cache of %d slots", src, n, S);
  if (n == 0 || B == 0) return LL_OK;
  size_t row16 = (size_t)C / 8;
  int step = src - dst;
  for (int done = 0; done < n; done += step) {
    int m = (n - done < step) ? (n - done) : step;
    size_t n16 = (size_t)m * row16;
    int blocks = (int)((n16 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(copy_rows_kernel, dim3(blocks, B, 2), dim3(256), 0, (hipStream_t)stream, (uint4*)cache_k,
                       (uint4*)cache_v, (size_t)S * row16, (size_t)(dst + done) * row16, (size_t)(src + done) * row16,
                       n16);
  }
  return ll_check_launch("ll_kv_roll");
}

extern "C" int ll_patchify(const ll_bf16* x, ll_bf16* patches, int B, int F, int Cin, int H, int W, ll_stream stream) {
  LL_REQUIRE(H % 2 == 0 && W % 2 == 0, "ll_patchify: H=%d, W=%d must be even", H, W);
  size_t total = (size_t)B * F * (H / 2) * (W / 2) * Cin * 2;
  if (total == 0) return LL_OK;
  hipLaunchKernelGGL(patchify_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)x, (bf16*)patches, B, F, Cin, H, W);
  return ll_check_launch("ll_patchify");
}

extern "C" int ll_sinusoid(const float* t, ll_bf16* out, int n, int dim, ll_stream stream) {
  LL_REQUIRE(dim % 2 == 0, "ll_sinusoid: dim=%d must be even", dim);
  int total = n * (dim / 2);
  if (total == 0) return LL_OK;
  hipLaunchKernelGGL(sinusoid_kernel, dim3((total + 127) / 128), dim3(128), 0, (hipStream_t)stream, t, (bf16*)out, n, dim);
  return ll_check_launch("ll_sinusoid");
}

extern "C" int ll_unpatchify_x0(const ll_bf16* head, const ll_bf16* xt, const float* sigma, ll_bf16* flow, ll_bf16* x0,
                                int B, int F, int Cout, int H, int W, ll_stream stream) {
  LL_REQUIRE(H % 2 == 0 && W % 2 == 0, "ll_unpatchify_x0: H=%d, W=%d must be even", H, W);
  size_t total = (size_t)B * F * Cout * H * (W / 2);
  if (total == 0) return LL_OK;
  hipLaunchKernelGGL(unpatchify_x0_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)head, (const bf16*)xt, sigma, (bf16*)flow, (bf16*)x0, B, F, Cout, H, W);
  return ll_check_launch("ll_unpatchify_x0");
}

extern "C" int ll_add_noise(const ll_bf16* x0, const ll_bf16* noise, const float* sigma, ll_bf16* out, int N,
                            long long inner, ll_stream stream) {
  LL_REQUIRE(inner % 8 == 0, "ll_add_noise: inner=%lld must be a multiple of 8", inner);
  if (N == 0 || inner == 0) return LL_OK;
  long long inner8 = inner / 8;
  hipLaunchKernelGGL(add_noise_kernel, dim3((unsigned)((inner8 + 255) / 256), N), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)x0, (const bf16*)noise, sigma, (bf16*)out, inner8);
  return ll_check_launch("ll_add_noise");
}

extern "C" int ll_sigma_lookup(const float* t, const float* timesteps, const float* sigmas, float* out, int n,
                               int n_table, ll_stream stream) {
  LL_REQUIRE(n_table > 0, "ll_sigma_lookup: empty table");
  if (n == 0) return LL_OK;
  hipLaunchKernelGGL(sigma_lookup_kernel, dim3(n), dim3(64), 0, (hipStream_t)stream, t, timesteps, sigmas, out, n_table);
  return ll_check_launch("ll_sigma_lookup");
}

// ---- synthetic data (bench.py / tests: there are no checkpoints or datasets here) -----------------------------------------------
// out[i] = hash_normal / hash_uniform of counter lo + i under `stream_const` -- the same integers longlive_amd/synth.py evaluates
// with int64 tensor ops, so the bits are identical on any host and on the device.  In the library because torch's int64
// elementwise kernels are what a python process under `rocprofv3 --pmc` dies in on this image (profiles/r03_pmc_inpipe.md,
// DESIGN.md): with this kernel bench.py issues none of them.
#include "synth_hash.h"
__global__ __launch_bounds__(256) void synth_hash_kernel(float* __restrict__ out, unsigned long long lo, long long n,
                                                         unsigned long long stream_const, int kind) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint64_t idx = (uint64_t)lo + (uint64_t)i;
  out[i] = kind == 0 ? ll_synth_uniform(idx, (uint64_t)stream_const) : ll_synth_normal(idx, (uint64_t)stream_const);
}

extern "C" int ll_synth_hash(float* out, long long lo, long long n, unsigned long long stream_const, int kind, ll_stream stream) {
  LL_REQUIRE(kind == 0 || kind == 1, "ll_synth_hash: kind %d (0 = uniform, 1 = normal)", kind);
  LL_REQUIRE(n >= 0 && lo >= 0 && (n + 255) / 256 < (1ll << 31), "ll_synth_hash: bad range lo=%lld n=%lld", lo, n);
  if (n == 0) return LL_OK;
  LL_REQUIRE(out != nullptr, "ll_synth_hash: out is NULL");
  hipLaunchKernelGGL(synth_hash_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out,
                     (unsigned long long)lo, n, stream_const, kind);
  return ll_check_launch("ll_synth_hash");
}
