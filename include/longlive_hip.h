/*
 * longlive_hip.h -- C ABI of liblonglive_hip.so: the MI355X (gfx950) native replacement for the per-frame
 * denoising hot path of LongLive (reference: kpham-augment/LongLive).
 *
 * The reference has no FFI/plugin registry: its "operator API" on this path is the set of PyTorch module calls made
 * inside CausalWanModel._forward_inference.  Each entry point below replaces one of those call sites (file:line of
 * the reference is cited per function); the Python host layer (longlive_amd/) mirrors the reference's Python
 * interface on top of this ABI.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc / torch allocation); bf16 tensors are `uint16_t` bit patterns;
 *   - every function is asynchronous on `stream` (a hipStream_t passed as void*), never allocates, never syncs;
 *   - every function returns 0 on success, a negative ll_status otherwise; ll_last_error() gives the message;
 *   - "rows" are tokens (latent patches); C = model width; token order inside a forward is (frame, h, w).
 */
#ifndef LONGLIVE_HIP_H
#define LONGLIVE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint16_t ll_bf16;
typedef void* ll_stream;

enum ll_status {
  LL_OK = 0,
  LL_ERR_INVALID_ARG = -1,   /* shape / alignment precondition violated (message names it) */
  LL_ERR_LAUNCH = -2,        /* hipLaunchKernel failed */
  LL_ERR_UNSUPPORTED = -3
};

/* GEMM epilogues (ll_gemm_bf16). */
enum ll_epilogue {
  LL_EPI_BIAS = 0,           /* out = bf16(acc + bias)                                         nn.Linear */
  LL_EPI_BIAS_GELU = 1,      /* out = bf16(gelu_tanh(bf16(acc + bias)))                        ffn.0 + GELU */
  LL_EPI_BIAS_GATE_RES = 2,  /* out = res + bf16(bf16(acc+bias) * gate[frame(row)])            causal_model.py:456,467 */
  LL_EPI_BIAS_RES = 3        /* out = res + bf16(acc + bias)                                   causal_model.py:460 */
};

/* ABI version of this header.  Any change to an existing signature or the removal of an entry point bumps it; the Python binding
 * (longlive_amd/_lib.py) and any other caller must find ll_version() == LL_ABI_VERSION or refuse the library: a stale .so would
 * otherwise shift `stream` and the pointers silently.  100 = rounds 1-3; 105 = round 4's removals (workspace arguments of
 * ll_flash_attn, the split-K hand-off entry points); 106 = round 5 (ll_conv_cl_rms added). */
#define LL_ABI_VERSION 106
int ll_version(void);
const char* ll_last_error(void);
/* Development knob for A/B timing of kernel variants (tools/kbench, tools/kenergy, LL_TUNING=key=value,... for bench.py);
 * the defaults are the shipped configuration; every key names a path some shipped call can take.  Keys:
 *   "gemm_asm"  bit 0 = the generated one-wave-per-SIMD GEMM kernels where they cover the call (gemm_asm_224_gelu: FFN1;
 *               gemm_asm_192_bias: QKV with its V-cache redirect; gemm_asm_128_*: N <= 2048 with bias / gate-residual / residual),
 *               bits 2 / 3 = leave the GELU / the 128-wide kernels out, bit 4 (16) = ll_gemm_w8a8 / ll_gemm_w8a8_qkv on the
 *               generated W8A8 kernels too (bit-identical results, measured slower end to end), bit 5 (32) = launches with more
 *               tiles than CUs run the PERSISTENT form (gemm_asmp_*: one workgroup per CU walks its tiles and stages the next
 *               tile's first pieces under the current epilogue; bit-identical results); default 35; 0 = HIP kernels only
 *   "gemm_variant" / "gemm_variant_wide" (N >= 4096 only)  tile of the HIP kernels (int8, embeddings / head, gemm_asm = 0):
 *               0 = auto (cost model), 2 = 256x128, 3 = 256x256, 5 = 256x192, 6 = 256x224
 *   "gemm_group_m"  m-tiles per group of the GEMM tile walk (default 4; <= 1: N fastest);  "gemm_lds_epi" 0 / 1 / 2 = HIP epilogues
 *               staged through LDS: none / all but GELU (default) / all
 *   "attn_asm"  1 (default) = the generated attention kernel (flash_attn_asm_kernel) for single key ranges of at least
 *               "attn_asm_min_keys" keys (default 512: self- and cross-attention), 0 = the HIP kernels
 *   "attn_variant"  HIP attention: 0 = plain kernel, 1 = software-pipelined, 2 = + ping-pong wave groups from "attn_pp_min_keys"
 *               keys on (default 2);  "attn_xcd" 0 / 1 = XCD-aware workgroup placement (default 1)
 *   "conv_halo" 0 / 1 = halo-tile convolution kernel of the VAE decoder (default 1)
 * Unknown key: LL_ERR_INVALID_ARG. */
int ll_set_tuning(const char* key, int value);
/* Host-only introspection: the kernel instance + tile + grid that ll_gemm_bf16 / ll_gemm_w8a8 / ll_flash_attn would launch
 * for a shape under the current tuning, as text in out[cap] (bench.py's per-kernel table names kernels from here). */
int ll_gemm_plan(int M, int N, int K, int int8, char* out, int cap);
/* The same for a call whose epilogue is known (LL_EPI_*): names the generated one-wave-per-SIMD kernel (gemm_asm_*, tuning key
 * "gemm_asm") where ll_gemm_bf16 / ll_gemm_w8a8 takes it, else ll_gemm_plan's text.  plain = 1: an ordinary call (no V-cache
 * output, no per-batch modulation vector); 2: the fused QKV call (ll_gemm_bf16_qkv, one batch element); 0: a call with a
 * modulation vector. */
int ll_gemm_plan_epi(int M, int N, int K, int int8, int epilogue, int plain, char* out, int cap);
int ll_flash_attn_plan(int Lq, int H, int B, int seg0_len, int seg1_len, int seg_adjacent, char* out, int cap);

/* ---- norms / modulation ------------------------------------------------------------------------------------- */

/* out[r,:] = LN(x[r,:]) * (1 + s) + t  with s = bf16(mod[scale_idx,:] + e[b,f,scale_idx,:]), t likewise, f = frame of
 * row r.  Replaces `norm1(x).unflatten(..) * (1 + e[1]) + e[0]` (wan/modules/causal_model.py:445,463-464) and
 * CausalHead's 2-way form (:506-507).  x,out [B, L, C]; e [B, F, nmod, C]; mod [nmod, C]; L = F * frame_len. */
int ll_ln_modulate(const ll_bf16* x, ll_bf16* out, const ll_bf16* e, const ll_bf16* mod, int nmod, int shift_idx,
                   int scale_idx, int B, int L, int C, int F, float eps, ll_stream stream);

/* out[l, bf, i, :] = bf16(mods[l, i, :] + e[bf, i, :]) for all layers l of one forward: `e = modulation + e0`
 * (wan/modules/causal_model.py:440) evaluated once per (layer, frame) instead of once per token row.  A layer's slice
 * out[l] [B*F, nmod, C] may be passed as `e` with mod = NULL to ll_ln_modulate / ll_ln_modulate_q8 and to the gate-residual
 * epilogue of ll_gemm_bf16 / ll_gemm_w8a8: same values, two vector loads and six operations per element fewer. */
int ll_modulation_table(const ll_bf16* e, const ll_bf16* mods, ll_bf16* out, int num_layers, int BF, int nmod, int C,
                        ll_stream stream);

/* The fp32 form of the table for ll_ln_modulate_tab: out[l, bf, i, :] = float(bf16(mods[l, i, :] + e[bf, i, :])), and for the chunks
 * whose bit is set in one_plus_mask (the scale chunks 1 and 4 of a block's six) float(bf16(1 + that)) -- exactly the `1 + e[1]` the
 * reference forms per token row (wan/modules/causal_model.py:445,463), once per (layer, frame).  out [num_layers, BF, nmod, C] fp32. */
int ll_modulation_table_f32(const ll_bf16* e, const ll_bf16* mods, float* out, int num_layers, int BF, int nmod, int C,
                            unsigned one_plus_mask, ll_stream stream);

/* ll_ln_modulate / ll_ln_modulate_q8 from a layer's slice tab [B*F, nmod, C] of ll_modulation_table_f32 (scale_idx must be a chunk
 * of its one_plus_mask): out = bf16(bf16(bf16(LN(x)) * tab[scale_idx]) + tab[shift_idx]), the same bits with eight vector
 * instructions per element pair fewer.  Exactly one of out (bf16 [B, L, C]) and q (int8 [B, L, C] with qscale [B*L]) is non-NULL. */
int ll_ln_modulate_tab(const ll_bf16* x, ll_bf16* out, int8_t* q, float* qscale, const float* tab, int nmod, int shift_idx,
                       int scale_idx, int B, int L, int C, int F, float eps, ll_stream stream);

/* out = LayerNorm(x) * w + b  (norm3, wan/modules/causal_model.py:397-399,460; wan/modules/model.py:89-99). */
int ll_layernorm_affine(const ll_bf16* x, const ll_bf16* w, const ll_bf16* b, ll_bf16* out, int rows, int C,
                        float eps, ll_stream stream);

/* int8 mode: the same two kernels emitting per-row symmetric int8 + scale (bit-identical to ll_quantize_rows applied to
 * their bf16 output) so the following W8A8 GEMM reads its operand without an extra pass. */
int ll_ln_modulate_q8(const ll_bf16* x, int8_t* q, float* qscale, const ll_bf16* e, const ll_bf16* mod, int nmod,
                      int shift_idx, int scale_idx, int B, int L, int C, int F, float eps, ll_stream stream);
int ll_layernorm_affine_q8(const ll_bf16* x, const ll_bf16* w, const ll_bf16* b, int8_t* q, float* qscale, int rows,
                           int C, float eps, ll_stream stream);

/* out = bf16(x * rsqrt(mean(x^2) + eps)) * w over the FULL width C (WanRMSNorm, wan/modules/model.py:70-86).
 * ldx / ldo = row strides in elements (lets the caller normalise a column slice of a fused projection). */
int ll_rmsnorm(const ll_bf16* x, const ll_bf16* w, ll_bf16* out, int rows, int C, int ldx, int ldo, float eps,
               ll_stream stream);

/* Fused q/k RMSNorm + 3-axis RoPE + KV-cache insert for self-attention
 * (wan/modules/causal_model.py:122-126,206-211,264-269,302-311; causal_rope_apply :32-60).
 *   qkv      [B, L, 3C]  fused projection output (q | k | v)
 *   q_out    [B, L, C]   roped queries
 *   cache_k/v[B, S, C]   rows [write_start, write_start+write_len) receive roped k / v of tokens
 *                        [roped_offset, roped_offset+write_len)
 *   rope_f   [1024, nf, 2] (cos,sin) fp32 for the frame axis, rope_hw [frame_len, nhw, 2] for the (h,w) axes;
 *            nf + nhw = head_dim / 2.  start_frame = current_start / frame_len.
 *   cache_v may be NULL: V was already inserted by ll_gemm_bf16_qkv / ll_gemm_w8a8_qkv and the v third of qkv is not read. */
int ll_qk_norm_rope_kv_store(const ll_bf16* qkv, const ll_bf16* wq, const ll_bf16* wk, const float* rope_f,
                             const float* rope_hw, ll_bf16* q_out, ll_bf16* cache_k, ll_bf16* cache_v, int B, int L,
                             int C, int head_dim, int frame_len, int start_frame, int S, int write_start,
                             int roped_offset, int write_len, float eps, ll_stream stream);

/* cache[:, dst:dst+n] = cache[:, src:src+n] for K and V, src > dst (left shift that discards evicted tokens while
 * the sink stays put: wan/modules/causal_model.py:257-260, 874-877).  Overlap-safe (chunked by src-dst). */
int ll_kv_roll(ll_bf16* cache_k, ll_bf16* cache_v, int B, int S, int C, int dst, int src, int n, ll_stream stream);

/* ---- dense contractions (MFMA) -------------------------------------------------------------------------------- */

/* out[M,N] = epilogue(x[M,K] @ w[N,K]^T + bias[N]); bf16 in/out, fp32 accumulate.  K % 64 == 0, N % 8 == 0.
 * Replaces nn.Linear q/k/v/o, ffn.0/ffn.2, text_embedding, patch_embedding (as GEMM), head.head
 * (wan/modules/causal_model.py:90-93,406-408,599-603; wan/modules/model.py:172-193).
 * res [M, ldo] may alias out.  For LL_EPI_BIAS_GATE_RES: gate = bf16(mod[gate_idx,:] + e[b, f, gate_idx, :]) with
 * b = row / rows_per_batch, f = (row % rows_per_batch) / frame_len; e [B, F, nmod, N], mod [nmod, N]; mod = NULL: e already
 * holds that sum (ll_modulation_table). */
int ll_gemm_bf16(const ll_bf16* x, const ll_bf16* w, const ll_bf16* bias, ll_bf16* out, int M, int N, int K, int ldx,
                 int ldo, int epilogue, const ll_bf16* res, const ll_bf16* e, const ll_bf16* mod, int nmod,
                 int gate_idx, int rows_per_batch, int frame_len, ll_stream stream);

/* Small-M form of ll_gemm_bf16 for the 512-token linears of the text side (umT5 self-attention / FFN projections,
 * wan/modules/t5.py:65-117,134-160; the text K/V projections of cross-attention, wan/modules/model.py:183-188): a grid of
 * ceil(M / 256) x (N / 128) tiles fills a fraction of the device, so K is cut into ll_gemm_ksplit_plan(M, N, K) ranges (0 = not
 * taken for this shape on this device), each range's fp32 tile sums go to `workspace` ([splits][M][N] floats, caller-owned, at
 * least ll_gemm_ksplit_workspace_bytes bytes, 16-byte aligned, no initialisation needed) and one pass adds them in a fixed order
 * and applies the epilogue (LL_EPI_BIAS or LL_EPI_BIAS_RES).  Results equal ll_gemm_bf16's up to the order of the fp32 sum;
 * bit-identical run to run.  workspace = NULL or plan = 0: it IS ll_gemm_bf16.  The NUMBER of ranges depends on N and K only;
 * WHETHER the path is taken depends on M and on the device's CU count (tiles * 2 <= CUs) and on tuning key "gemm_asm"
 * (bit 0 off or bit 3 set: never), so a row's fp32 sum order is stable only inside that region. */
int ll_gemm_ksplit_plan(int M, int N, int K);
long long ll_gemm_ksplit_workspace_bytes(int M, int N, int K);
int ll_gemm_bf16_ksplit(const ll_bf16* x, const ll_bf16* w, const ll_bf16* bias, ll_bf16* out, int M, int N, int K, int ldx,
                        int ldo, int epilogue, const ll_bf16* res, void* workspace, long long workspace_bytes, ll_stream stream);
/* ll_gemm_bf16_ksplit (LL_EPI_BIAS_RES) followed by T5LayerNorm of the new residual stream (wan/modules/t5.py:57-63,119-160:
 * x = x + linear(.); h = norm(x)): out = x_new [M, N] (ldo = N), h_out = bf16(norm_w * bf16(x_new * rsqrt(mean(x_new^2) + eps))).
 * On the small-M path the K-range sum, bias, residual and norm are one pass over each row; otherwise ll_gemm_bf16 +
 * ll_t5_rmsnorm.  The two outputs carry the same bits either way. */
int ll_gemm_bf16_ksplit_t5norm(const ll_bf16* x, const ll_bf16* w, const ll_bf16* bias, ll_bf16* out, int M, int N, int K, int ldx,
                               int ldo, const ll_bf16* res, const ll_bf16* norm_w, float eps, ll_bf16* h_out, void* workspace,
                               long long workspace_bytes, ll_stream stream);

/* W8A8 variant of ll_gemm_bf16 for BASELINE config 5 ("INT8-quantized linear layers"; the reference ships no INT8 code,
 * reports.md:24,39): out = epilogue(sx[m] * sw[n] * (xq[M,K] . wq[N,K]^T) + bias) with int8 operands, exact int32
 * accumulation on v_mfma_i32_16x16x64_i8 and the same fused epilogues.  sx [M] / sw [N] fp32; K % 128 == 0. */
int ll_gemm_w8a8(const int8_t* xq, const float* sx, const int8_t* wq, const float* sw, const ll_bf16* bias, ll_bf16* out,
                 int M, int N, int K, int ldo, int epilogue, const ll_bf16* res, const ll_bf16* e, const ll_bf16* mod,
                 int nmod, int gate_idx, int rows_per_batch, int frame_len, ll_stream stream);

/* The fused q|k|v projection (N = 3 C, LL_EPI_BIAS) with the V third written straight into the KV cache by the GEMM epilogue:
 * token t of batch b goes to cache_v[b, write_start + (t - roped_offset), :] when 0 <= t - roped_offset < write_len, exactly the
 * insert of wan/modules/causal_model.py:264-269,302-311 (V is copied unrotated, so the bits are the projection's).  The q and k
 * thirds land in out[M, ldo]; its v third is left unwritten.  M = B * L.  Saves one write and one read of V per layer. */
int ll_gemm_bf16_qkv(const ll_bf16* x, const ll_bf16* w, const ll_bf16* bias, ll_bf16* out, int M, int N, int K, int ldx, int ldo,
                     ll_bf16* cache_v, int B, int L, int S, int write_start, int roped_offset, int write_len, ll_stream stream);
int ll_gemm_w8a8_qkv(const int8_t* xq, const float* sx, const int8_t* wq, const float* sw, const ll_bf16* bias, ll_bf16* out,
                     int M, int N, int K, int ldo, ll_bf16* cache_v, int B, int L, int S, int write_start, int roped_offset,
                     int write_len, ll_stream stream);

/* Symmetric per-row int8 quantisation: scale[r] = max|x[r,:]| / 127 (1 for an all-zero row), q = rint(x / scale).
 * Per token for activations, per output channel for weights ([N,K] rows), x row stride ldx elements. */
int ll_quantize_rows(const ll_bf16* x, int8_t* q, float* scale, int rows, int K, int ldx, ll_stream stream);

/* Small-M linear (M <= 8): out = act_out(act_in(x) @ w^T + b); act: 0 none, 1 SiLU.  time_embedding /
 * time_projection (wan/modules/causal_model.py:605-608,976-979). */
int ll_linear_small(const ll_bf16* x, const ll_bf16* w, const ll_bf16* bias, ll_bf16* out, int M, int N, int K,
                    int act_in, int act_out, ll_stream stream);

/* Non-causal softmax(q k^T / sqrt(d)) v, head_dim 128, keys taken from up to two row ranges of the K/V buffers
 * ([seg0_start, +seg0_len) then [seg1_start, +seg1_len)): frame sink + sliding window of the KV cache, or the 512
 * text tokens for cross-attention.  Replaces attention()/flash_attention() (wan/modules/attention.py:43-197) and the
 * sink/window gather + cat (wan/modules/causal_model.py:331-360).
 *   q,out [B, Lq, H*128] with row strides ldq/ldo; k,v [B, Sk, H*128] with row stride ldk, batch stride k_batch_stride.
 *   A single key range of >= 512 keys runs the generated kernel (flash_attn_asm_kernel: 4 waves x 64 query rows, one wave per
 *   SIMD), shorter or two-piece ranges the HIP kernels. */
int ll_flash_attn(const ll_bf16* q, const ll_bf16* k, const ll_bf16* v, ll_bf16* out, int B, int Lq, int H, int ldq,
                  int ldo, int ldk, long long k_batch_stride, int seg0_start, int seg0_len, int seg1_start,
                  int seg1_len, float scale, ll_stream stream);

/* Cross-attention's q path with two launches fewer: `q = self.norm_q(self.q(x))` then attention over the text keys
 * (wan/modules/model.py:172,189; WanRMSNorm :78-86 normalises over ALL H*128 channels of a row).
 *   ll_gemm_bf16_ssq   = ll_gemm_bf16 with LL_EPI_BIAS that also writes ssq[N / 128][M] (fp32): per row and 128-column n-tile the
 *                        sum of squares of the bf16 outputs (the projection's epilogue; fixed summation order).
 *   ll_flash_attn_qnorm = ll_flash_attn over ONE key range whose Q prologue sums the planes in plane order, forms
 *                        rsq(sum / C + eps) and applies bf16(bf16(x * rinv) * norm_w) -- WanRMSNorm's rounding points -- before its
 *                        own scaling: the separate RMSNorm launch (and its 2 x 14.4 MB of traffic) disappears.
 * Both exist only as generated gfx950 kernels: ll_gemm_ssq_planes (planes written, 0 = not covered) and ll_flash_attn_qnorm_ok
 * (1 / 0) say whether a shape is covered under the current tuning; otherwise the caller runs ll_gemm_bf16 + ll_rmsnorm +
 * ll_flash_attn (the entry points return LL_ERR_INVALID_ARG rather than fall back).  q [B, Lq, H*128] contiguous rows (ldq = H*128),
 * ssq rows are b * Lq + l. */
int ll_gemm_ssq_planes(int M, int N, int K);
int ll_gemm_bf16_ssq(const ll_bf16* x, const ll_bf16* w, const ll_bf16* bias, ll_bf16* out, float* ssq, int M, int N, int K, int ldx,
                     int ldo, ll_stream stream);
int ll_flash_attn_qnorm_ok(int H, int nkeys);
int ll_flash_attn_qnorm(const ll_bf16* q, const float* ssq, const ll_bf16* norm_w, float eps, const ll_bf16* k, const ll_bf16* v,
                        ll_bf16* out, int B, int Lq, int H, int ldq, int ldo, int ldk, long long k_batch_stride, int key_start,
                        int nkeys, float scale, ll_stream stream);

/* ---- embeddings / head / scheduler ------------------------------------------------------------------------------ */

/* im2col of Conv3d k=s=(1,2,2) (wan/modules/causal_model.py:599-600,959-963): x [B,F,Cin,H,W] (the wrapper's layout,
 * utils/wan_wrapper.py:249 permute folded in) -> patches [B, F*(H/2)*(W/2), Cin*4] with column (c, p, q). */
int ll_patchify(const ll_bf16* x, ll_bf16* patches, int B, int F, int Cin, int H, int W, ll_stream stream);

/* sinusoidal_embedding_1d (wan/modules/model.py:15-25) in fp64 then cast to bf16: out [n, dim]. */
int ll_sinusoid(const float* t, ll_bf16* out, int n, int dim, ll_stream stream);

/* unpatchify ('fhwpqrc->cfphqwr', wan/modules/causal_model.py:1240-1263) + flow->x0 in fp64
 * (utils/wan_wrapper.py:175-199): head [B, F*h*w, 4*Cout] -> flow, x0 [B,F,Cout,H,W]; x0 = xt - sigma[b,f]*flow. */
int ll_unpatchify_x0(const ll_bf16* head, const ll_bf16* xt, const float* sigma, ll_bf16* flow, ll_bf16* x0, int B,
                     int F, int Cout, int H, int W, ll_stream stream);

/* FlowMatchScheduler.add_noise (utils/scheduler.py:159-176): out = bf16((1-sigma)*x0 + sigma*noise), fp32,
 * sigma[n] per leading index; x0/noise/out [N, inner]. */
int ll_add_noise(const ll_bf16* x0, const ll_bf16* noise, const float* sigma, ll_bf16* out, int N, long long inner,
                 ll_stream stream);

/* out[i] = sigmas[argmin_j |timesteps[j] - t[i]|]: the table lookup shared by flow->x0 and add_noise
 * (utils/wan_wrapper.py:195-197; utils/scheduler.py:172-174).  All fp32 device arrays; lowest index wins ties. */
int ll_sigma_lookup(const float* t, const float* timesteps, const float* sigmas, float* out, int n, int n_table,
                    ll_stream stream);

/* Synthetic data for bench.py / the tests (no reference call site: the reference loads checkpoints, inference.py:72-94, and draws
 * noise with torch.randn, :193-195): out[i] (fp32, device) = the counter hash of longlive_amd/synth.py at counter lo + i under
 * stream_const -- kind 0: uniform [0, 1) with 24 bits, kind 1: Irwin-Hall normal.  Bit-identical to synth.hash_uniform /
 * hash_normal on any host (csrc/synth_hash.h is compiled for both sides; tests/test_synth_hash.py). */
int ll_synth_hash(float* out, long long lo, long long n, unsigned long long stream_const, int kind, ll_stream stream);

/* ---- VAE decoder (wan/modules/vae.py, utils/wan_wrapper.py:83-116) ------------------- */

/* CausalConv3d 3x3x3 / (3,1,1) / 1x1x1 and Conv2d 3x3 / 1x1 (wan/modules/vae.py:17-36; the Upsample + Conv2d pair of
 * Resample, vae.py:74-84, with upsample=1) as one implicit GEMM on channels-last activations:
 *   out[(t,ho,wo), co] = bias[co] (+ res[(t,ho,wo), co]) + sum x[t+kt-(KT-1), (ho+kh-p)>>up, (wo+kw-p)>>up, ci] * w[co, (kt,kh,kw), ci]
 * x points at the first of T new frames [T,H,W,Cin]; when KT == 3 the two frames BEFORE it in memory (x - 2*H*W*Cin
 * elements) must hold the previous two input frames of the stream -- the reference's feat_cache (vae.py:29-34) -- or
 * zeros at the start of a stream.  zero16 = 16 zero bytes on the device (source of spatial / K padding);
 * w [Cout, Kpad] bf16 with k = ((kt*KH + kh)*KH + kw)*Cin + ci, zero padded to Kpad = ceil(KT*KH*KH*Cin / 64) * 64;
 * out/res rows of ldo elements, output spatial size (H<<up, W<<up). */
int ll_conv_cl(const ll_bf16* x, const ll_bf16* zero16, const ll_bf16* w, const ll_bf16* bias, const ll_bf16* res,
               ll_bf16* out, int T, int H, int W, int Cin, int Cout, int Kpad, int KT, int KH, int upsample, int ldo,
               ll_stream stream);
/* ll_conv_cl whose epilogue also applies the RMS_norm (+ SiLU) the decoder runs on that tensor next (ResidualBlock: conv -> RMS_norm
 * -> SiLU -> conv, wan/modules/vae.py:193-220; rounding points of ll_rms_silu_cl): out_rms [pixels, Cout] = rms_silu(out), written by
 * the same launch; `out` (the un-normalised tensor, with the residual when `res`) may be NULL when nothing else reads it.  Only for
 * convolutions whose workgroup holds every channel of a pixel -- ll_conv_cl_rms_ok(...) = 1: the halo-tile kernel with Cout = 96 (the
 * 480x832 stage, 70 % of the decoder's FLOPs); otherwise run ll_conv_cl + ll_rms_silu_cl.  ldo = Cout. */
int ll_conv_cl_rms_ok(int H, int W, int Cin, int Cout, int KT, int KH, int upsample);
int ll_conv_cl_rms(const ll_bf16* x, const ll_bf16* zero16, const ll_bf16* w, const ll_bf16* bias, const ll_bf16* res, ll_bf16* out,
                   const ll_bf16* rms_gamma, ll_bf16* out_rms, int rms_silu, int T, int H, int W, int Cin, int Cout, int Kpad, int KT,
                   int KH, int upsample, int ldo, ll_stream stream);

/* RMS_norm over channels (+ SiLU) (wan/modules/vae.py:39-55,193-197) on channels-last rows with the reference's bf16
 * rounding points: n = bf16(||x||); y = bf16(bf16(bf16(x / max(n, 1e-12)) * sqrt(C)) * gamma); out = silu(y) if do_silu. */
int ll_rms_silu_cl(const ll_bf16* x, const ll_bf16* gamma, ll_bf16* out, long long pixels, int C, int do_silu,
                   ll_stream stream);

/* p[:, :N] = softmax(scale * s[:, :N]) along rows of ld elements, p[:, N:ld] = 0 (the decoder's single-head attention,
 * F.scaled_dot_product_attention at vae.py:249-254, as GEMM + softmax + GEMM). */
int ll_softmax_rows(const ll_bf16* s, ll_bf16* p, int rows, int N, int ld, float scale, ll_stream stream);

/* Latent un-scaling z / (1/std) + mean in bf16 (utils/wan_wrapper.py:99-110, wan/modules/vae.py:548-550) fused with the
 * layout change [T, C, h, w] -> channels-last [T, h, w, C]. */
int ll_vae_unscale_cl(const ll_bf16* z, const ll_bf16* mean, const ll_bf16* inv_std, ll_bf16* out, int T, int C, int h,
                      int w, ll_stream stream);

/* Decoder output: channels-last bf16 [T,H,W,ldc] (first 3 channels) -> fp32 [T,3,H,W] clamped to [-1,1]
 * (wan/modules/vae.py decode's clamp_ + utils/wan_wrapper.py:112-116). */
int ll_cl_to_tchw_clamp(const ll_bf16* x, float* out, int T, int H, int W, int ldc, ll_stream stream);

/* ---- umT5 text encoder (wan/modules/t5.py, utils/wan_wrapper.py:16-57) ----------------- */

/* T5LayerNorm.forward (wan/modules/t5.py:57-63): out = bf16(w * bf16(x * rsqrt(mean(x^2) + eps))), fp32 statistics;
 * x, out [rows, C] contiguous, any C % 8 == 0 (4096 for umT5-xxl). */
int ll_t5_rmsnorm(const ll_bf16* x, const ll_bf16* w, ll_bf16* out, int rows, int C, float eps, ll_stream stream);

/* T5FeedForward's gated activation (t5.py:46-50,134-139): out[M,F] = bf16(h[:, F:2F] * GELU(h[:, 0:F])) with the
 * reference's python tanh-GELU evaluated op by op in bf16; h [M, 2F] = x @ [gate.0.weight; fc1.weight]^T. */
int ll_t5_gated_gelu(const ll_bf16* h, ll_bf16* out, long long M, int F, ll_stream stream);

/* token_embedding(ids) (t5.py:297): out[i, :] = table[ids[i], :]; ids int64 on the device, validated by the caller. */
int ll_gather_rows(const ll_bf16* table, const long long* ids, ll_bf16* out, int n, int C, long long vocab, ll_stream stream);

/* T5Attention.forward (t5.py:85-117), self-attention of one sequence, head_dim 64, L in {64,128,256,512} keys = queries:
 * out = bf16(softmax_fp32(bf16(bf16(q k^T) + bias), keys >= seq_len masked) v), no 1/sqrt(d) scaling.
 * q,k [L, ...] with row stride ldqk (head h at columns h*64..); vt [H*64, L] = V transposed (x Wv^T computed as
 * Wv x^T); bias_tab [H, 2L-1] with bias_tab[h, j - i + L - 1] = pos_embedding[bucket(j - i), h] (t5.py:219-263);
 * out [L, ...] row stride ldo. */
int ll_t5_attention(const ll_bf16* q, const ll_bf16* k, const ll_bf16* vt, const ll_bf16* bias_tab, ll_bf16* out, int L,
                    int H, int ldqk, int ldo, int seq_len, ll_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* LONGLIVE_HIP_H */
