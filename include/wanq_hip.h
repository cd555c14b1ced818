/*
 * wanq_hip.h -- C ABI of libwanq_hip.so: the MI355X (gfx950) hot path of the quantized Wan2.1 DiT.
 *
 * This is the drop-in boundary.  The reference has no C ABI: its "ABI" is two pybind11 modules
 * taking torch::Tensor (ViDiT-Q/kernels/csrc/{fused,qgemm}/pybind.cpp).  Every entry point below
 * names the reference function(s) it replaces; the Python package
 * wan2.1-quantization_amd/viditq_extension re-exports them under the reference's own names and
 * argument order through ctypes.
 *
 * Conventions
 *   - plain device pointers + sizes, no torch types; `stream` is a hipStream_t passed as void*
 *     (NULL = the legacy default stream).  Calls only enqueue work; nothing synchronises.
 *   - return value 0 = enqueued; non-zero = refused (WANQ_E_*), nothing enqueued, message from
 *     wanq_last_error().  Shape/dtype errors are reported, never asserted (the reference aborts the
 *     process on tile-divisibility violations: w8a8_gemm_cuda.cu:678-680).
 *   - row-major contiguous tensors; activations [rows, cols], weights [N, K] (nn.Linear layout).
 *   - dtype codes WANQ_F16/BF16/F32 for floating tensors; per-token / per-channel vectors
 *     (scale, sum, bias ...) carry their own dtype code so both the reference's fp16 buffers
 *     (K/viditq_extension/nn/base.py:12-26) and fp32 buffers (simulation-path accuracy) work.
 *   - integer results (int8 codes, int32 accumulators) are bit-exact w.r.t. oracle/; see DESIGN.md.
 */
#ifndef WANQ_HIP_H
#define WANQ_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { WANQ_F16 = 0, WANQ_BF16 = 1, WANQ_F32 = 2, WANQ_I32 = 3, WANQ_I16 = 4 };

enum {
  WANQ_OK = 0,
  WANQ_E_ARG = 1,     /* bad pointer / dtype / flag combination */
  WANQ_E_SHAPE = 2,   /* unsupported shape (message says which rule) */
  WANQ_E_LAUNCH = 3   /* hipGetLastError() after launch was not hipSuccess */
};

/* Last refusal message of the calling thread ("" if none).  Never NULL. */
const char* wanq_last_error(void);
/* Version of this ABI (bumped on any signature change). */
int wanq_abi_version(void);

/* ------------------------------------------------------------------------------------------------
 * Per-token dynamic int8 quantisation (+ dequantised row sum).
 *   q[r,c]   = clamp(rne(x[r,c] / scale_r), -128, 127),  scale_r = max(absmax_r / 127, 1e-6)
 *   sum[r]   = scale_r * sum_c q[r,c]                       (sum may be NULL)
 * act: 0 = identity, 1 = tanh-GELU applied to x first.
 * static_amax != 0: `scale` is an INPUT holding absmax_r (reference quant_sum_static reads it so).
 * Replaces fused.quant_sum / quant_sum_static / gelu_quant_sum
 *   (ViDiT-Q/kernels/csrc/fused/fused.cu:30-232, hosts :524-706) and is bit-identical to
 *   DynamicQuantizer.quantize (quant_utils/qdiff/base/base_quantizer.py:101-157) for any cols % 8 == 0
 *   (the reference needs cols % 128 == 0 and cols <= 8192). */
int wanq_quant_rows(const void* x, int x_dtype, int8_t* q, void* scale, void* sum, int vec_dtype,
                    int64_t rows, int cols, int act, int static_amax, void* stream);

/* The same per-token dynamic symmetric quantiser at a narrower range (codes still int8, the GEMMs are unchanged):
 *   q[r,c] = rne(x[r,c] / scale_r),  scale_r = max(absmax_r / n_levels, floor),  n_levels = 2^(b-1) - 1 in [1, 127]
 * floor = 0: no floor -- a row of zeros gets scale 0 and codes 0 (the reference divides 0 by 0 there and returns NaN).
 * Replaces MixedPrecisionDynamicQuantizer.quantize, symmetric branch, at the active entry of its bit-width list
 *   (quant_utils/qdiff/base/mixed_precision_quantizer.py:135-146,171-175: no eps floor) and DynamicQuantizer.quantize for
 *   n_bits < 8 (base/base_quantizer.py:116-128,154-157: floor 1e-6).  Bit-identical codes and scales. */
int wanq_quant_rows_levels(const void* x, int x_dtype, int8_t* q, void* scale, void* sum, int vec_dtype,
                           int64_t rows, int cols, int n_levels, float floor, void* stream);

/* ------------------------------------------------------------------------------------------------
 * LayerNorm (no bias, optional gamma) -> optional adaLN modulate -> fp output OR int8 quant (+sum).
 *   y = (x - mean) * rstd * gamma;  y = y * (1 + mscale[b]) + mshift[b]   (b = row / rows_per_batch)
 *   out_fp != NULL : write y in out_dtype.      q != NULL : quantise y like wanq_quant_rows.
 * gamma / mshift / mscale may be NULL (identity).  mod_stride = elements between batches.
 * Statistics and modulation are fp32 (simulation-path semantics, Wan model.py:327), not half2.
 * Replaces fused.layernorm_nobias, layernorm_nobias_quant_nosum_fuse, layernorm_nobias_quant_sum_fuse,
 *   layernorm_nobias_t2i_fuse, layernorm_nobias_t2i_quant_sum_fuse
 *   (ViDiT-Q/kernels/csrc/fused/fused.cu:234-380, hosts :485-522,:708-915).  Any cols % 8 == 0 up to
 *   16384 (the reference: cols/4 <= 1024 threads). */
int wanq_layernorm_rows(const void* x, int x_dtype, const void* gamma, const void* mshift,
                        const void* mscale, int mod_dtype, int64_t mod_stride, int64_t rows_per_batch,
                        float eps, void* out_fp, int out_dtype, int8_t* q, void* scale, void* sum,
                        int vec_dtype, int64_t rows, int cols, void* stream);

/* ------------------------------------------------------------------------------------------------
 * out = y * gate[b] + residual   (fp32 arithmetic; each tensor has its own dtype).
 * Replaces fused.gate_residual_fuse (fused.cu:382-483, host :917-961). */
int wanq_gate_residual(const void* y, int y_dtype, const void* gate, int gate_dtype, int64_t gate_stride,
                       const void* residual, int res_dtype, void* out, int out_dtype, int64_t rows,
                       int cols, int64_t rows_per_batch, void* stream);

/* ------------------------------------------------------------------------------------------------
 * W8A8 GEMM on int8 MFMA with fused dequant epilogue:
 *   acc[m,n] = sum_k a[m,k] * w[n,k]                                  (int32, exact)
 *   y        = acc*sa[m]*sw[n] (+ asum[m]*zp[n]*sw[n]) (+ bias[n])     (fp32, this order)
 *   y        = gelu_tanh(y)                     if WANQ_EPI_GELU
 *   y        = residual[m,n] + y * gate[n]      if WANQ_EPI_GATE_RES  (gate fp32[N], residual/out same dtype)
 *   out      = cast(y)  to out_dtype;  out_dtype == WANQ_I32 stores acc (no scales read).
 * sa/asum: per-token, dtype tok_dtype (F16|F32).  sw/bias: per-channel, dtype ch_dtype (F16|F32).
 * zp: zp_dtype WANQ_I16 (reference buffer) or WANQ_F32; NULL = symmetric weights.  bias may be NULL.
 * Any M >= 1 (ragged last tile handled in-kernel, no host padding: replaces pad_to_multiple_2d,
 * wan/quant_wanx_cuda.py:313-328); N % 8 == 0; K % 16 == 0.
 * Replaces qgemm.w8a8_of16_bias_weight_asym / _sym / w8a8_o32 / w8a8_of16_nobias_weight_sym_qserve
 *   (ViDiT-Q/kernels/csrc/qgemm/w8a8/w8a8_gemm_cuda.cu:624-838,
 *    ViDiT-Q/kernels/csrc/qgemm/w8a8/w8a8_gemm_cuda_qserve.cu:527-611). */
enum { WANQ_EPI_GELU = 1, WANQ_EPI_GATE_RES = 2 };
int wanq_gemm_w8a8(const int8_t* a, const int8_t* w, void* out, int out_dtype, const void* sa,
                   const void* asum, int tok_dtype, const void* sw, const void* bias, int ch_dtype,
                   const void* zp, int zp_dtype, const float* gate, const void* residual, int epi_flags,
                   int64_t M, int N, int K, void* stream);

/* Diagnostic: which int8 GEMM kernel wanq_gemm_w8a8 / wanq_gemm_w4a8 launch.  0 = automatic (default: the ping-pong persistent
 * kernel where a problem is eligible, else the persistent kernel, else the 128 x 128 kernel), 1 = always the 128 x 128
 * kernel, 2 = never the ping-pong kernel.  Process-wide; returns the previous setting (or -1 for a bad value, nothing changed).
 * Accumulators are bit-identical across the three kernels and so are the outputs of the two persistent ones; the 128 x 128
 * kernel sums the epilogue terms in another order (one unit of the output type; tests/test_gpu_gemm.py).  The environment variables
 * WANQ_GEMM_V1=1 / WANQ_GEMM_PP=0 set the same thing at start-up for whole-step A/B runs.  No reference counterpart. */
int wanq_gemm_select_kernel(int which);

/* ------------------------------------------------------------------------------------------------
 * PTQ calibration reduction: running per-channel absmax over tokens,
 *   colmax[c] = max(colmax[c], max_r |x[r,c]|)        (colmax fp32[cols], caller zero-initialises)
 * Replaces SaveActivationHook.__call__ default branch
 *   (ViDiT-Q/examples/Wan2.1/get_calib_data_wanx.py:262-267) -- and, because ptq takes .max(dim=0)
 *   over the stacked calls (ptq_wanx.py:336), the stack itself. */
int wanq_col_absmax(const void* x, int x_dtype, float* colmax, int64_t rows, int cols, void* stream);

/* v fake-quantisation of the reference's quantized attention (ViDiT-Q/examples/Wan2.1/models/quant_opensora.py:438-440:
 * DynamicQuantizer over all tokens for every (head, channel)), on the token-major [rows, cols] tensor: per COLUMN c
 *   delta_c = max(colmax[c] / (2^(b-1) - 1), 1e-6),   out = clamp(rne(x / delta_c), -2^(b-1), 2^(b-1) - 1) * delta_c.
 * colmax: fp32 [cols] from wanq_col_absmax over the same rows.  In place (out == x) is allowed. */
int wanq_fake_quant_cols(const void* x, int x_dtype, const float* colmax, void* out, int out_dtype, int n_bits,
                         int64_t rows, int cols, void* stream);

/* Fake-quantisation with a PRECOMPUTED delta of x's own shape, elementwise over n values (n % 8 == 0):
 *   d = max(delta, 1e-6);  plain (bits == NULL): d' = d / (2^b - 1), out = clamp(rne(x / d'), 0, 2^b - 1) * d'
 *   bits != NULL (int32 per element): d' = d / (2^bits - 1), out = min(rne(x / d'), 2^bits - 1) * d', 0 bits -> 0.
 * Replaces DynamicQuantizer.forward_with_quant_params (quant_utils/qdiff/base/base_quantizer.py:164-206: the fake-quant step of
 *   the reference's block-wise attention-map quantisers, symmetric quantisers only).  Bit-identical for fp32 x.  In place allowed. */
int wanq_fake_quant_with_delta(const void* x, int x_dtype, const float* delta, const int32_t* bits, void* out, int out_dtype,
                               int n_bits, int64_t n, void* stream);

/* Per-row min / max / absmax of a weight matrix (StaticQuantizer.init_quant_params statistics,
 *   quant_utils/qdiff/base/base_quantizer.py:70-90).  Any of the outputs may be NULL. */
int wanq_row_minmax(const void* w, int w_dtype, float* row_min, float* row_max, float* row_absmax,
                    int64_t rows, int cols, void* stream);

/* Static per-output-channel weight quantisation with given params (fp32 [rows]):
 *   q = clamp(rne(w/delta) - zp, qmin, qmax)  -> fake-quant (q+zp)*delta -> fp32, and/or int8 codes (q saturated to
 *   [-128,127] on top of [qmin,qmax]: the reference's own clamp is looser than the bit-width).
 *   (base_quantizer.py:56-68; export: wan/quant_wanx_cuda.py:39-53).  q8 / deq may be NULL. */
int wanq_weight_quant(const void* w, int w_dtype, const float* delta, const float* zp, int qmin, int qmax,
                      int8_t* q8, float* deq, int64_t rows, int cols, void* stream);

/* Reference-format int8 export of a weight matrix, all arithmetic in HALF precision as the reference does it:
 *   q8 = clamp( round( f16(w) / delta ) - zp, -128, 127 )   delta, zp: fp16 [rows]
 * Replaces quantize_and_save_weight_ (ViDiT-Q/examples/Wan2.1/wan/quant_wanx_cuda.py:39-53); bit-identical to it
 * (tests/golden a12_*).  The codes feed `int_weight.pt` written with reference_format=True. */
int wanq_weight_export_f16(const void* w, int w_dtype, const void* delta_f16, const void* zp_f16, int8_t* q8,
                           int64_t rows, int cols, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Attention front-end for q / k:  y = x * rsqrt(mean(x^2) + eps) * weight   (RMSNorm over ALL cols),
 * then the 3-axis rotary embedding per head: pairs (2i, 2i+1) of every head are multiplied by
 * (cos + i sin) from rope[pos, i] with pos = row % rows_per_batch; rows with pos >= positions (sequence
 * padding) are left unrotated.  weight fp32[cols] or NULL (rope only); rope fp32[positions, head_dim/2, 2]
 * or NULL (norm only).  In place (out == x) is allowed.
 * Replaces WanRMSNorm.forward + rope_apply (ViDiT-Q/examples/Wan2.1/wan/modules/model.py:43-89), which the
 * reference evaluates with several torch passes and a float64 complex multiply. */
int wanq_rmsnorm_rope(const void* x, int x_dtype, const float* weight, const float* rope, void* out,
                      int out_dtype, int64_t rows, int cols, int head_dim, int64_t rows_per_batch,
                      int64_t positions, float eps, void* stream);

/* The same, additionally (or only: out may be NULL) emitting the row as per-(token, head) symmetric int8 codes for the int8
 * Q.K^T attention below: for every head slice of 128 columns  delta = max(absmax / 127, 1e-6),  q8 = clamp(rne(y / delta))
 * -- DynamicQuantizer on [tokens*heads, head_dim] rows, the q / k recipe of the reference's quantized attention
 * (ViDiT-Q/quant_utils/qdiff/base/quant_attn.py:168-174, examples/Wan2.1/models/quant_opensora.py:431-436).
 * q8: int8 [rows, cols].  qscale: fp32, two planes of [cols/128][scale_stride]: delta[h][row], then -12582912 * delta[h][row]
 * (the constant of the attention kernel's dequantising fma); scale_stride >= rows (for keys: rows rounded up to 64).
 * head_dim must be 128. */
int wanq_rmsnorm_rope_q8(const void* x, int x_dtype, const float* weight, const float* rope, void* out,
                         int out_dtype, int8_t* q8, float* qscale, int64_t scale_stride, int64_t rows, int cols,
                         int head_dim, int64_t rows_per_batch, int64_t positions, float eps, void* stream);

/* The same as wanq_rmsnorm_rope with the store scattered per head: head h (head_dim columns) of row r lands at
 * out + head_map[2h] + r * head_map[2h+1] (elements of out_dtype).  head_map: DEVICE int64 [cols/head_dim][2].
 * Writes the Ulysses head-scatter all-to-all send buffers ([P][rows][w] per head chunk) directly, in place of the
 * permute + contiguous pack the reference's all_to_all_4D does around the collective
 * (ViDiT-Q/examples/Wan2.1/wan/distributed/xdit_context_parallel.py:147-192, yunchang SeqAllToAll4D). */
int wanq_rmsnorm_rope_scatter(const void* x, int x_dtype, const float* weight, const float* rope, void* out,
                              int out_dtype, const int64_t* head_map, int64_t rows, int cols, int head_dim,
                              int64_t rows_per_batch, int64_t positions, float eps, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Flash-attention forward, non-causal:  o[q,h,:] = softmax_k(q[q,h,:].k[k,h,:] * scale) v[k,h,:] over the
 * first Lk keys.  Token-major tensors [tokens, heads*head_dim] with a token stride in ELEMENTS (so q/k/v may
 * be column slices of one packed buffer).  dtype: WANQ_BF16; head_dim: 128.  fp32 online softmax, bf16 MFMA.
 * Replaces flash_attention(q, k, v, k_lens) (ViDiT-Q/examples/Wan2.1/wan/modules/attention.py:24-130, which
 * calls the external flash_attn library) for batch size 1. */
int wanq_attention_fwd(const void* q, const void* k, const void* v, void* o, int dtype, int64_t Lq,
                       int64_t Lk, int heads, int head_dim, int64_t q_stride, int64_t k_stride,
                       int64_t v_stride, int64_t o_stride, float scale, void* stream);

/* The same with the keys split over `splits` workgroups per (query block, head) ("split-KV"): every share writes
 * unnormalised fp32 partials (O, reference maximum, row sum) into `workspace` and a second kernel merges them.  For launches
 * whose (query blocks x heads) grid leaves a large part of the last round of CUs idle -- 3 heads x 128 query blocks on 256
 * CUs under 4-way sequence parallelism: 1.5 rounds cost 2 -- halving the work unit restores the balance.
 * wanq_attention_split_workspace() gives the bytes `workspace` (16-byte aligned, device memory) must hold; splits <= 1 or
 * more splits than 64-key tiles fall back to fewer.  No counterpart in the reference (flash_attn picks its own splits). */
int64_t wanq_attention_split_workspace(int64_t Lq, int heads, int head_dim, int splits);

int wanq_attention_fwd_split(const void* q, const void* k, const void* v, void* o, int dtype, int64_t Lq,
                             int64_t Lk, int heads, int head_dim, int64_t q_stride, int64_t k_stride,
                             int64_t v_stride, int64_t o_stride, float scale, int splits, void* workspace,
                             int64_t workspace_bytes, void* stream);

/* Diagnostic: which form of the bf16 attention kernel wanq_attention_fwd launches.  The kernel exists with 8 waves per workgroup
 * (256 queries, three ring stages, one workgroup per CU) and with 4 (128 queries, two stages, two workgroups per CU); key sequences
 * up to `nw4_keys` run the 4-wave form (start-up default 1024 or WANQ_ATTN_NW4_KEYS: cross-attention), longer ones the 8-wave form.
 * 0 = always 8 waves, a large value = always 4, -1 = back to the start-up value.  Process-wide; returns the previous setting.
 * Outputs are bit-identical across the two forms (tests/test_gpu_block.py).  No reference counterpart. */
int64_t wanq_attention_select_form(int64_t nw4_keys);

/* Quantized Q.K^T (the reference's `attn.qk` fake-quant recipe, quant_attn.py:168-174, run on the integer matrix cores):
 *   o[q,h,:] = softmax_k( (q8[q,h,:] . k8[k,h,:]) * delta_q[h][q] * delta_k[h][k] * scale ) v[k,h,:]
 * q8 / k8: int8 [tokens, heads*128] with byte strides q8_stride / k8_stride; q_scale: fp32 [heads][qs_stride]; k_scale: fp32
 * two planes [heads][ks_stride] (delta_k, then -12582912*delta_k: the layout wanq_rmsnorm_rope_q8 writes), ks_stride >= Lk
 * rounded up to 64.  S = K8.Q8^T on v_mfma_i32_32x32x32_i8 (integer-exact), P.V in bf16 as above; v / o bf16.
 * splits as in wanq_attention_fwd_split (1 = none; workspace from wanq_attention_split_workspace).
 * The reference wires this recipe for OpenSORA only (Q/base/quant_attn.py is imported, not used, by its Wan model). */
int wanq_attention_qk8_fwd(const int8_t* q8, const float* q_scale, int64_t qs_stride, const int8_t* k8,
                           const float* k_scale, int64_t ks_stride, const void* v, void* o, int dtype, int64_t Lq,
                           int64_t Lk, int heads, int head_dim, int64_t q8_stride, int64_t k8_stride,
                           int64_t v_stride, int64_t o_stride, float scale, int splits, void* workspace,
                           int64_t workspace_bytes, void* stream);

/* Attention with a QUANTISED ATTENTION MAP (the reference's `attn.attn_map` / `cross_attn.attn_map`, group 'row':
 * QuantizedAttentionMapOpenSORA, ViDiT-Q/quant_utils/qdiff/base/quant_attn.py:118-173, applied between softmax and `attn @ v`
 * at examples/Wan2.1/models/quant_opensora.py:459-476): every KEY column of the post-softmax map is one dynamic quantisation
 * group shared by all queries (DynamicQuantizer, base_quantizer.py:101-162), which on a map in [0, 1] is
 *   P~[q,k] = rne(P[q,k] / delta_k) * delta_k,  delta_k = max_q P[q,k] / L,  L = 2^(n_bits-1) - 1 (sym) or 2^n_bits - 1 (asym).
 * The reference materialises the N x N map (and asserts against flash attention); this entry point streams it in three passes
 * over the keys (row statistics, column maxima, quantised P.V), so it also runs at lengths where the map does not fit.  The map
 * is kept in fp32 (the reference rounds it to the model's 16-bit dtype first when it runs under autocast).
 * workspace: wanq_attention_map_workspace() bytes of device memory, 16-byte aligned.  q / k / v / o as in wanq_attention_fwd. */
int64_t wanq_attention_map_workspace(int64_t Lq, int64_t Lk, int heads);
int wanq_attention_map_quant_fwd(const void* q, const void* k, const void* v, void* o, int dtype, int64_t Lq, int64_t Lk,
                                 int heads, int head_dim, int64_t q_stride, int64_t k_stride, int64_t v_stride,
                                 int64_t o_stride, float scale, int n_bits, int sym, void* workspace,
                                 int64_t workspace_bytes, void* stream);
/* The reference's FULL quantised-attention recipe in one call (models/quant_opensora.py:431-476: the q / k / v quantisers AND the
 * attention-map quantiser): q and k as the per-(token, head) int8 codes and fp32 scale planes wanq_rmsnorm_rope_q8 writes
 * (q_scale / k_scale: plane [heads][stride] of delta; the key plane 16-byte aligned, its stride a multiple of 4 covering whole
 * 64-key tiles), S = K8 . Q8^T on the int8 matrix cores in all three passes; v is the caller's, already fake-quantised
 * (wanq_fake_quant_cols).  Same workspace, map quantiser and output as wanq_attention_map_quant_fwd. */
int wanq_attention_map_quant_qk8_fwd(const int8_t* q8, const float* q_scale, int64_t qs_stride, const int8_t* k8,
                                     const float* k_scale, int64_t ks_stride, const void* v, void* o, int dtype, int64_t Lq,
                                     int64_t Lk, int heads, int head_dim, int64_t q8_stride, int64_t k8_stride,
                                     int64_t v_stride, int64_t o_stride, float scale, int n_bits, int sym, void* workspace,
                                     int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * ViDiT activation transform fused with the per-token quantiser:
 *   y = hadU(x * premul),   hadU = (H_K (x) H_128) / sqrt(cols)   (natural-order Walsh-Hadamard on each
 *   128-wide block, then the reference's +-1 table of order K across blocks)
 * premul fp32[cols] = channel_mask * rotation_signs (either may be all ones), or NULL.  had_k = 0: no rotation, y = x * premul
 * (SmoothQuant, any cols % 8 == 0); otherwise had_k must equal cols / 128
 * (the caller states the factorisation it expects); the table across blocks is the one get_hadK picks for this width
 * (ViDiT-Q/quant_utils/qdiff/quarot/quarot_utils.py:100-155), a fixed function of cols: Sylvester for 2^p (had_k <= 32),
 * Paley-I of order 12 for 1536, H_2 (x) Paley-I of order 20 for 5120 -- generated inside the library, not passed in.
 * Equals `x*channel_mask -> (x.double() @ rotation_matrix)` of ViDiTQuantizedLinear.forward
 * (ViDiT-Q/quant_utils/qdiff/viditq/viditq_quant_layer.py:62-63) because row i of the random Hadamard
 * matrix is sign_i * hadU(e_i) (quarot_utils.py:186-192); evaluated in fp32 registers with O(n log n + nK) adds instead
 * of a dense fp64 GEMM, no LDS.  Outputs: fp (out_fp) and/or int8 codes + scale (+sum), as wanq_quant_rows.  (No activation
 * option: tanh-GELU in front of a transformed layer is fused into the producing GEMM, WANQ_EPI_GELU.) */
int wanq_rotate_quant_rows(const void* x, int x_dtype, const float* premul, int had_k, void* out_fp, int out_dtype,
                           int8_t* q, void* scale, void* sum, int vec_dtype, int64_t rows, int cols, void* stream);

/* LayerNorm -> modulate (as wanq_layernorm_rows) -> ViDiT transform -> int8 quantise, one kernel. */
int wanq_layernorm_rotate_quant_rows(const void* x, int x_dtype, const void* gamma, const void* mshift,
                                     const void* mscale, int mod_dtype, int64_t mod_stride, int64_t rows_per_batch,
                                     float eps, const float* premul, int had_k, int8_t* q, void* scale, void* sum,
                                     int vec_dtype, int64_t rows, int cols, void* stream);

/* The same, for up to three consumers of ONE normalised row: self-attention q / k / v share norm1 and the modulation
 * (ViDiT-Q/examples/Wan2.1/wan/modules/model.py:327-331) but each ViDiTQuantizedLinear has its own channel mask and
 * rotation signs (viditq_quant_layer.py:30-38), so the reference normalises, scales, rotates (fp64 GEMM) and quantises
 * three times.  Here x is read and normalised once; set t gets hadU(LN(x) * premul[t]) quantised into q[t] / scale[t] /
 * sum[t].  premul / q / scale / sum are host arrays of nsets (1..3) device pointers. */
int wanq_layernorm_rotate_quant_rows_multi(const void* x, int x_dtype, const void* gamma, const void* mshift,
                                           const void* mscale, int mod_dtype, int64_t mod_stride, int64_t rows_per_batch,
                                           float eps, int nsets, const float* const* premul, int had_k,
                                           int8_t* const* q, void* const* scale, void* const* sum, int vec_dtype,
                                           int64_t rows, int cols, void* stream);

/* ------------------------------------------------------------------------------------------------
 * 4-bit weights.  Storage: row-major [N, K/2] bytes, K % 32 == 0; each group of 32 consecutive codes takes 16 bytes = 4
 * dwords (P0a, P1a, P0b, P1b); for a 16-code half e[0..15] (a = codes 0-15, b = codes 16-31 of the group), as unsigned
 * nibbles u = e + bias:   P0 byte i = u[i] | u[4+i] << 4,   P1 byte i = u[8+i] | u[12+i] << 4   (i = 0..3),
 * so `P & 0x0f0f0f0f` and `(P >> 4) & 0x0f0f0f0f` are the dwords of an int8 MFMA operand.  bias 8 for signed codes in
 * [-8,7] (qdiff 4-bit asym, base_quantizer.py:32,89-90), bias 0 for unsigned codes 0..15 (QServe convention,
 * ViDiT-Q/kernels/csrc/qgemm/w4a8/w4a8_per_channel_gemm_cuda_qserve.cu:287-299).  The reference exports a W4A8 GEMM
 * (w4a8_of16_nobias_weight_asym_qserve) but ships neither a packer nor a module that calls it, and its layout is an NVIDIA
 * ldmatrix interleave; this layout is the CDNA4 counterpart. */
int wanq_pack_w4(const int8_t* q, uint8_t* packed, int bias, int64_t rows, int cols, void* stream);
int wanq_unpack_w4(const uint8_t* packed, int8_t* q, int bias, int64_t rows, int cols, void* stream);

/* W4A8 GEMM: as wanq_gemm_w8a8 with `w_packed` = UNSIGNED nibbles u in [0,15] in the layout above (N x K/2 bytes), expanded to
 * int8 in registers next to the MFMA (the weight panel's HBM / LDS bytes halve):
 *   acc[m,n] = sum_k a[m,k] * u[n,k];   y = acc*sa[m]*sw[n] (+ asum[m]*zp[n]*sw[n]) (+ bias[n]) ...   (same epilogue flags)
 * With zp = -zero this is y = acc*sW*sA - (sW*zW)*sumA (w4a8_per_channel_gemm_cuda_qserve.cu:580-587); signed qdiff codes q
 * stored with bias 8 use zp = zero_point - 8.  K % 32 == 0.
 * Replaces qgemm.w4a8_of16_nobias_weight_asym_qserve (ViDiT-Q/kernels/csrc/qgemm/w4a8/w4a8_per_channel_gemm_cuda_qserve.cu:304-656). */
int wanq_gemm_w4a8(const int8_t* a, const uint8_t* w_packed, void* out, int out_dtype, const void* sa,
                   const void* asum, int tok_dtype, const void* sw, const void* bias, int ch_dtype,
                   const void* zp, int zp_dtype, const float* gate, const void* residual, int epi_flags,
                   int64_t M, int N, int K, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Step-level elementwise work of the denoising loop in one kernel:  out[o] = sum_i coef[o][i] * in[i]  over fp32 tensors of
 * `numel` elements (n_out <= 4, n_in <= 8; `in` / `out` are HOST arrays of device pointers, 16-byte aligned, outputs may
 * alias inputs element for element).  Classifier-free guidance, the model-output conversion and the UniPC / DPM++ / Euler
 * predictor and corrector are all such combinations (ViDiT-Q/examples/Wan2.1/wan/text2video.py:260-269,
 * wan/utils/fm_solvers_unipc.py:303-307,354-630): one launch replaces the ~12 elementwise torch kernels of a step.
 * coef: HOST fp32 [n_out][n_in], copied into the kernel arguments by this call (the launch does not read it afterwards). */
int wanq_lincomb(int n_out, int n_in, const float* coef, const float* const* in, float* const* out, int64_t numel,
                 void* stream);

/* ------------------------------------------------------------------------------------------------
 * The fp32 ends of a DiT pass (ViDiT-Q/examples/Wan2.1/wan/modules/model.py:580-610 embeddings, :372-400 Head, :633-656
 * unpatchify), one fp32 matrix-core tile kernel with the gather / LayerNorm / scatter folded into its loads and stores.
 * Activations: 0 none, 1 GELU (tanh form), 2 SiLU.  All tensors fp32, row-major, 16-byte aligned; K a multiple of 16.
 *
 * wanq_linear_f32:  out[M, N] = act_out(act_in(x)[M, K] . w[N, K]^T + bias)      (bias may be NULL)
 *   x holds x_rows <= M rows; the missing ones are zeros (text_embedding pads its input to text_len, model.py:600-605).
 *   replaces nn.Linear / nn.Sequential(Linear, act, Linear) of time_embedding, time_projection, text_embedding.
 * wanq_time_sinusoid: sinusoidal_embedding_1d (model.py:18-28) of n positions (t_kind: 0 fp32, 1 int64, 2 fp64, 3 int32), float64
 *   angles, cos half first; out fp32 [n, dim]. */
int wanq_linear_f32(const float* x, int64_t x_rows, const float* w, const float* bias, float* out, int64_t M, int N, int K,
                    int in_act, int out_act, void* stream);
int wanq_time_sinusoid(const void* t, int t_kind, float* out, int n, int dim, void* stream);
/* wanq_patch_embed: Conv3d(C -> N, kernel = stride = (pt, ph, pw)) on latent [C, F, H, W] (model.py:580-584), w = the
 *   convolution weight [N, C, pt, ph, pw]; out [out_rows, N], token (f, h, w) in row (f * H/ph + h) * W/pw + w, rows from the token
 *   count up to out_rows zero (the reference pads the sequence to seq_len, model.py:586-590). */
int wanq_patch_embed(const float* latent, const float* w, const float* bias, float* out, int C, int F, int H, int W, int pt,
                     int ph, int pw, int N, int64_t out_rows, void* stream);
/* wanq_head_fwd: Head.forward (model.py:391-399): LayerNorm(x[rows, K], eps, no affine) * (1 + modulation[1] + e) + modulation[0] + e,
 *   then Linear(K -> N).  modulation [2, K], e [K].  unpatchify == 0: out [rows, N].  unpatchify != 0: rows must be the token count
 *   of the latent [C, F, H, W] under patch (pt, ph, pw), N == C * pt * ph * pw <= 64, and out is that latent (model.py:633-656). */
int wanq_head_fwd(const float* x, const float* modulation, const float* e, const float* w, const float* bias, float* out,
                  int64_t rows, int K, int N, float eps, int unpatchify, int C, int F, int H, int W, int pt, int ph, int pw,
                  void* stream);

#ifdef __cplusplus
}
#endif
#endif /* WANQ_HIP_H */
