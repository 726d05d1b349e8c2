/*
 * indextts_hip.h -- C ABI of libindextts_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the IndexTTS
 * inference hot path (GPT-2 acoustic-token decoder with KV cache + conditioned BigVGAN vocoder).
 *
 * Conventions
 *   - every entry point returns ITTS_OK (0) or an ITTS_ERR_* code; the message is available from
 *     itts_last_error() (thread-local).  No exceptions cross the ABI, no hidden device allocations: all
 *     buffers (outputs, workspaces) are passed in by the caller.  Launches are asynchronous on `stream`
 *     (a hipStream_t passed as void*; NULL = default stream) and are hipGraph-capturable.
 *   - `dtype` selects the storage type T of weights / T-typed activations: ITTS_F32, ITTS_BF16, ITTS_F16.
 *     All accumulation is fp32.  The GPT residual stream, logits and LayerNorm statistics are always fp32.
 *   - activations of the vocoder are channels-last: [B][T][C] (time-major rows, channels contiguous).
 *
 * Reference interfaces replaced (paths relative to the reference repo, CreateIntelligens/index-tts-lora):
 *   itts_aa_snake_fwd      <- the reference's only native op: anti_alias_activation_cuda.forward(input, up_filter,
 *                             down_filter, alpha, beta), indextts/BigVGAN/alias_free_activation/cuda/
 *                             anti_alias_activation.cpp:19-22 / anti_alias_activation_cuda.cu:44-181,214-256, and its
 *                             torch twin alias_free_torch/act.py:10-28 (the numerical oracle).
 *   itts_gemm_skinny,
 *   itts_ln_reduce,
 *   itts_attn_decode,
 *   itts_embed_step,
 *   itts_sample            <- one cached decode step of GPT2InferenceModel.forward (indextts/gpt/model.py:163-193)
 *                             + the HF generate() logits processors / token selection reached from model.py:710-715.
 *   itts_gemm_conv,
 *   itts_attn_prefill,
 *   itts_layernorm         <- prefill branch (model.py:151-162) and the teacher-forced latent pass
 *                             UnifiedVoice.forward(return_latent=True) (model.py:548-597, 459-474).
 *   itts_gemm_conv,
 *   itts_aa_snake_fwd,
 *   itts_tanh_pcm          <- BigVGAN.forward / AMPBlock1.forward (indextts/BigVGAN/models.py:203-252, 65-74) and the
 *                             output stage of IndexTTS.infer (indextts/infer.py:892-893).
 */
#ifndef INDEXTTS_HIP_H
#define INDEXTTS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ITTS_OK 0
#define ITTS_ERR_INVALID 1 /* bad argument / unsupported shape (nothing was launched) */
#define ITTS_ERR_LAUNCH 2  /* HIP reported an error at launch */

#define ITTS_F32 0
#define ITTS_BF16 1
#define ITTS_F16 2

#define ITTS_ABI_VERSION 8 /* 8: prompt front-end (itts_subsample_conv, itts_mha_small, itts_glu_dwconv_ln_silu, itts_rows, itts_geglu, itts_prefix_rows) and speaker encoder (itts_im2col_reflect, itts_res2_step, itts_se_gate, itts_scale_resid, itts_col_stats, ITTS_EPI_RELU_AFFINE_*), itts_kv_share_rows, ITTS_EPI_SILU_STORE, y_row0 / y_mtp and any M with rows_per_wg in itts_gemm_skinny, itts_gemm_conv ksplit <= 64; 7: LayerNorm folded into the consuming skinny GEMM (ln_c), residual epilogue with a packed T copy, rows_per_wg / wide_wg, bump words in itts_gemm_skinny / itts_embed_step (+ clamp, packed copy); the reducer tail is gone; paged KV cache (kv_tab / kv_bs); 6: per-row clocks (row_step0) in itts_embed_step / itts_sample_args (slot refill), itts_attn_prefill_prefix / _shared, kv_share in itts_attn_decode; 5: itts_ln_reduce takes up to 6 slabs */

int itts_abi_version(void);
const char* itts_last_error(void);
/* The product library keeps no process-wide mutable state besides this thread-local error string and immutable tables:
 * tuning overrides and in-kernel time stamps exist only in the diagnostic build (include/indextts_hip_diag.h). */

/* ------------------------------------------------------------------------------------------------------------------
 * Packed weight layout (shared by itts_gemm_skinny and itts_gemm_conv).  A logical matrix W[K][N] (K = reduction
 * dim, N = output columns) is stored as 1-KiB blocks in MFMA B-fragment order:
 *     block(nt, ks) , nt = n/16 in [0, ceil(N/16)), ks = k/KS in [0, ceil(K/KS)),  KS = 32 (bf16/f16) or 16 (f32)
 *     inside a block: lane l = (g<<4)|c (g = 0..3, c = 0..15) owns 16 contiguous bytes = E elements
 *                     W[ks*KS + g*E + e][nt*16 + c],  e = 0..E-1,  E = 8 (bf16/f16) or 4 (f32)
 *     block order: ((tap * NT + nt) * KT + ks), zero-padded in K and N.
 * itts_pack_weight performs this packing on the device.
 * ------------------------------------------------------------------------------------------------------------------ */
int itts_pack_weight(const void* w /* [taps][K][N] T, row-major */, void* packed, int taps, int K, int N, int dtype,
                     void* stream);
/* bytes needed for `packed` */
int64_t itts_packed_bytes(int taps, int K, int N, int dtype);

/* ------------------------------------------------------------------------------------------------------------------
 * Anti-aliased periodic activation (Activation1d(SnakeBeta)):
 *   replicate-pad 5 | x2 zero-stuffed 12-tap FIR (gain 2) | x + sin^2(x e^alpha)/(e^beta + 1e-9) | replicate-pad 5/6
 *   | 12-tap stride-2 FIR.   alpha/beta are the log-scale per-channel parameters; fp32 accumulation.
 * layout 0: x,y are [B][T][C] (channels-last, product path);  layout 1: [B][C][T] (the reference op's layout).
 * valid_rows (layout 0 only, NULL = all T rows): int32 [B] on the device, the length of each batch element's sequence --
 * the filters' replicate padding clamps there, as if the element were processed alone; rows past it are not written.
 * ------------------------------------------------------------------------------------------------------------------ */
int itts_aa_snake_fwd(const void* x, void* y, const float* alpha_log, const float* beta_log, const float* up_filter12,
                      const float* down_filter12, int B, int T, int C, int dtype, int layout, const int32_t* valid_rows,
                      void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Packed activation layout of the decode step (optional, per operand).  A T-typed activation matrix X[M][K] that a skinny
 * GEMM consumes can be stored in MFMA-fragment order, 1 KiB per (k-step, 16-row tile):
 *     element (m, k) at  ((k / KS * MTP + m / 16) * 64 + (k % KS) / E * 16 + m % 16) * E + k % E,   MTP = ceil(M / 16),
 *     KS = 32, E = 8 (bf16/f16) or KS = 16, E = 4 (f32); the buffer holds K/KS * MTP KiB (rows up to 16*MTP exist).
 * The GEMM then reads a fragment as one contiguous 1-KiB wave-load instead of 16 rows x 64 bytes.  Producers that can
 * write it: itts_ln_reduce (y_packed), itts_attn_decode (out_packed), itts_gemm_skinny (y_packed), itts_embed_step (h_packed).
 * ------------------------------------------------------------------------------------------------------------------ */

/* ------------------------------------------------------------------------------------------------------------------
 * Paged KV cache (optional, per call: kv_tab != NULL).  The contiguous cache of one layer is T [rows][H][smax][64].  The paged
 * form is a pool T [blocks][H][kv_bs][64], kv_bs in {16, 32, 64} positions per block, plus a block table int32
 * [rows][ITTS_KV_TAB] on the device: position j of cache row r lives in block kv_tab[r][(j / kv_bs) % ITTS_KV_TAB], offset
 * j % kv_bs.  The table is a RING over the position index: the decode loop's shared write position only grows, a row's live
 * window [pad_r, pos] must stay under (ITTS_KV_TAB - 2) * kv_bs positions, and the caller (GPTEngine) deals blocks to rows as
 * the loop advances and takes them back when a row has stopped -- a refilled decode slot reclaims the blocks of the row that
 * left.  Every table entry must always name a block of the pool (rows past their window point at a scratch block): the
 * kernels read clamped / masked positions without branching.  `smax` is ignored in the paged form.
 * Entry points that take (kv_tab, kv_bs): itts_gemm_skinny (QKV epilogue: append), itts_attn_decode, itts_attn_prefill_packed /
 * _prefix / _shared (prompt rows in, cached prefix out).  Beam search keeps the contiguous form (its per-position row table).
 * ------------------------------------------------------------------------------------------------------------------ */
#define ITTS_KV_TAB 64

/* ------------------------------------------------------------------------------------------------------------------
 * Skinny GEMM for the decode step: Y[M][N] = epi( X[M][K] @ W[K][N] + bias ), up to 96 rows (bf16/f16; 16 in fp32) per
 * pass over the weights (larger M is processed in row chunks by the entry point, or -- rows_per_wg > 0 -- as ONE launch
 * whose row tiles are dealt to grid.z: the prompt front-end's ~150-row GEMMs).  1-3 16-column tiles (x one K slice)
 * per workgroup, the K range split over the workgroup's waves and reduced deterministically through LDS; every global
 * load is issued before its first use (one memory round trip per launch); 8/16-byte epilogue accesses.
 * ------------------------------------------------------------------------------------------------------------------ */
#define ITTS_EPI_STORE 0      /* y (T [M][N]) = v */
#define ITTS_EPI_GELU_STORE 1 /* y (T [M][N]) = gelu_new(v) */
#define ITTS_EPI_RESID_F32 2  /* yf (fp32 [M][N]) += v   (residual stream update; ksplit 1: one owner per element); if y != NULL
                                 the new values are also stored as T in y ([M][N], or packed with y_packed) */
#define ITTS_EPI_QKV_CACHE 3  /* cols [0,D): y (T [M][D]) = v ; [D,2D): K cache ; [2D,3D): V cache, at position *pos */
#define ITTS_EPI_STORE_F32 4  /* yf (fp32 [M][N]) = v               (logits) */
#define ITTS_EPI_SLAB_F32 5   /* yf (fp32 [ksplit][M][N]): slice ks stores its partial product (bias added by slice 0);
                                 the slabs are summed in order by itts_ln_reduce */
#define ITTS_EPI_SILU_STORE 6 /* y (T [M][N]) = v * sigmoid(v)   (the Conformer's feed-forward activation) */
#define ITTS_EPI_RELU_AFFINE_STORE 7      /* y (T) = relu(v) * post_scale[n] + post_shift[n]: a TDNN block of the speaker encoder,
                                             eval-mode BatchNorm behind the ReLU (indextts/BigVGAN/ECAPA_TDNN.py:79-130) */
#define ITTS_EPI_RELU_AFFINE_TANH_STORE 8 /* y (T) = tanh(relu(v) * post_scale[n] + post_shift[n])  (the attention branch of its pooling) */

typedef struct itts_skinny_args {
  int dtype;
  int M, N, K;
  const void* wp;    /* packed W */
  const float* bias; /* [N] or NULL */
  const void* x;     /* T [M][K] */
  int epi;
  void* y;
  float* yf;
  void* kcache; /* T [M][heads][smax][64] for this layer */
  void* vcache;
  const int32_t* pos; /* device scalar: cache row to write */
  int heads, smax;
  int ksplit; /* split-K over workgroups (grid.y); > 1 only with ITTS_EPI_SLAB_F32 */
  /* LayerNorm folded into this GEMM (ln_c != NULL; bf16 / f16, ksplit 1, N % 4 == 0, storing epilogues):
   *     v = rstd[m] * ( (x @ wp)[m][n] - mean[m] * ln_c[n] ) + bias[n]
   * with mean / rstd the LayerNorm statistics (eps = ln_eps, 0 = 1e-5) of row m of x itself, computed by the kernel from the
   * operand fragments.  For wp = pack(gamma . W), ln_c[n] = sum_k (gamma . W)[k][n] (of the T-rounded values),
   * bias[n] = sum_k beta[k] W[k][n] + b[n] this equals LN(x; gamma, beta) @ W + b: HF GPT2Block's ln_1 -> c_attn and
   * ln_2 -> c_fc (indextts/gpt/model.py:163-193) without a LayerNorm launch in front of the GEMM. */
  const float* ln_c;
  float ln_eps;
  int32_t* bump; /* device word incremented once by the launch, or NULL (the decode loop's step counter: a launch that does
                    not read it advances it) */
  /* geometry hints (results differ only in the summation order): rows_per_wg 0 = every workgroup covers all rows of a
   * <= 96-row chunk; 16 / 32 = the row tiles are dealt to grid.z (more, lighter workgroups for GEMMs that run without
   * split-K).  wide_wg != 0: 16-wave workgroups where that keeps a long K to one pass (16 rows per workgroup only). */
  int rows_per_wg, wide_wg;
  /* ITTS_EPI_QKV_CACHE into a paged cache (see "Paged KV cache"): block table and block size, or NULL / 0 */
  const int32_t* kv_tab;
  int kv_bs;
  /* Packed-activation layout (see "Packed activation layout" above): x_packed -- x is packed [K/KS][ceil(M/16)][64][E];
   * y_packed -- y (ITTS_EPI_STORE / ITTS_EPI_GELU_STORE / ITTS_EPI_SILU_STORE / ITTS_EPI_RESID_F32, N % KS == 0) is written packed.
   * y_row0 / y_mtp (packed y only; 0 / 0 = the rows are the whole operand): the M rows land at rows [y_row0, y_row0 + M) of a
   * packed operand of y_mtp row tiles (y_row0 % 16 == 0) -- the Perceiver's [latents ; context] operand is written by two GEMMs.
   * x_mtp (packed x only; 0 = ceil(M / 16)): x is the first M rows of a packed operand of x_mtp row tiles. */
  int x_packed, y_packed;
  int y_row0, y_mtp, x_mtp;
  const float* post_scale; /* [N], ITTS_EPI_RELU_AFFINE_* only */
  const float* post_shift;
} itts_skinny_args;
int itts_gemm_skinny(const itts_skinny_args* a, void* stream);
/* launch geometry itts_gemm_skinny would use: out8 = {grid.x, grid.y, waves per workgroup, column tiles per workgroup,
 * k-steps per wave, dynamic LDS bytes, grid.z, row tiles per workgroup} (host-only, launches nothing) */
int itts_skinny_plan(int dtype, int M, int N, int K, int ksplit, int rows_per_wg, int wide_wg, int fold, int* out8);

/* ------------------------------------------------------------------------------------------------------------------
 * Tiled MFMA GEMM / 1-D convolution, channels-last.
 *   acc(b,t,n) = sum_{j<taps} sum_{c<Cin} X[b][t + off0 + j*dil][c] * W[j][c][n]      (rows outside [0,Tin) are zero)
 *   v = scale * (act(acc + bias[n] + bias2[b][n]) + resid(b,t,n))
 *   y(b,t,n) = (accumulate ? y(b,t,n) : 0) + v
 * element (b,t,n) of y / resid lives at base + b*y_bstride + (t*N + n + y_shift), and is touched only when
 * 0 <= t*N + n + y_shift < y_limit.  A plain GEMM is taps=1, off0=0, B=1, Tin=Tout=M.  A transposed convolution with
 * stride u, kernel 2u (or u) is a 2-tap (1-tap) convolution with N = u*Cout and y_shift = -pad*Cout.
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct itts_conv_args {
  int dtype;
  int B, Tin, Tout, Cin, N;
  int taps, off0, dil;
  const void* x;
  int64_t x_bstride; /* elements between batches of x */
  const void* wp;    /* packed [taps][ceil(N/16)][ceil(Cin/KS)] blocks */
  const float* bias; /* [N] or NULL */
  const float* bias2; /* [B][N] or NULL */
  int act;            /* 0 none, 1 gelu_new */
  void* y;
  int y_f32; /* y and resid are fp32 instead of T */
  int64_t y_bstride, y_shift, y_limit;
  const void* resid; /* same mapping as y, or NULL */
  int accumulate;
  float scale;
  /* ragged batches (NULL = every batch element has Tin valid rows): int32 [B] on the device; input rows >= valid_rows[b]
   * of batch element b read as zeros -- the convolution's own zero padding, so each element equals a run on its own --
   * and output tiles that only see that padding are not computed (their rows of y are left untouched). */
  const int32_t* valid_rows;
  /* split-K of a plain GEMM (taps = 1, B = 1, N % 128 == 0; 0 / 1 = off, <= 64): the K range is cut into ksplit slices that run as
   * separate tiles of ONE launch, and y (fp32, y_f32 = 1) receives the slabs [ksplit][Tout][N] -- slice ks holds the rows' partial
   * products over its K range; no bias / bias2 / resid / accumulate / act (itts_ln_reduce sums the slabs in order, adds the bias
   * and the residual and applies the LayerNorm that follows; itts_rows sums any number of them).  For GEMMs with few output tiles (the prefill's N = 1280 projections:
   * 160 tiles of 128 x 128 for 512 workgroup slots). */
  int ksplit;
} itts_conv_args;
int itts_gemm_conv(const itts_conv_args* a, void* stream);


/* LayerNorm over the last dim of fp32 rows; y is T (y_f32 = 0) or fp32 (y_f32 = 1).  If w2 != NULL a second LayerNorm
 * (w2,b2) is applied to the result of the first (ln_f followed by final_norm). */
int itts_layernorm(const float* h, const float* w, const float* b, const float* w2, const float* b2, void* y, int y_f32,
                   int M, int D, int dtype, void* stream);

/* Residual update + LayerNorm for the decode step:
 *   if (nslab > 0)  h[m][:] += bias[:] + slab[0][m][:] + ... + slab[nslab-1][m][:]      (fixed order; h updated in place)
 *                   [+ runtime LoRA, below]
 *   y[m][:] = LN(h[m][:]; w, b)   (then LN(.; w2, b2) if w2 != NULL),  y is T [M][D].
 * nslab <= 6.  slab is fp32 [nslab][M][slab_stride] as written by itts_gemm_skinny(ITTS_EPI_SLAB_F32) with N = slab_stride (0 = D).
 * y_packed != 0 (needs D % 256 == 0): y is written in the packed activation layout.
 * state_bump (int32[2] device words or NULL): both words are incremented once by this launch -- the decode loop's step
 * counter and cache position, advanced here (a launch that reads neither) instead of by the sampling kernel.
 * Runtime LoRA of the producing projection (lora_b != NULL; unmerged adapters, peft convention y = x W + (x A^T) B^T alpha/r):
 * the projection's packed weight carries A^T * alpha/r as lora_r extra output COLUMNS (N = D + 16*ceil(r/16) in the
 * GEMM), so slab columns [D, D + r) hold the split-K partials of x A; this kernel adds (sum of those) @ lora_b, with lora_b
 * = B^T fp32 [r][D].  The LoRA A.B product is thereby fused into the output projection + its residual update: no extra
 * launch, no merged copy of the weight. */
typedef struct itts_ln_reduce_args {
  int dtype, M, D;
  float* h;
  const float* slab;
  int nslab, slab_stride;
  const float* bias;
  const float* w;
  const float* b;
  const float* w2;
  const float* b2;
  void* y;
  int y_packed;
  int32_t* state_bump;
  const float* lora_b;
  int lora_r;
} itts_ln_reduce_args;
int itts_ln_reduce(const itts_ln_reduce_args* a, void* stream);

/* h[b][:] = table[tokens[b]][:] + pos_table[p][:],  p = clamp(*step - row_step0[b] + pos_add, 0, pos_rows - 1)
 * (fp32 tables, fp32 h; pos_rows = rows of pos_table: a finished slot that keeps stepping never reads past the table).
 * bump (device word or NULL) is incremented once by the launch: the decode loop's cache position, which this launch does
 * not read.  h_packed (T, packed activation layout for M = B rows, or NULL): a T-typed copy of the same rows -- the operand
 * of the first LayerNorm-folded GEMM of the step.
 * row_step0 (int32 [B] on the device or NULL = zeros): the loop step at which row b started decoding -- a decode slot that
 * was refilled with a new utterance in the middle of the loop counts its mel positions from its own first token. */
int itts_embed_step(const int32_t* tokens, const float* table, const float* pos_table, const int32_t* step, int pos_add,
                    float* h, int B, int D, int32_t* bump, const int32_t* row_step0, int pos_rows, void* h_packed, int dtype,
                    void* stream);

/* Single-query attention over the KV cache (decode).  q,out: T [B][H*64]; caches T [B][H][smax][64];
 * keys j in [pad[b], *pos] are visible (the key at *pos was just appended).  scale 1/8.  out_packed != 0: out is written in
 * the packed activation layout (M = B, K = H*64).
 * kv_rows / kv_step (both or neither): beam search without cache copies -- position j of logical row b is read from
 * physical cache row kv_rows[(*kv_step & 1)][b][j] (int32 [2][B][smax], maintained by itts_beam_kv_rows).
 * Entries of different rows may name the SAME physical row (the beams of a batch element share their prompt and common
 * history; the engine caches the prompt once per batch element).
 * skip_rows (int32 [B] on the device or NULL): rows with a nonzero entry are left out (out[b] keeps its old contents) --
 * the decode loop passes its `finished` flags, a finished row's logits no longer matter.
 * kv_share (one int32 word on the device, or NULL): (p0 << 8) | C with C <= 255 -- a promise that the first C keys / values of
 * every row, positions [pad[b], pad[b] + C), hold the same bytes as cache row 0's positions [p0, p0 + C) (a one-prompt
 * batch's conditioning latents after itts_attn_prefill_shared); they are then read from row 0 (one L2-resident copy instead
 * of B copies from HBM).  0 = no sharing.  Ignored with kv_rows. */
int itts_attn_decode(const void* q, const void* kcache, const void* vcache, void* out, const int32_t* pad,
                     const int32_t* pos, int B, int H, int smax, int dtype, int out_packed, const int32_t* kv_rows,
                     const int32_t* kv_step, const int32_t* skip_rows, const int32_t* kv_share, const int32_t* kv_tab, int kv_bs,
                     void* stream);

/* Causal self-attention over a whole (left-padded) sequence.  qkv: T [B][S][3*H*64] (q|k|v); out: T [B][S][H*64];
 * query i sees key j iff pad[b] <= j <= i; rows with no visible key produce zeros.  If kcache/vcache are non-NULL the
 * k/v rows are also written to the caches (T [B][H][smax][64]) at rows [0,S). */
int itts_attn_prefill(const void* qkv, void* out, void* kcache, void* vcache, const int32_t* pad, int B, int S, int H,
                      int smax, int dtype, void* stream);

/* The same over PACKED rows: no padding rows exist; the rows of batch element b are [row_off[b], row_off[b+1]) of qkv /
 * out (row_off: int32 [B+1], device), each at most Smax long, plain causal attention inside each element.  With caches,
 * local row i of element b is written to cache row cache_shift[b] + i (cache_shift NULL = 0): with cache_shift = the
 * element's left padding this reproduces the cache layout of the padded form. */
int itts_attn_prefill_packed(const void* qkv, void* out, void* kcache, void* vcache, const int32_t* row_off,
                             const int32_t* cache_shift, int B, int Smax, int H, int smax, int dtype, const int32_t* kv_tab,
                             int kv_bs, void* stream);

/* Packed rows behind a CACHED PREFIX (the teacher-forced latent pass, model.py:459-474 / 548-597, for a batch whose prompt the
 * decode loop has cached): element b's sequence is  pre_len[b] keys / values read from cache row pre_row[b], positions
 * pre_pos0[b] .. pre_pos0[b] + pre_len[b] - 1 (caches T [rows][H][smax][64], read only)  |  its rows [row_off[b], row_off[b+1])
 * of qkv.  Only the qkv rows are queries (query i sits at sequence position pre_len[b] + i and sees every key up to it); out has
 * the rows of qkv.  Key tiles are cut from sequence position 0, so every output row equals, bit for bit, the row
 * itts_attn_prefill_packed produces when the prefix's k / v rows are part of qkv. */
int itts_attn_prefill_prefix(const void* qkv, void* out, const void* kcache, const void* vcache, const int32_t* row_off,
                             const int32_t* pre_len, const int32_t* pre_row, const int32_t* pre_pos0, int B, int Smax, int H,
                             int smax, int dtype, const int32_t* kv_tab, int kv_bs, void* stream);

/* Packed rows behind a prefix that is SHARED and computed in the same pass (the prefill of a batch whose elements all start
 * with the same conditioning latents, model.py:606-667 with one prompt: those rows see only themselves, so their hidden states
 * are the same in every element and are computed once).  Element e of the E packed elements owns rows [row_off[e],
 * row_off[e+1]) of qkv / out; its sequence is  pre_len[e] keys taken from qkv rows pre_row0[e] .. pre_row0[e] + pre_len[e] - 1
 * (k / v thirds of those rows)  |  its own rows.  pre_len[e] = 0 for the shared block itself.  Same key tiling from sequence
 * position 0 as itts_attn_prefill_packed over the un-shared layout: same bits per row.  With caches (T [rows][H][smax][64]),
 * element e's OWN rows are appended to cache row w_row[e] at positions w_pos0[e] + local row (the caller copies the shared
 * block's cache rows to the other elements' cache rows). */
int itts_attn_prefill_shared(const void* qkv, void* out, void* kcache, void* vcache, const int32_t* row_off,
                             const int32_t* pre_len, const int32_t* pre_row0, const int32_t* w_row, const int32_t* w_pos0,
                             int E, int Smax, int H, int smax, int dtype, const int32_t* kv_tab, int kv_bs, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Token selection for one decode step, on device (no host sync in the loop).
 * state (int32[8], device): [0] = step k (number of tokens already generated), [1] = cache position of the NEXT token
 * to be fed, [2] = number of finished rows.  The kernel reads logits (fp32 [B][V]), applies repetition penalty over
 * {extra_ids} U history[b][0..k), temperature, top-k, top-p, draws (Philox4x32-10 keyed by seed, counter (row,k,0,0))
 * or takes the argmax, then appends to history, updates tokens/finished, and increments state[0], state[1].
 * Rows already finished, or with force_stop[b] == k, emit stop_token.
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct itts_sample_args {
  const float* logits;
  int B, V, ldl; /* ldl = row stride of logits */
  int32_t* tokens;   /* [B] out: token selected this step */
  int32_t* history;  /* [B][hist_cap] generated tokens */
  int hist_cap;
  int32_t* finished; /* [B] */
  int32_t* state;    /* [8] */
  const int32_t* extra_ids; /* ids always penalised (the fake prefix: 1 and 8192) */
  int n_extra;
  const int32_t* force_stop; /* [B] or NULL */
  float rep_penalty, temperature, top_p;
  int top_k, do_sample; /* do_sample != 0 requires 1 <= top_k <= 1024 (candidate store; top_k <= 0 is refused, not truncated) */
  uint64_t seed;     /* Philox key = seed + the 64-bit value in state[4..5] (lo, hi): a captured launch serves any seed */
  int stop_token;
  float* dbg_scores; /* optional [B][V] processed scores (-inf = removed), for parity tests; NULL in production */
  int no_advance;    /* != 0: leave state[0] / state[1] alone -- the caller advances them in the next step's first
                        itts_ln_reduce launch (state_bump), which removes a device-wide fence and a returning atomic
                        per row from this kernel; itts_embed_step is then given pos_add + 1 */
  const int32_t* row_step0; /* [B] or NULL (= zeros): loop step at which row b started (slot refill).  Row b's own step
                        k_b = state[0] - row_step0[b] indexes its history, bounds its repetition-penalty window and is what
                        force_stop[b] is compared with; the Philox counter stays (b, state[0]). */
} itts_sample_args;
int itts_sample(const itts_sample_args* a, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Beam search / beam-sample step (num_beams > 1) -- replaces, for the generate() call at indextts/gpt/model.py:710-715,
 * transformers 4.44.2 GenerationMixin._beam_search + BeamSearchScorer.process + BeamHypotheses.add/is_done
 * (third-party; restated in oracle/beam_ref.py).  Rows are laid out batch-major: row = b*num_beams + beam.
 * One call = one decoding step for all batch elements: log-softmax, repetition penalty, (temperature, top-k, top-p with
 * min_tokens_to_keep = 2 when do_sample), running beam scores, 2*num_beams candidates drawn without replacement (own
 * Philox stream: counter (b, step, draw, 0)) or taken, scorer bookkeeping, hypothesis store, per-row token histories.
 * state[0] = step, state[1] = cache position (both advanced by the call), state[2] = finished batch elements.
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct itts_beam_args {
  const float* logits; /* [B*num_beams][ldl] */
  int B, num_beams, V, ldl;
  int32_t* tokens;      /* [B*num_beams] out: token appended to each row */
  int32_t* src;         /* [B*num_beams] out: row (absolute index) each row continues */
  float* beam_scores;   /* [B*num_beams] in/out: running sum of log-probabilities (init 0, -1e9, -1e9, ... per batch element) */
  int32_t* hist;        /* [2][B*num_beams][hist_cap] generated tokens, ping-pong by step parity (input = step & 1) */
  int hist_cap;
  float* hyp_score;     /* [B][num_beams] closed hypotheses: score, generated length, tokens */
  int32_t* hyp_len;
  int32_t* hyp_tok;     /* [B][num_beams][hist_cap] */
  int32_t* n_hyp;       /* [B] */
  float* worst;         /* [B] lowest score among the closed hypotheses (init +1e9) */
  int32_t* done;        /* [B] */
  int32_t* state;       /* [8] */
  const int32_t* extra_ids; /* ids always penalised (the fake prefix: 1 and 8192) */
  int n_extra;
  float rep_penalty, temperature, top_p, length_penalty;
  int top_k, do_sample; /* do_sample != 0 requires 1 <= top_k <= 128 (one 1024-entry candidate pool per batch element) */
  uint64_t seed;        /* Philox key = seed + the 64-bit value in state[4..5], as in itts_sample_args */
  int eos_token;
  /* scratch between the step's two launches (per-row candidates -> per-batch-element pooling); caller-owned, no init:
   * cand_scores / cand_ids [B*num_beams][ITTS_BEAM_CAND] each, cand_n [B*num_beams] */
  float* cand_scores;
  int32_t* cand_ids;
  int32_t* cand_n;
} itts_beam_args;
#define ITTS_BEAM_CAND 1024
int itts_beam_step(const itts_beam_args* a, void* stream);

/* KV rows follow their beams as a TABLE (GPT2InferenceModel._reorder_cache, model.py:207-218, without moving cache bytes):
 * kv_rows int32 [2][rows][smax], ping-pong by step parity; entry [r][j] = physical cache row that holds position j of
 * logical row r (rows = B*num_beams).  Call right after itts_beam_step (state[0] = k+1, state[1] = P): table (k+1)&1 is
 * written from table k&1 -- [r][j] = old[src[r]][j] for j < P, and [r][P] = r (where the next step appends).  Start from
 * table 0 = identity.  itts_attn_decode reads the table through its kv_rows / kv_step arguments. */
int itts_beam_kv_rows(int32_t* kv_rows, const int32_t* src, const int32_t* state, int rows, int smax, void* stream);

/* The same by COPYING cache rows (kept as the reference form; the decode loop uses the table):
 * KV cache rows follow their beams (GPT2InferenceModel._reorder_cache, model.py:207-218): row r <- row src[r] for cache
 * positions [0, state[1]), every layer, K and V; batch elements whose rows map to themselves are skipped.
 * Cache layout [layers][rows][heads][smax][64]; layer_stride in elements. */
int itts_beam_reorder_kv(void* kcache, void* vcache, const int32_t* src, const int32_t* state, int layers, int B, int num_beams,
                         int heads, int smax, int64_t layer_stride, int dtype, void* stream);

/* Keys / values of cache positions [p0, p0 + C) of row 0 -> positions [pad[b], pad[b] + C) of rows 1 .. B-1, all `layers` layers in
 * one launch (layer l of a cache at + l * layer_stride elements): a one-prompt batch's conditioning rows, computed once by the
 * shared-prefix prefill (itts_attn_prefill_shared), reach every row's cache.  Both cache forms (kv_tab NULL = contiguous). */
int itts_kv_share_rows(void* kcache, void* vcache, int layers, int64_t layer_stride, int B, int H, int C, int p0, const int32_t* pad,
                       int smax, const int32_t* kv_tab, int kv_bs, int dtype, void* stream);

/* pcm[b][i] = trunc( clamp(32767 * tanh(x[b][i]), -32767, 32767) ) as int16; also writes fp32 wav if wav != NULL.
 * apply_tanh = 0 skips the tanh (input already in (-1,1)). */
int itts_tanh_pcm(const void* x, float* wav, int16_t* pcm, int64_t n, int dtype, int apply_tanh, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Prompt front-end: the Conformer + Perceiver conditioner of UnifiedVoice.get_conditioning (indextts/gpt/model.py:487-546;
 * conformer_encoder.py:167-290,360-386, conformer/subsampling.py:111-143, conformer/attention.py, perceiver.py:181-312) as a
 * chain of these launches and LayerNorm-folded itts_gemm_skinny calls over one prompt (~150 rows): bf16 / f16 only, fp32
 * residual stream, every GEMM operand in the packed activation layout.  The host side is indextts/gpt/conditioner.py.
 * ------------------------------------------------------------------------------------------------------------------ */
/* y (T [T2][C * F2], T2 = (T-3)/2+1, F2 = (F-3)/2+1) = relu(Conv2d(1, C, 3, stride 2)(mel [T][F])) laid out as the operand of
 * the Linear(C * F2 -> d) that follows: element (t, c * F2 + f).  w [C][9], b [C] fp32. */
int itts_subsample_conv(const float* mel, const float* w, const float* b, void* y, int T, int F, int C, int dtype, void* stream);

/* Multi-head attention over a short sequence, head dim 64:  out[i] = softmax_j( scale * ((q_i + u) . k_j + (q_i + v) . p_j) ) v_j
 * for i < Tq, j < Tk.  q / k / v: T rows with the given strides (elements, multiples of 8), head h at columns [64 h, 64 h + 64).
 * pos (T [H][Tk][64], the layer's projected position table) with bias_u / bias_v (fp32 [H * 64]) = the Conformer's relative-
 * position attention WITHOUT rel_shift; pos = NULL: plain attention (the term and the biases are absent).  out: T in the packed
 * activation layout of a [Tq][H * 64] operand with out_mtp row tiles. */
typedef struct itts_mha_args {
  int dtype;
  int Tq, Tk, H;
  const void* q;
  const void* k;
  const void* v;
  int64_t q_stride, k_stride, v_stride;
  const void* pos;
  const float* bias_u;
  const float* bias_v;
  float scale;
  void* out;
  int out_mtp;
} itts_mha_args;
int itts_mha_small(const itts_mha_args* a, void* stream);

/* The Conformer convolution module between its pointwise convolutions: x (T [T][2 C], value | gate) -> GLU -> depthwise
 * Conv1d(taps, zero "same" padding; w fp32 [C][taps], b [C]) -> LayerNorm(ln_w, ln_b, eps) -> SiLU -> y (T, packed layout of
 * [T][C] with y_mtp row tiles).  C % 128 == 0, C <= 2048; taps 7 / 15 / 31. */
int itts_glu_dwconv_ln_silu(const void* x, const float* w, const float* b, const float* ln_w, const float* ln_b, void* y, int T,
                            int C, int taps, int y_mtp, float eps, int dtype, void* stream);

/* Row operations on an fp32 residual stream [M][D] (D % 4 == 0, D <= 2048):
 *   v = (x ? x[m] : 0) + (bias ? bias : 0) + slab[0][m] + ... + slab[nslab-1][m]       (fixed order; slab fp32 [nslab][M][D])
 *   norm 1: v = LayerNorm(v; w, b, eps)     norm 2: v = v / max(|v|_2, 1e-12) * sqrt(D) * w     norm 0: as is
 *   y (fp32 [M][D], may alias x, or NULL) = v;   y_packed (T, or NULL): rows [y_row0, y_row0 + M) of a packed operand of y_mtp
 *   row tiles (0 = ceil(M / 16)).  */
typedef struct itts_rows_args {
  int dtype, M, D;
  const float* x;
  const float* slab;
  int nslab;
  const float* bias;
  int norm;
  const float* w;
  const float* b;
  float eps;
  float* y;
  void* y_packed;
  int y_row0, y_mtp;
} itts_rows_args;
int itts_rows(const itts_rows_args* a, void* stream);

/* y (T, packed layout of [M][Kp], y_mtp row tiles, 0 = ceil(M / 16)) = gelu(h[:, Kp:]) * h[:, :Kp]  (erf gelu; h T [M][2 Kp]). */
int itts_geglu(const void* h, void* y, int M, int Kp, int y_mtp, int dtype, void* stream);

/* The GPT prompt rows of UnifiedVoice.prepare_gpt_inputs (indextts/gpt/model.py:606-667): per batch row b the ids of text [B][L]
 * (int64) that are neither start_tok nor stop_tok (n of them, order kept) become start | ids | stop, embedded as
 * text_emb[id] + text_pos[j] (fp32 tables of n_tok / n_pos rows, j = index in that sequence), placed behind the C conditioning latents
 * (conds fp32 [conds_rows][C][D], conds_rows 1 = shared by all rows) and right-aligned in P = C + L + 2 positions:
 *   emb (fp32 [B][P][D]): L - n zero rows, the latents, the n + 2 text rows;  mask (int64 [B][P + 1]): 0 on the zero rows, else 1
 *   (the last slot is the start-mel token's);  pad (int32 [B]) = L - n.   L <= 2046, D % 4 == 0. */
int itts_prefix_rows(const int64_t* text, const float* conds, int conds_rows, const float* text_emb, const float* text_pos, float* emb,
                     int64_t* mask, int32_t* pad, int B, int L, int C, int D, int start_tok, int stop_tok, int n_tok, int n_pos,
                     void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Speaker encoder (ECAPA-TDNN, indextts/BigVGAN/ECAPA_TDNN.py:470-581) over one prompt: its 1 x 1 convolutions are
 * itts_gemm_skinny calls with the ITTS_EPI_RELU_AFFINE_* epilogues; the rest is below.  Activations are T-typed packed operands
 * [frames][channels] of `mtp` row tiles; bf16 / f16 only.  The host side is indextts/BigVGAN/speaker_engine.py.
 * ------------------------------------------------------------------------------------------------------------------ */
/* y (packed [T][Kp]) column j * F + f = x[reflect(t + (j - (taps-1)/2) * dil)][f] (x fp32 [T][F]), zeros from taps * F on: a k-tap
 * convolution with reflect "same" padding (nnet/CNN.py:430-488) becomes a plain GEMM over this operand. */
int itts_im2col_reflect(const float* x, void* y, int T, int F, int taps, int dil, int Kp, int y_mtp, int dtype, void* stream);
/* One step of a Res2Net block (scale 8, 64-channel chunks): cat[:, 64 c : 64 c + 64] = BN(relu(conv_k3_dil(y1 chunk c [+ cat chunk
 * c - 1 unless first]) + bias)), reflect padding; first also copies chunk 0 (cat[:, :64] = y1[:, :64]).  wp = itts_pack_weight of the
 * [3 * 64][64] matrix (row = tap * 64 + input channel); bias / scale / shift fp32 [64]. */
int itts_res2_step(const void* y1, void* cat, const void* wp, const float* bias, const float* scale, const float* shift, int T, int mtp,
                   int chunk, int dil, int first, int dtype, void* stream);
/* gate (fp32 [C]) = sigmoid(w2 relu(w1 mean_t(y) + b1) + b2): the squeeze-and-excitation gate; w1 T [H][C], w2 T [C][H] row-major. */
int itts_se_gate(const void* y, const void* w1, const float* b1, const void* w2, const float* b2, float* gate, int T, int C, int H,
                 int mtp, int dtype, void* stream);
/* out = gate[c] * y + res over packed [T][C] operands of one geometry (out may be a k-step run of a wider operand). */
int itts_scale_resid(const void* y, const void* res, const float* gate, void* out, int T, int C, int mtp, int dtype, void* stream);
/* Per-channel statistics over time of packed x [T][C]: weights w_t = softmax_t(logit[t][c]) (logit T row-major [T][C]) or 1 / T
 * (logit NULL); m = sum w x, s = sqrt(max(sum w (x - m)^2, 1e-12)); out (T [2 C]) = [m | s] * scale + shift (fp32 [2 C], or NULL):
 * the global-context statistics and the attentive statistics pooling + BatchNorm (ECAPA_TDNN.py:543-581). */
int itts_col_stats(const void* x, const void* logit, const float* scale, const float* shift, void* out, int T, int C, int mtp, int dtype,
                   void* stream);

#ifdef __cplusplus
}
#endif
#endif /* INDEXTTS_HIP_H */
