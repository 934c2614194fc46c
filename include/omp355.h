/* libomp355 -- MI355X (gfx950) kernels for the OmniParser inference hot path.
 *
 * C ABI.  The reference (AlibabaResearch/AdvancedLiterateMachinery, OCR/OmniParser) has no
 * FFI/plugin layer: its hot path is plain nn.Module composition on ATen ops.  Each entry point
 * below therefore names the reference *function* it replaces (paths relative to
 * OCR/OmniParser/).  INTEGRATION.md shows the ctypes stub a maintainer would drop into the
 * reference for each of them.
 *
 * Rules of the ABI
 *   - every function returns 0 on success or a negative errno-style code; omp_last_error()
 *     returns a thread-local description of the last failure;
 *   - the CALLER owns every buffer (device pointers); the library never allocates device
 *     memory, never synchronises the device and launches only on the given stream
 *     (hipStream_t passed as void*; NULL = the null stream);
 *   - matrices are row-major; activations are token-major [rows, channels];
 *     weights keep the reference's nn.Linear layout [out_features, in_features];
 *   - dtype enum: OMP_F32 / OMP_BF16 for activations+weights; biases, LayerNorm affine
 *     parameters, embedding and bias tables are always fp32; accumulation is fp32.
 *   - development hooks (kernel selectors, trace buffers, measurement brackets, CU-masked streams) are declared in
 *     csrc/omp355_debug.h, not here: they are exported for the tests and tools of this repository only.
 */
#ifndef OMP355_H
#define OMP355_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OMP_ABI_VERSION 21
#define OMP_MAX_DEC_LAYERS 8

enum { OMP_OK = 0, OMP_ERR_LAUNCH = -5, OMP_ERR_INVALID = -22, OMP_ERR_UNSUPPORTED = -95 };
/* OMP_BF16X2 = split-bf16 pair rows: a [rows, C] tensor is stored as rows of 2C bf16, hi = bf16(x) in columns 0..C-1 and
 * lo = bf16(x - hi) in columns C..2C-1 (16 mantissa bits, the same bytes as fp32).  It is the GEMM-operand format of the
 * bf16x3 engine: x.w ~ hi.w_hi + lo.w_hi + hi.w_lo on the bf16 matrix cores (see omp_gemm_args.a_wrap). */
enum { OMP_F32 = 0, OMP_BF16 = 1, OMP_BF16X2 = 2 };
enum { OMP_ACT_NONE = 0, OMP_ACT_GELU = 1, OMP_ACT_RELU = 2 };
enum { OMP_STORE_PLAIN = 0, OMP_STORE_KBLK = 2, OMP_STORE_VBLK = 3, OMP_STORE_ROWSTAT = 4 };
/* decoder kinds (reference: model/transformer.py:26-33 pt/poly/rec decoders) */
enum { OMP_DEC_PT = 0, OMP_DEC_POLY = 1, OMP_DEC_REC = 2 };

typedef void* omp_stream_t; /* hipStream_t */

const char* omp_last_error(void);
int omp_abi_version(void);

/* ---- LayerNorm ---------------------------------------------------------------------------
 * y = (x - mean) / sqrt(var + eps) * gamma + beta over the last dim (biased variance).
 * Replaces nn.LayerNorm call sites: swin_transformer.py:208,250,288,441,616 and
 * transformer.py:326,374,437,441,450.  Writes y (dtype y_dtype, may be NULL; OMP_BF16X2 = split pairs [rows, 2C]) and/or y_f32. */
int omp_layernorm(const void* x, int x_dtype, const float* gamma, const float* beta, void* y,
                  int y_dtype, float* y_f32, int64_t rows, int C, float eps, omp_stream_t s);

/* ---- GEMM with fused epilogue ---------------------------------------------------------------
 * C[m,n] = act(sum_k A[m,k] * W[n,k] + bias[n]) + residual[m,n]
 * Replaces nn.Linear / 1x1 nn.Conv2d call sites: swin_transformer.py:127,148 (qkv, proj),
 * :31-34 (Mlp fc1+GELU, fc2), :294 (PatchMerging.reduction), fpn.py:24-36 (fpn_in 1x1 convs),
 * omniparser.py:31 (input_proj), transformer.py:448 (linear1/2) and the in/out projections of
 * nn.MultiheadAttention (transformer.py:386-387), block/mlp.py:11-13 (prediction heads).
 * bias_row (device int32, optional) selects row *bias_row of a [rows, bias_row_stride] bias table
 * (used for the per-position query/key bias of the decoders).
 * trans_out: store C transposed per batch of `trans_rows` consecutive rows:
 *   C[(m / trans_rows) * N * trans_ld + n * trans_ld + (m % trans_rows)]   (ldc ignored). */
typedef struct {
  const void* A;
  int64_t lda;
  const void* W;
  int64_t ldw;
  const float* bias;
  const int32_t* bias_row;
  int64_t bias_row_stride;
  const void* residual;
  int64_t ldr;
  void* C;
  int64_t ldc;
  int64_t M;
  int32_t N;
  int32_t K;
  int32_t dtype;     /* of A and W */
  int32_t out_dtype; /* of C and residual: dtype, OMP_F32, or OMP_BF16X2 (dtype OMP_BF16, no residual: C is [M, 2N] split pairs) */
  int32_t act;
  int32_t trans_out;
  int64_t trans_rows; /* rows (tokens) per batch item */
  int64_t trans_ld;   /* pitch of a transposed row, >= trans_rows */
  /* optional fused LayerNorm prologue (decoder steps, M <= 64): A is then the fp32 residual stream
   * [M,K] and the GEMM consumes LN(A) * ln_gamma + ln_beta (transformer.py:437,441,450 + the Linear) */
  const float* ln_gamma;
  const float* ln_beta;
  float ln_eps;
  int32_t small_m_splitk; /* hint: M <= 64 weight-streaming GEMM -> split-K workgroups */
  /* destination layout.  OMP_STORE_KBLK / OMP_STORE_VBLK write the decoder cross-attention memory
   * K / V straight into the head-blocked slabs omp_dec_cross_attn_step streams (see DESIGN.md
   * "cross-attention memory layout"):
   *   KBLK: M = kv_images*kv_tokens memory tokens, N = n_slabs*kv_heads*64 features ->
   *         C[slab][image][head][kv_mpad][64]
   *   VBLK: operands swapped by the caller (A = Wv [n_slabs*kv_heads*64, K], W = memory tokens
   *         [kv_images*kv_tokens, K], bias_along_m = 1) ->
   *         C[slab][image][head][kv_mpad/kv_key_block][64][kv_key_block], keys of a block in
   *         matrix-core order (bf16, block 32: slot 8g+4*half+r <- key 16*half+4g+r; f32, block 16: natural)
   *   with out_dtype OMP_BF16X2 (kv_key_block 32): SPLIT-PLANE slabs -- every 32-key block holds [hi plane | lo plane] of bf16,
   *         KBLK C[slab][image][head][kv_mpad/32][2][32][64], VBLK C[slab][image][head][kv_mpad/32][2][64][32] (the bytes of the
   *         fp32 slabs; read by omp_dec_cross_attn_step with dtype OMP_BF16X2) */
  int32_t store_mode;
  int32_t bias_along_m; /* bias[m] instead of bias[n] */
  int32_t kv_images, kv_tokens, kv_mpad, kv_heads, kv_key_block;
  /* optional second destination (plain row-major, pitch ldc2, same dtype as C): act(A W^T + bias) WITHOUT the residual.
   * One GEMM then yields both `memory` and `memory + pos` of omniparser.py:31 / transformer.py:88-96 (the reference
   * adds pos inside nn.MultiheadAttention's key path); requires a residual, OMP_STORE_PLAIN, no trans_out. */
  void* C2;
  int64_t ldc2;
  /* bf16x3 products (fp32-grade results at a third of the bf16 matrix-core rate; the parity engine):
   * A is a split-bf16 pair tensor [M, 2*K0] = [a_hi | a_lo] (OMP_BF16X2 rows, dtype stays OMP_BF16), W is the once-per-
   * checkpoint image [N, 3*K0] = [w_hi | w_hi | w_lo] of an fp32 weight, K = 3*K0 and a_wrap = 2*K0: A columns at and
   * beyond a_wrap are read from column (k - a_wrap), i.e. A is consumed as [a_hi | a_lo | a_hi] without being stored
   * that way.  The contraction is then a_hi.w_hi + a_lo.w_hi + a_hi.w_lo (the a_lo.w_lo term, 2^-16 relative, is
   * dropped).  With swapped operands (OMP_STORE_VBLK) the weight is the [hi | lo] side and the activation arrives as
   * [hi | hi | lo] (omp_split_bf16, triple = 1).  0 = plain product. */
  int32_t a_wrap;
} omp_gemm_args;
int omp_gemm_bias_act(const omp_gemm_args* a, omp_stream_t s);

/* ---- Fused Swin MLP (bf16 engine):  y = x + fc2(GELU(fc1(LayerNorm(x)))) ---------------------------------------
 * Replaces `x = x + self.drop_path(self.mlp(self.norm2(x)))`, swin_transformer.py:250, with Mlp.forward (:30-36)
 * inlined: one launch, the [tokens, hidden] activation never reaches HBM.  x, y: bf16 [M, C] with row pitches
 * ldx / ldy (y may alias x); ln_gamma / ln_beta / b2 (= fc2.bias): fp32 [C]; wpack: fc1.weight, fc1.bias and
 * fc2.weight re-laid out once per checkpoint by model/packing.py::pack_mlp (hidden / 32 sub-chunk images of
 * C*128 + 1024 bytes; layout documented there and in csrc/mlp.hip).  C in {128, 256, 512}, hidden % 32 == 0. */
int omp_swin_mlp_fused(const void* x, int64_t ldx, const float* ln_gamma, const float* ln_beta, float eps,
                       const void* wpack, const float* b2, void* y, int64_t ldy, int64_t M, int C, int hidden,
                       omp_stream_t s);
/* the same with x / y of type x_dtype: OMP_BF16, or OMP_F32 = the fp32 residual stream of the bf16 engine (the products
 * still run on bf16 operands: LayerNorm(x) and the hidden activations are rounded to bf16, x itself never is) */
int omp_swin_mlp_fused2(const void* x, int x_dtype, int64_t ldx, const float* ln_gamma, const float* ln_beta, float eps,
                        const void* wpack, const float* b2, void* y, int64_t ldy, int64_t M, int C, int hidden,
                        omp_stream_t s);

/* ---- Swin-B stage 2 (C = 512, hidden 2048) as row-owner chains (round 5, csrc/dec_rows.hip) ------------------------------------------
 * Everything of a SwinTransformerBlock (swin_transformer.py:196-253) except the window attention core is row-local, so a block is the
 * window attention kernel (omp_swin_window_attn2 on bf16 q | k | v) plus ONE launch of this entry point:
 *   mode 0: qkv = bf16(LayerNorm(x; n1) Wqkv^T + bqkv)                                          (:208 + :127, the stage's first block)
 *   mode 1: x += att Wproj^T + bproj  (:148, :246);  x += fc2(GELU(fc1(LayerNorm(x; n2))))     (:250 with Mlp.forward :30-36 inlined);
 *           and, when n1_g != NULL, the NEXT block's qkv = bf16(LayerNorm(x; n1) Wqkv'^T + bqkv')
 * x: fp32 residual stream [M, 512], in place; att: bf16 [M, 512] (attention output in token order); qkv: bf16 [M, 1536].  A workgroup owns
 * omp_dec_rows_tile() tokens and streams the weights: wstream is written once per checkpoint by model/packing.py (pack_rows_embed_qkv for
 * mode 0, pack_rows_ffn_qkv / pack_rows_ffn for mode 1: proj, 8 chunks of [fc1 rows of 256 hidden units, fc2 columns of the same units],
 * then the next block's qkv; fragment format as omp_dec_rows_ffn).  bf16 matrix-core operands (LayerNorm outputs and hidden activations
 * are rounded to bf16 exactly where the launch-per-Linear path rounds), fp32 accumulation, the bf16 engine's GELU. */
typedef struct {
  int64_t M;
  float eps;
  int32_t mode;
  float* x;
  const void* att;
  void* qkv;
  const void* wstream;
  int64_t wave_stride;
  const float *proj_b, *n2_g, *n2_b, *fc1_b, *fc2_b;   /* mode 1 */
  const float *n1_g, *n1_b, *qkv_b;                    /* the LayerNorm + qkv Linear behind the chain (mode 0: the chain itself) */
  int32_t x3;                                          /* 1 = the parity engine's chain: att split pairs bf16 [M, 1024], qkv fp32 [M, 1536], split weight stream */
} omp_swin_rows_args;
int omp_swin_rows_block(const omp_swin_rows_args* a, omp_stream_t s);

/* ---- Swin patch embedding: zero-pad to x4, 4x4/4 conv (as K=48 dot products), LayerNorm -----
 * Replaces PatchEmbed.forward, swin_transformer.py:427-443.  img is NCHW fp32 (as the reference
 * feeds it); w is the conv weight [E,3,4,4] fp32; out is [B, ceil(H/4)*ceil(W/4), E]. */
int omp_patch_embed_ln(const float* img, const float* w, const float* b, const float* gamma,
                       const float* beta, void* out, int out_dtype, int B, int H, int W, int E,
                       float eps, omp_stream_t s);

/* ---- Fused (shifted-)window attention core -------------------------------------------------------
 * Replaces everything between the qkv and proj Linears of one SwinTransformerBlock:
 * F.pad to x7 (after norm1 => padded tokens carry bias-only q/k/v), torch.roll, window_partition,
 * q*scale, q@k^T + relative_position_bias (+ SW-MSA mask, -100), softmax, @v, window_reverse,
 * roll back, crop.  swin_transformer.py:209-244 + :129-147 + mask construction :369-387.
 * qkv: [B*H*W, 3C] (q|k|v, heads contiguous, head_dim 32); qkv_bias fp32 [3C];
 * rel_bias_table fp32 [(2*7-1)^2, nH]; out: [B*H*W, C].  shift = 0 or 3. */
int omp_swin_window_attn(const void* qkv, const float* qkv_bias, const float* rel_bias_table,
                         void* out, int dtype, int B, int H, int W, int C, int nH, int window,
                         int shift, omp_stream_t s);
/* The same with the relative-position bias EXPANDED once per checkpoint (swin_transformer.py:133-135 re-gathers
 * table[index] on every call): bias_expanded fp32 [nH][64][64] from omp_swin_expand_bias (bias / scale per (query, key),
 * -inf on the padding key slots); rel_bias_table may then be NULL.  out_dtype = dtype, or OMP_BF16X2 with dtype OMP_F32
 * (out is then [B*H*W, 2C] split pairs for the bf16x3 proj GEMM). */
int omp_swin_window_attn2(const void* qkv, const float* qkv_bias, const float* rel_bias_table, const float* bias_expanded,
                          void* out, int dtype, int out_dtype, int B, int H, int W, int C, int nH, int window, int shift,
                          omp_stream_t s);
int omp_swin_expand_bias(const float* rel_bias_table, int nH, float* out, omp_stream_t s);

/* ---- Attention half of a Swin block in one launch (bf16 engine, C = 128 with 4 heads: Swin-B stage 0) ------------
 * Replaces `shortcut = x; x = norm1(x); pad; roll; window_partition; attn(qkv -> softmax -> proj); window_reverse;
 * roll back; crop; x = shortcut + drop_path(x)`, swin_transformer.py:196-247 (SwinTransformerBlock.forward up to the
 * first residual) with WindowAttention.forward (:119-151) inlined: q / k / v, the attention weights and the
 * attention output never reach HBM.  x, out: fp32 residual stream [B*H*W, C] (out may be x: a window reads and writes
 * its own tokens only); ln_gamma / ln_beta: norm1; qkv_w bf16 [3C, C], qkv_b fp32 [3C]; bias_expanded from
 * omp_swin_expand_bias; proj_w bf16 [C, C], proj_b fp32 [C].  The products take bf16 operands exactly where the
 * unfused chain of the bf16 engine rounds (LayerNorm output, q / k / v, P, attention output).
 * OMP_ERR_INVALID for any other C / nH / window. */
int omp_swin_attn_block(const void* x, void* out, const float* ln_gamma, const float* ln_beta, float eps, const void* qkv_w,
                        const float* qkv_b, const float* bias_expanded, const void* proj_w, const float* proj_b, int B, int H,
                        int W, int C, int nH, int window, int shift, omp_stream_t s);
/* The same block for C = 256 with 8 heads (Swin-B stage 1), whose weights fit neither LDS nor registers: wpack is the
 * fragment-major image of qkv.weight and proj.weight written once per checkpoint by model/packing.py::pack_attn_block
 * (bf16 [4 C C]); every wave streams its head's operand fragments from it. */
int omp_swin_attn_block_packed(const void* x, void* out, const float* ln_gamma, const float* ln_beta, float eps, const void* wpack,
                               const float* qkv_b, const float* bias_expanded, const float* proj_b, int B, int H, int W, int C,
                               int nH, int window, int shift, omp_stream_t s);

/* ---- PatchMerging gather + LayerNorm(4C) ------------------------------------------------------------
 * Replaces swin_transformer.py:281-293 (pad to even, 2x2 gather in order (0,0),(1,0),(0,1),(1,1),
 * LN).  x: [B,H,W,C] -> y: [B, ceil(H/2)*ceil(W/2), 4C]; the 4C->2C reduction is a GEMM. */
int omp_patch_merge_gather_ln(const void* x, const float* gamma, const float* beta, void* y,
                              int dtype, int B, int H, int W, int C, float eps, omp_stream_t s);
/* the same with distinct input / output types: fp32 residual stream in, bf16 or split-bf16 (OMP_BF16X2) rows out */
int omp_patch_merge_gather_ln2(const void* x, int x_dtype, const float* gamma, const float* beta, void* y, int y_dtype,
                               int B, int H, int W, int C, float eps, omp_stream_t s);

/* ---- fp32 rows -> split-bf16 pair rows (OMP_BF16X2) ---------------------------------------------------------------
 * y[r, c] = hi = bf16(x[r, c]); y[r, C + c] = lo = bf16(x[r, c] - hi)            (triple = 0: [hi | lo], ldy >= 2C)
 * triple = 1 writes [hi | hi | lo] (ldy >= 3C): the W-side image of an activation (K / V^T projection with swapped
 * operands).  Producers that can write split rows themselves do (omp_layernorm, omp_gemm_bias_act, omp_swin_window_attn2);
 * this is the conversion for the others (FPN output, decoder memory).  No reference counterpart: storage format. */
int omp_split_bf16(const float* x, int64_t ldx, void* y, int64_t ldy, int64_t rows, int C, int triple, omp_stream_t s);

/* ---- FPN top-down + resample + concat, evaluated only where input_proj samples it -------------
 * Replaces fpn.py:25-44 (nearest top-down adds, bilinear resample of p2/p4/p5 to c3's grid,
 * concat) fused with the stride-`stride` sampling of omniparser.py:15,31.  l2..l5 are the four
 * lateral 1x1-conv outputs (token-major, 256 channels); out: [B, ceil(h3/stride)*ceil(w3/stride),
 * 1024] ordered (p2,p3,p4,p5). */
int omp_fpn_fuse(const void* l2, const void* l3, const void* l4, const void* l5, void* out,
                 int dtype, int B, int h2, int w2, int h3, int w3, int h4, int w4, int h5, int w5,
                 int stride, omp_stream_t s);

/* ---- Padding mask at a feature level: nearest-neighbour resize, exactly F.interpolate(mask[None].float(), size=(h, w))
 * .to(torch.bool)[0] of swin_transformer.py:621 (source index = min(floor(dst * (float)H / h), H - 1)).
 * in: uint8 [B,H,W] (1 = padding), out: uint8 [B,h,w]. */
int omp_mask_nearest(const uint8_t* in, uint8_t* out, int B, int H, int W, int h, int w, omp_stream_t s);

/* ---- Sine position embedding ---------------------------------------------------------------------
 * Replaces PositionEmbeddingSine.forward (normalize=True), backbone/position_embedding.py:24-44.
 * mask: [B,h,w] uint8 (1 = padding); pos: [B, h*w, 2*npf] token-major, (y feats | x feats). */
int omp_sine_posembed(const uint8_t* mask, void* pos, int dtype, int B, int h, int w, int npf,
                      float temperature, omp_stream_t s);

/* ---- Decoder step pieces (KV-cached restatement of transformer.py:74-100,430-454) -----------------
 * All rows of a phase sit at the same sequence position *d_pos (device int32). */

/* x = LN(word_emb[seq[r, *d_pos]] + pos_tab[*d_pos]); writes fp32 x and/or y (dtype).
 * Replaces DecoderEmbeddings.forward, transformer.py:302-328. */
int omp_dec_embed_ln(const int32_t* seq, int seq_ld, const int32_t* d_pos, const float* word_emb,
                     const float* pos_tab, const float* gamma, const float* beta, float* x, void* y,
                     int y_dtype, int R, int d, float eps, omp_stream_t s);

/* Append this step's k,v (from qkv [R,3d]) to the caches [R,Lmax,d] at *d_pos and attend the new
 * query over positions 0..*d_pos (causal self-attention with KV cache; head_dim = d/nH = 64).
 * Replaces self_attn of transformer.py:412-414/:438-440 + generate_square_subsequent_mask. */
int omp_dec_self_attn_step(const void* qkv, void* kcache, void* vcache, void* out,
                           const int32_t* d_pos, int dtype, int R, int nH, int d, int Lmax,
                           omp_stream_t s);

/* Cross attention of R query rows over per-image memory K / V^T in the head-blocked slabs written by
 * omp_gemm_bias_act(OMP_STORE_KBLK / OMP_STORE_VBLK):  K [image][head][Mpad][64],
 * V^T [image][head][Mpad/KB][64][KB] (KB = 32 bf16 / 16 f32), img_stride = nH*Mpad*64 elements.
 * Rows are grouped by image: groups[g] = {row0, nrows <= 16*q_tiles, image}; a workgroup streams its
 * key range of one (image, head) once for all rows of the group.  Keys are cut in n_split workgroup
 * splits (power of two <= 16) x 4 waves; n_split > 1 goes through `partial`
 * (fp32 [R][nH][n_split][68]) and a merge kernel.  The memory is never replicated per query row.
 * dtype OMP_BF16X2 (round 4, the parity engine): K / Vt are SPLIT-PLANE slabs -- every 32-key block is [hi plane | lo plane] of
 * bf16 (K [image][head][Mpad/32][2][32][64], V^T [image][head][Mpad/32][2][64][32], img_stride = nH*Mpad*128 bf16 elements; the
 * bytes of the fp32 slabs, written by omp_gemm_bias_act with out_dtype OMP_BF16X2 and a blocked store mode), q and out are fp32
 * rows; scores and values run as three bf16 matrix-core products each, i.e. fp32-grade results at the HBM rate.
 * Replaces multihead_attn of transformer.py:416-420/:442-446 with memory.repeat (transformer.py:88-96). */
int omp_dec_cross_attn_step(const void* q, int64_t ldq, const void* K, const void* Vt, int64_t img_stride,
                            int Mpad, const uint8_t* key_mask, const int32_t* groups, int n_groups,
                            int q_tiles, int R, float* partial, void* out, int64_t ldo, int dtype, int M,
                            int nH, int n_split, omp_stream_t s);

/* Greedy sampling of one step from logits [R, ld] fp32: softmax over the support, candidate
 * filtering, argmax, probability; appends the token at seq[r, *d_pos + 1], the probability at
 * probs[r, *d_pos + 1], maintains finished/lengths for the point decoder, then (if advance)
 * increments *d_pos.  Replaces transformer.py:106-129 (pt), :258-263 (poly), :272-282 (rec) and
 * the KIE variants :154-183.  `step0` = sequence position of the first generated token.
 * A wave per row; beyond 1024 rows (OMP355_SAMPLE_BLOCK_MAX_ROWS) with 16-byte aligned rows (ld % 4 == 0) and vocab <= 2048 the row is held in
 * registers (one pass over the logits).  Probabilities may differ in the last bit between the kernels (summation order); tokens do not. */
typedef struct {
  int32_t kind;        /* OMP_DEC_* */
  int32_t num_bins, pt_eos, poly_eos, rec_eos, vocab;
  int32_t vie_categories, infer_vie; /* infer_vie: period-3 pt pattern; poly/rec slice class logits */
  int32_t suppress_eos;              /* forced-length benchmarking: EOS is never selectable */
  int32_t step0;
} omp_sample_cfg;
int omp_head_softmax_mask_argmax(const float* logits, int ld, int R, const omp_sample_cfg* cfg,
                                 int32_t* seq, float* probs, int seq_ld, int32_t* finished,
                                 int32_t* lengths, int32_t* d_pos, int advance, omp_stream_t s);

/* ---- Results of an engine call -> padded tensors (payload of the per-call all-gather) --------------------------
 * Replaces the per-image result assembly of engine/val.py:39-60 for a batch: instance n of image b is row row0[b] + n of
 * points [R,2], poly [R, poly_ld >= 32], rec [R, rec_ld >= rec_len], rec_probs [R, prob_ld] (device int32 / fp32, the
 * decoders' own buffers); ids [B, N, 2 + 32 + rec_len] int32 = point | polygon | recognition tokens, probs [B, N, rec_len];
 * instances beyond counts[b] (or N) are zero. */
int omp_pack_spotting(const int32_t* points, const int32_t* poly, int poly_ld, const int32_t* rec, int rec_ld,
                      const float* rec_probs, int prob_ld, const int32_t* row0, const int32_t* counts, int B, int N,
                      int rec_len, int32_t* ids, float* probs, omp_stream_t s);

/* ---- One decoder, many steps (the launch-bound inner loop runs from C++, optionally as a
 *      hipGraph replay).  Replaces Transformer.decode driven by decode_pt_seq / the poly and rec
 *      loops, transformer.py:74-141,252-284. */
typedef struct {
  const void* sa_in_w;         /* [3d,d] */
  const float* sa_bias_tab;    /* [Pmax,3d]: in_proj_bias + pos_tab @ [Wq;Wk;0]^T */
  const void* sa_out_w;
  const float* sa_out_b;
  const void* ca_q_w;          /* [d,d] */
  const float* ca_qbias_tab;   /* [Pmax,d] */
  const void* ca_out_w;
  const float* ca_out_b;
  const void* ff1_w;
  const float* ff1_b;
  const void* ff2_w;
  const float* ff2_b;
  const float *n1_g, *n1_b, *n2_g, *n2_b, *n3_g, *n3_b;
  void* kcache;                /* [R,Lmax,d] */
  void* vcache;
  const void* crossK;          /* this layer's K slab   [B][nH][Mpad][64] */
  const void* crossVt;         /* this layer's V^T slab [B][nH][Mpad/KB][64][KB] */
  /* row-owner chains (omp_decoder_plan.rows_fused, see omp_dec_rows_mid / omp_dec_rows_ffn): packed weight streams of this layer */
  const void* rows_mid;        /* sa_out_w, ca_q_w.  Bound on EVERY layer of a bf16 pre-norm plan WITHOUT rows_fused (d_model 512, 8 heads, more than 63 rows), it
                                  makes the launch-per-Linear step run out-projection + residual, norm2 and the cross-attention query as ONE omp_dec_rows_mid
                                  launch (16-row workgroups) instead of three; NULL: three launches */
  const void* rows_ffn;        /* ca_out_w, ff1_w / ff2_w in 16 chunks, then the NEXT layer's sa_in_w -- the last layer: h0_w, h1_w, h2_w */
  /* bytes between the streams of consecutive waves, AS RETURNED BY THE PACKER that wrote the stream (model/packing.py::pack_rows_*: fragments per
   * wave x 1 KB).  The kernels know how many fragments each chain consumes; omp_decoder_run refuses a plan whose strides differ (a packer or
   * layout change must fail loudly, not mis-stream weights). */
  int64_t rows_mid_stride, rows_ffn_stride;
} omp_dec_layer;

typedef struct {
  int32_t dtype, n_layers, d_model, n_heads, d_ff, vocab, pre_norm;
  int32_t R, Lmax, M, Mpad, n_tiles, q_tiles, n_split, n_prompt; /* n_tiles = row groups, see omp_dec_cross_attn_step */
  /* gemm_x3 = 1 (dtype OMP_F32, pre_norm, R > 64: the many-row phases of the bf16x3 engine): every matrix pointer below
   * (sa_in_w ... ff2_w, h0_w ... h2_w) is the [out, 3 * in] bf16 image [w_hi | w_hi | w_lo] of the fp32 weight and the step's
   * products run as split-bf16 products (omp_gemm_args.a_wrap); activations, caches, slabs and every other kernel stay fp32. */
  int32_t gemm_x3;
  /* kv_split = 1 (dtype OMP_F32): crossK / crossVt are split-plane slabs (OMP_BF16X2 with OMP_STORE_KBLK / OMP_STORE_VBLK, 32-key
   * blocks of [hi plane | lo plane]); q and the attention output stay fp32 (with gemm_x3 the kernels write the out-projection's
   * pair rows themselves).  kv_img_stride then counts bf16 elements: nH * Mpad * 128. */
  int32_t kv_split;
  /* rows_fused = 1 (dtype OMP_BF16, or OMP_F32 with gemm_x3: the parity engine's chains over split operands; pre_norm, d_model 512, d_ff 2048,
   * 8 heads; the host sets it for phases of thousands of rows): the
   * Linear layers of a step run as row-owner chains -- omp_dec_rows_ffn(embedding | q k v), then per layer self-attention,
   * omp_dec_rows_mid, cross-attention, omp_dec_rows_ffn -- 18 launches per step instead of 50; layers[l].rows_mid / rows_ffn and
   * rows_embed (layer 0's sa_in_w) are the packed streams (model/packing.py::pack_rows_*). */
  int32_t rows_fused;
  /* XCD placement of the chains' workgroups (round 6; speed only, never correctness).  Bit x set = XCD x may run them; 0 = all eight.  The
   * dispatcher is observed to place block b on XCD b % 8: a chain launch of at most 32 x popcount(mask) workgroups is launched with the blocks of
   * the other XCDs retiring at once, so that its weight stream fills popcount(mask) private L2s instead of eight (the polygon decoder's chains
   * 0x0F, the recognition decoder's 0xF0: each weight set lives in four L2s).  Larger launches ignore it. */
  int32_t rows_xcd_mask;
  float eps;
  const void* rows_embed;
  int64_t rows_embed_stride;   /* as layers[].rows_*_stride */
  omp_dec_layer layers[OMP_MAX_DEC_LAYERS];
  const float *word_emb, *pos_tab, *emb_g, *emb_b, *fn_g, *fn_b;
  const void *h0_w, *h1_w, *h2_w;
  const float *h0_b, *h1_b, *h2_b;
  int64_t kv_img_stride;
  const uint8_t* key_mask;
  const int32_t* tiles;
  /* state */
  int32_t* seq;
  int32_t seq_ld;
  int32_t* d_pos;  /* int32[2]: {current position, ticket word of the sampling kernel (zero-initialised by the caller)} */
  float* probs;
  int32_t* finished;
  int32_t* lengths;
  /* scratch */
  float* x;      /* [R,d] fp32 residual stream */
  float* x2;     /* [R,d] fp32 (post-norm temp) */
  void* y;       /* [R,d] */
  void* qkv;     /* [R,3d] */
  void* att;     /* [R,d] */
  void* q;       /* [R,d] */
  void* ffh;     /* [R,d_ff] */
  void* hh0;     /* [R,d] */
  void* hh1;     /* [R,d] */
  float* partial;
  float* logits; /* [R,vocab] */
  omp_sample_cfg sample;
} omp_decoder_plan;

/* Runs steps for input positions first_pos .. first_pos+n_steps-1 (host-side counter must match
 * *d_pos).  Positions < n_prompt-1 are prompt prefill (no head / sampling).  graph_slot >= 0
 * caches a hipGraph of the sampling step under that slot (call omp_decoder_graph_reset when the
 * plan's pointers or shapes change); graph_slot < 0 launches eagerly. */
int omp_decoder_run(const omp_decoder_plan* plan, int first_pos, int n_steps, int graph_slot,
                    omp_stream_t s);
int omp_decoder_graph_reset(int graph_slot);
/* The capture gate (round 6, ABI 21; no reference counterpart: engine/val.py:19-35 runs one image at a time on one thread).  omp_decoder_run
 * captures a graph on the caller's stream the first time a graph_slot is used.  While a stream captures, HIP fails -- and invalidates the
 * capture on -- hipEventSynchronize / hipEventQuery / hipStreamWaitEvent of every event LAST RECORDED IN THAT STREAM, events of earlier,
 * uncaptured work included (hipErrorCapturedEvent / hipErrorStreamCaptureIsolation).  A host thread that waits on an event recorded in a
 * stream which ANOTHER thread drives through omp_decoder_run must bracket the event call with enter / leave: a capture holds the same
 * process-wide mutex from begin to instantiate.  Keep the bracket non-blocking (query, stream-wait; poll instead of hipEventSynchronize) and
 * never call omp_decoder_run inside it.  Not needed for events of the calling thread's own streams.  Always OMP_OK. */
int omp_capture_gate_enter(void);
int omp_capture_gate_leave(void);
/* Two decoders' many-row phases as ONE interleaved schedule (round 6; the polygon and recognition loops of Transformer.forward,
 * transformer.py:252-284, which depend on the points only).  Both plans must be rows_fused plans with the same number of layers.
 * Positions first_pos .. first_pos + max(n_steps_a, n_steps_b) - 1: while both decoders have steps left, every cross-attention kernel of
 * both goes to the ONE stream sx in the order a[0], b[0], a[1], b[1] ..., chained by events to the decoders' own streams sa / sb, which
 * carry everything else (self-attention, the row-owner chains, sampling): one decoder's chains -- matrix-core work on half the compute
 * units -- run beside the OTHER decoder's HBM-bound cross-attention instead of beside its own kind.  sa and sb should be created with a
 * HIGHER priority than sx (a chain workgroup needs a whole CU's LDS: the queue priority decides who is placed when a launch drains).  The
 * longer decoder finishes alone on its stream.  Eager launches; three distinct non-default streams; the caller joins sa and sb afterwards.
 * Results are the same bits as two omp_decoder_run calls (the same kernels on the same operands: only WHEN they run changes). */
int omp_decoder_run_pair(const omp_decoder_plan* plan_a, const omp_decoder_plan* plan_b, int first_pos, int n_steps_a, int n_steps_b,
                         omp_stream_t sa, omp_stream_t sb, omp_stream_t sx);

/* ---- Many-row decoder phases: the Linear chain between two attention kernels as ONE launch (round 5, csrc/dec_rows.hip) -----------
 * Replaces, for phases of thousands of rows (polygon / recognition decoders of a large engine call), the per-Linear launches of
 * TransformerDecoderLayer.forward_pre, transformer.py:430-454, at d_model 512 / d_ff 2048 / bf16 operands:
 *   omp_dec_rows_mid:  x += att Wo^T + bo  (self-attention out-projection + residual, :438-440);
 *                      q  = bf16(LayerNorm2(x) Wq^T + qbias_tab[*d_pos])  (:441-446, query side of multihead_attn)
 *   omp_dec_rows_ffn:  prologue 0: x1 = x + att Wo^T + bo (cross-attention out-projection + residual, :442-447);
 *                                  x  = x1 + relu(LayerNorm3(x1) W1^T + b1) W2^T + b2  (:448-453)
 *                      prologue 1: x  = LayerNorm(word_emb[seq[r, *d_pos]] + pos_tab[*d_pos])  (DecoderEmbeddings, :302-328: layer 0)
 *                      tail 0:     qkv = bf16(LayerNorm_t(x) Win^T + bias_tab[*d_pos])  (the NEXT layer's norm1 + in_proj, :437-440)
 *                      tail 1:     logits = h2(relu(h1(relu(h0(LayerNorm_t(x))))))  (decoder norm :374 + prediction head, block/mlp.py:11-13)
 * A workgroup owns omp_dec_rows_tile() consecutive rows (operand tile in LDS, residual / accumulators in registers) and streams the
 * chain's weights; every wave w (8 per workgroup) walks its own linear stream of 1 KB matrix-core fragments starting at
 * wstream + w * wave_stride, written once per checkpoint by model/packing.py::pack_rows_* in consumption order:
 *   a product of N = 512 (128) output features over K: for each k-step of 32, for each of the wave's 4 (1) feature tiles of 16 --
 *   features 64 w + 16 t .. (16 w ..) of the pass --: fragment[lane][8] = W[f0 + (lane & 15)][32 ks + 8 (lane >> 4) .. + 8];
 *   the FFN interleaves, per chunk of 256 hidden units, linear1's pass (N = 256: features 32 w + 16 t, t = 0..1, K = 512) with linear2's
 *   (N = 512, K = 256);
 *   the vocabulary projection is padded with zero rows to a multiple of 128 features (512-feature passes, then 128-feature passes).
 * wave_stride must equal (fragments of the chain per wave) x 1 KB -- what the packer returns; the entry points refuse anything else.
 * The buffer carries PF x 1 KB of slack behind the last wave's stream, PF = fragments the kernels keep in flight (the ring runs ahead of the
 * stream's end): 8 KB for the bf16 chains, 16 KB for the x3 chains (pairs of hi / lo fragments); model/packing.py allocates 16 KB for both.
 * att: bf16 [R, 512]; x: fp32 [R, 512] in place; q: bf16 [R, 512]; qkv: bf16 [R, 1536]; logits: fp32 [R, vocab], vocab % 4 == 0. */
typedef struct {
  int32_t R;
  float eps;
  const int32_t* d_pos;
  float* x;
  const void* att;
  const void* wstream;
  int64_t wave_stride;
  const float* out_b;
  const float *ln_g, *ln_b;        /* mid: norm2; ffn prologue 0: norm3 */
  const float* qbias_tab;          /* mid: [Pmax, 512] */
  void* q;                         /* mid */
  int32_t prologue, tail;          /* ffn */
  const float *ff1_b, *ff2_b;
  const int32_t* seq;
  int32_t seq_ld;
  const float *word_emb, *pos_tab, *emb_g, *emb_b;
  const float *lnt_g, *lnt_b;      /* the LayerNorm in front of the tail */
  const float* bias_tab;           /* tail 0: [Pmax, 1536] */
  void* qkv;
  const float *h0_b, *h1_b, *h2_b; /* tail 1 */
  float* logits;
  int32_t vocab;
  /* x3 = 1: the PARITY engine's chains (csrc/dec_rows_x3.hip): every product as three bf16 matrix-core products of split operands.  att is then
   * split pairs bf16 [R, 1024] = [hi | lo] (omp_split_bf16 / the split-plane cross-attention), q / qkv are fp32 [R, 512] / [R, 1536], and
   * wstream carries per (k-step, feature tile) the fragment of w_hi then of w_lo (twice the fragments; model/packing.py, x3=True); a
   * workgroup owns 48 rows. */
  int32_t x3;
  int32_t xcd_mask;                /* XCDs whose CUs may run the launch's workgroups (omp_decoder_plan.rows_xcd_mask); 0 = all */
} omp_dec_rows_args;
int omp_dec_rows_mid(const omp_dec_rows_args* a, omp_stream_t s);
int omp_dec_rows_ffn(const omp_dec_rows_args* a, omp_stream_t s);
int omp_dec_rows_tile(void);   /* the LARGEST tile: 80 rows per workgroup (the Swin chains always; decoder launches take 16 x {1..5} rows: the smallest
                                  tile that keeps the launch on half the device's compute units, csrc/dec_rows.hip rows_rtt) */

/* Cross-attention memory projection as a row-owner stream kernel (csrc/kv_rows.hip, round 5) -- replaces, for bf16 engines with d_model 512,
 * 8 heads and M % 64 == 0, the two omp_gemm_bias_act launches with OMP_STORE_KBLK / OMP_STORE_VBLK that computed what nn.MultiheadAttention
 * recomputes per step and instance (model/transformer.py:88-96, 442-446).  rows: [B * M, 512] bf16 memory rows (memory + pos for K, memory
 * for V^T); wstream: the [n_slabs * 512, 512] projection weights of all (decoder, layer) pairs as per-wave fragment streams
 * (model/packing.py::pack_kv_rows_k -- dims permuted per head so that a lane stores 16 consecutive dims -- / pack_kv_rows_v), wave_stride
 * bytes apart; bias [n_slabs * 512] fp32; out: the slab base, K [slab][B][8][Mpad][64] (vt = 0) or V^T [slab][B][8][Mpad / 32][64][32] in the
 * key-slot order of OMP_STORE_VBLK (vt = 1).  The padded tail (keys >= M) is not written.  Bit-identical to the tiled GEMM path. */
int omp_kv_project_rows(const void* rows, const void* wstream, int64_t wave_stride, const float* bias, void* out, int B, int M, int Mpad,
                        int n_slabs, int vt, omp_stream_t s);

/* ---- MGP-STR recogniser (reference: OCR/MGP-STR; BASELINE config 5) ------------------------------------
 * The ViT-B encoder reuses omp_layernorm / omp_gemm_bias_act / omp_dec_cross_attn_step (a ViT layer's k and v
 * projections write the blocked slabs, an image's 257 tokens are row groups of that image); these three entry
 * points cover what is specific to MGP-STR. */

/* bf16 self-attention of a ViT layer, all heads of all images in one launch: out[b*T + i, h*64 + :] =
 * softmax_j(q_i . k_j / 8) v_j.  Replaces timm Attention.forward (the blocks modules/mgp_str.py:71-74 runs).
 * q [B*T, ldq] (head h at columns h*64), K / Vt = one layer's blocked slabs [B][nH][Mpad][64] and
 * [B][nH][Mpad/32][64][32] as written by omp_gemm_bias_act with OMP_STORE_KBLK / OMP_STORE_VBLK (kv_key_block 32),
 * padded keys zero.  Built for dtype OMP_BF16 and Mpad == 288 (T <= 288, MGP-STR: 257); other shapes return
 * OMP_ERR_UNSUPPORTED and are served by omp_dec_cross_attn_step on the same slabs. */
int omp_vit_attn(const void* q, int64_t ldq, const void* K, const void* Vt, int Mpad, void* out, int64_t ldo,
                 int dtype, int B, int T, int nH, omp_stream_t s);
/* The same attention on ONE token-major projection qkv [B T, 3 nH 64] = [q | k | v] (timm's fused qkv Linear, modules/mgp_str.py:72-73; round 6): the key
 * rows reach LDS by strided DMA and the kernel builds the blocked V^T image itself, so the projection is one plain product of N = 3 nH 64 instead of a
 * plain one and two with blocked-slab epilogues (the V^T slab store costs 2.7x a plain product at 257 tokens per image).  bf16, T <= 288. */
int omp_vit_attn_qkv(const void* qkv, int64_t ld, void* out, int64_t ldo, int dtype, int B, int T, int nH, omp_stream_t s);

/* Patch embedding + cls token + position embedding.  Replaces timm PatchEmbed (Conv2d(3,E,4,4) -> flatten ->
 * transpose) and modules/mgp_str.py:66-70.  img NCHW fp32 [B,3,H,W] (H, W multiples of 4); w [E,3,4,4], bias [E],
 * cls [E], pos [(H/4)*(W/4)+1, E] fp32; out [B, (H/4)*(W/4)+1, E] token-major, token 0 = cls. */
int omp_vit_patch_embed(const float* img, const float* w, const float* bias, const float* cls, const float* pos,
                        void* out, int out_dtype, int B, int H, int W, int E, omp_stream_t s);

/* A^3 module core, modules/token_learner.py:27-31: maps[b,s,:] = softmax over the T tokens of sel[b,:,s];
 * pooled[b,s,:] = sum_i maps[b,s,i] * feat[b,i,:].  sel fp32 [B*T, ld_sel] (S <= 28 columns), feat [B*T, C]
 * (dtype), pooled fp32 [B*S, C], attn fp32 [B,S,T] or NULL. */
int omp_a3_pool(const float* sel, int ld_sel, const void* feat, int dtype, float* pooled, float* attn, int B, int T,
                int S, int C, omp_stream_t s);

/* Greedy decoding WITHOUT materialising the logits (round 6; MGP-STR's BPE / WordPiece heads: 13 824 rows x 50 257 / 30 522 classes are 2.8 / 1.7 GB
 * of fp32 per forward that the head product writes and the arg-max pass reads back -- test_final.py:145-170 needs only the greedy id and its
 * softmax probability).  omp_gemm_bias_act with store_mode = OMP_STORE_ROWSTAT (out_dtype f32; no residual / activation / second destination /
 * transposed output) runs the product on 128 x 128 tiles and, instead of the logits, stores per (row, column tile) the tile's statistics into
 * C = float [M][2 ceil(N / 128)][4], one record per 64-column half tile: {maximum of logit + bias, sum of exp(logit - maximum), index of the
 * maximum (int32 bits, lowest index on ties), 0}; a half without a valid column holds {-inf, 0, .}.  omp_row_stat_merge folds a row's tiles: ids[r] = arg max, prob[r] = 1 / sum_tiles s_t exp(m_t - max) -- the values
 * omp_row_argmax_prob computes from the full row (same product bits; the probability differs by the rounding of a different summation order). */
int omp_row_stat_merge(const float* stats, int R, int n_records, int32_t* ids, float* prob, omp_stream_t s);   /* n_records = 2 ceil(N / 128) */
/* Greedy id and its softmax probability for every row of logits fp32 [R, ld] (V columns used).  Replaces
 * topk(1) + softmax(...).max(dim=2) of test_final.py:145-170. */
int omp_row_argmax_prob(const float* logits, int64_t ld, int R, int V, int32_t* ids, float* prob, omp_stream_t s);

/* ---- context: ALL mutable state of the library ---------------------------------------------------
 * An omp_ctx holds the table of hipGraphs that omp_decoder_run captures under its graph_slot ids, plus the development
 * state declared in csrc/omp355_debug.h (kernel selectors, trace buffers, measurement brackets).  Every entry point
 * works on the CURRENT context of the calling host thread: the process default context until the thread calls
 * omp_ctx_make_current.  Two engines in one process (or one per pipeline lane) get independent state by creating
 * their own; destroying a context destroys the graphs it captured.  Beyond the context the library keeps only a
 * thread-local error string.  Scratch memory is owned by the caller (every buffer is an argument; omp_decoder_plan
 * lists the decoder's), so there is no omp_workspace_bytes to query. */
typedef struct omp_ctx omp_ctx;
int omp_ctx_create(omp_ctx** out);
int omp_ctx_destroy(omp_ctx* ctx);          /* not the default context */
int omp_ctx_make_current(omp_ctx* ctx);     /* NULL = back to the process default context */
omp_ctx* omp_ctx_current(void);             /* the calling thread's context (never NULL) */

/* ---- test-time image pre-processing (the step before the hot path) -----------------
 * Replaces dataset/transforms.py:249-298 (RandomResize([test_min_size], test_max_size) = Pillow bilinear resize
 * through torchvision F.resize), :312-322 (ToTensor, Normalize) and the padding / mask construction of
 * utils/nested_tensor.py:37-54 for ONE image of a batch.  src: uint8 HWC RGB on the device.  xbounds/xcoef,
 * ybounds/ycoef: Pillow's coefficient tables ([out][2] = first source index, count; [out][ks] 22-bit fixed point;
 * utils/preprocess.py builds them; may be NULL for an axis whose size does not change).  lut: float [3][256] =
 * ((p / 255) - mean_c) / std_c.  dst: this image's [3, dst_h, dst_w] fp32 slice of the batch tensor; mask: its
 * [dst_h, dst_w] uint8 slice (1 = padding) or NULL.  Bit-exact with the reference pipeline. */
int omp_resize_normalize_pad(const uint8_t* src, int64_t src_pitch, int in_h, int in_w, const int32_t* xbounds,
                             const int32_t* xcoef, int ksx, const int32_t* ybounds, const int32_t* ycoef, int ksy,
                             const float* lut, float* dst, uint8_t* mask, int out_h, int out_w, int dst_h,
                             int dst_w, omp_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* OMP355_H */
