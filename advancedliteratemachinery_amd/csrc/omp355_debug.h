/* libomp355 -- DEVELOPMENT hooks.  Not part of the product ABI (include/omp355.h): kernel selectors for A/B measurements
 * and cross-check kernels, per-workgroup trace buffers, hipEvent measurement brackets (bench.py's roofline legs), streams
 * on a CU subset (the negative overlap experiment of DESIGN.md 5) and the teacher-forced logits step of the parity tests.
 * The symbols are exported by libomp355.so for this repository's tests / tools / bench only; all of their state lives in
 * the omp_ctx of the calling thread. */
#ifndef OMP355_DEBUG_H
#define OMP355_DEBUG_H

#include "../../include/omp355.h"

#ifdef __cplusplus
extern "C" {
#endif

int omp_debug_swin_mlp_variant(int v); /* alternative (rows per wave, waves, ring depth) instantiations of the fused MLP; 100 = traced default */
int omp_debug_swin_mlp_trace(void* buffer); /* uint64 [workgroups][8] cycle sums written by variant 100 (csrc/mlp.hip); while set, omp_swin_attn_block launches its traced instantiation into the same buffer (csrc/swin_block.hip) and omp_swin_rows_block writes uint64 [workgroups][16] phase sums of wave 0 (csrc/dec_rows.hip: 0 whole, 1 prologue, 2 out-projection, 3 residual + LayerNorm, 4 linear1, 5 / 7 the two barriers of a chunk, 6 activation + LDS writes, 8 linear2, 9 x store, 10 tail LayerNorm, 11 tail products) */

/* ---- streams on a subset of the compute units (engine/pipeline.py: HBM-bound decoder phases of one engine call
 * next to the matrix-core-bound encoder of another) ---------------------------------------------------------------
 * mask: n_words x 32 bits, bit i = compute unit i in the runtime's numbering (hipExtStreamCreateWithCUMask).
 * omp_debug_where: every workgroup of a short probe grid records (XCC_ID, HW_ID) -> out[2 * n_workgroups]. */
int omp_stream_create_cu_mask(const uint32_t* mask, int n_words, omp_stream_t* out);
int omp_stream_destroy(omp_stream_t s);
int omp_debug_where(int32_t* out, int n_workgroups, omp_stream_t s);

/* Measurement hooks (bench.py roofline legs): hipEvent-bracket every EAGERLY launched kernel of a class on its launch
 * stream.  Classes (bit c of `mask`): 0 = decoder cross-attention kernels, 1 = large GEMMs of the encoder (gemm_256 / gemm_4w / gemm_4w_p /
 * gemm_dma 128x128 at M >= 32768 rows), 2 = fused Swin MLP, 3 = the same GEMM kernels on decoder-phase rows (M < 32768; round 4).  omp_prof_read_class returns total milliseconds, launch count and the summed work of the
 * bracketed launches (flops for classes 1, 2 and 3; 0 for class 0, whose bytes the caller computes).  omp_prof_read =
 * class 0 (kept for round-1 callers). */
int omp_prof_enable(int mask);
int omp_prof_read(double* total_ms, int64_t* count);
int omp_prof_read_class(int cls, double* total_ms, int64_t* count, double* work);
/* classes 1 and 2: summed algorithmic HBM bytes of the bracketed launches and the sum over launches of
 * max(flops / 2.5 PFLOP/s, bytes / 8 TB/s) -- the time they would take on their own rooflines */
int omp_prof_read_roofline(int cls, double* bytes, double* roofline_seconds);
/* GEMM kernel selector: 0 auto; 3 row-streaming; 4 split-K small-M; 5 / 6 DMA 128x128 / 64x64; 9 gemm_256; 10 (= 11) gemm_4w; 16 gemm_4w_r
 * (weights streamed into registers; K % 256 == 0); 20 gemm_4w_p (the same, persistent over tiles, register-only epilogue; M, N, K
 * multiples of 256); 22 its fused three-product instantiation for bf16x3 operands.  Wrong results, valid timing: 12..14 gemm_4w without DMA / fragment reads / MFMAs, 17 gemm_4w_r without MFMAs, 21
 * gemm_4w_p with 2/3 of its operand bytes.  Traces: 15 gemm_dma<128,128,2>, 18 gemm_4w_r (omp_debug_set_gemm_trace). */
int omp_debug_force_gemm_kernel(int which);
/* the selector omp_gemm_bias_act would take for these arguments (no launch, no device access: host logic, tests/test_host_logic.py);
 * < 0: the error code of its argument checks.  Like every hook of this header it acts on the calling thread's current context: do not
 * call it while another thread launches through the same context (its launches would be answered instead of executed) */
int omp_debug_gemm_choice(const omp_gemm_args* args);
/* development: device buffer uint64 [n_workgroups][8] that omp_debug_force_gemm_kernel(15 / 18) fills with s_memtime stamps
 * per workgroup: 0 start, 1 first K tile landed, 2 K loop done, 3 accumulators in LDS (18: every wave done), 4 stores retired, 5 XCC id */
int omp_debug_set_gemm_trace(void* buffer, int64_t n_workgroups);
int omp_debug_swin_attn_impl(int which); /* 0 = matrix-core kernel (default), 1 = scalar cross-check kernel, 2 = matrix cores with per-score table lookups */
int omp_debug_dec_fused(int mode);       /* 0 = fused few-row decoder step kernels where they apply (default), 1 = one launch per op everywhere (A/B, cross-check) */
int omp_debug_self_attn_impl(int which); /* 0 auto, 1 = one wave per (row, head), 2 = one wave per row (all 8 heads) */
int omp_debug_rows_tile_choice(int R, int mid); /* host logic, no launch: rows per workgroup (16..80) a decoder row-owner launch of R rows takes; mid != 0: omp_dec_rows_mid (also 16 rows) */
int omp_debug_rows_tile(int rtt);        /* decoder row-owner chains (bf16): rows per workgroup = 16 x rtt, 0 = chosen by row count (default), 2..5 forced */
int omp_debug_cross_nt(int on);          /* non-temporal K / V^T loads in the cross-attention kernels: 1 = always (default), 2 = only from 32 groups per launch, 0 = never */
int omp_debug_cross_q4(int on);          /* 1 = LDS-ring cross-attention for 33..64 rows/image in 64-key chunks, non-temporal DMA (default), 2 = one 32-key block per step, 4 = chunks with temporal loads, 0 = register-streaming kernel */

/* Single teacher-forced step that also leaves logits in plan->logits (parity tests). */
int omp_decoder_step_logits(const omp_decoder_plan* plan, int pos, omp_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* OMP355_DEBUG_H */
