// Many-row decoder phases (polygon / recognition: thousands of rows at d_model = 512): the Linear layers of a pre-norm decoder layer as
// ROW-OWNER chains, two launches per layer instead of eleven.
//
// Reference: OCR/OmniParser/model/transformer.py:430-454 (TransformerDecoderLayer.forward_pre):
//     x = x + self_attn(norm1(x) ...);  x = x + multihead_attn(norm2(x) ...);  x = x + linear2(relu(linear1(norm3(x))))
// Round 4 ran every Linear as its own tiled GEMM launch with the LayerNorms as launches of their own: at R = 10 240 rows a launch is
// 240-320 tiles on a 256-CU chip -- fill, drain and the operand traffic of small tiles (64 flop per operand byte) put the class at 0.09 of
// the matrix-core peak.  The boundaries between those launches are NOT all-to-all: LayerNorm, residual adds and every
// Linear are row-local, only the two attention kernels mix rows (keys of the row's own cache; keys of the row's image).  So a workgroup
// can OWN rows: it keeps RT = 80 rows resident (bf16 operand tile in LDS, fp32 residual / accumulators in registers) and walks the
// whole chain between two attention kernels, streaming the chain's weights through the matrix cores:
//
//     dec_rows_mid_kernel :  att -> out_proj + bias + x -> x' -> LayerNorm2 -> ca_q + position bias -> q
//     dec_rows_ffn_kernel :  att -> out_proj + bias + x -> LayerNorm3 -> linear1 + ReLU -> linear2 + bias + x -> x'
//                            -> [next layer's LayerNorm1 -> in_proj (q | k | v) + position bias]  |  [final norm -> 3-layer head -> logits]
//                            (prologue variant: token + position embedding + LayerNorm instead of the attention / FFN part: layer 0)
//
// What makes it fast (tools/probe_stream.hip, profiles/r05a_probe_stream.txt): the only operand that moves is the WEIGHT stream.  It is
// the same for every workgroup, so it is an L2 / MALL hit everywhere; a compute unit pulls it at 110 GB/s (29 TB/s over the chip), and
// with 80 resident rows each 1 KB fragment feeds 5 matrix-core instructions: 1.75 PFLOP/s with the loop below, against 0.3-0.6 for the
// tiled kernels on these shapes.  Design points:
//   * 8 waves; wave w owns the output features 64 w .. 64 w + 63 of a 512-feature pass (16 w .. + 15 of a 128-feature pass), so its
//     weight fragments are ITS OWN: they go global -> registers (no LDS, no barrier in the product loop) as one linear stream per wave
//     that the host packs in consumption order (model/packing.py::pack_rows_*): "next fragment" is `base += 1024`;
//   * the stream is prefetched PF = 8 fragments deep in a register ring ACROSS products, LayerNorms and barriers (it depends on no data).
//     hipcc does not keep such a ring in flight (it sinks the loads to their uses and drains vmcnt(0)): the loads are asm statements and
//     the waits are counted by hand (`s_waitcnt vmcnt(PF - 1)`: exactly PF of these loads are outstanding at every use; loads return in
//     order, so extra compiler-issued loads or stores only make the wait more conservative).  The build audit
//     (advancedliteratemachinery_amd/audit.py::audit_dec_rows) refuses a build in which the compiler copies a ring register;
//   * D[feature][row] = W[feature][:] . a[row][:] on 16x16x32 MFMAs: A operand = weight fragment, B operand = 16 rows x 32 k of the
//     resident tile (ds_read_b128, row pitch 1056 B = conflict-free); a lane's accumulator quad = 4 consecutive features of one row;
//   * LayerNorm over accumulator registers: two-pass fp32 statistics, the 8 waves' partial sums meet in LDS (2 barriers), the normalised
//     rows are written over the operand tile in place;
//   * FFN in 8 chunks of 256 hidden units: relu(linear1) of a chunk goes through an LDS tile into linear2's accumulators (two barriers
//     per chunk), which START as x + bias2: the residual add costs nothing;
//   * the row fragments of the next k-step are read from LDS before the matrix-core instructions of the current one.
// Operand rounding is where the launch-per-op path rounds (LayerNorm outputs, attention outputs, hidden activations, q / k / v: bf16;
// residual stream, accumulation, statistics: fp32).
#include <type_traits>
#include <utility>

#include "common.h"

// the parity engine's chains (csrc/dec_rows_x3.hip)
int omp_rows_x3_mid(const omp_dec_rows_args* a, hipStream_t st);
int omp_rows_x3_ffn(const omp_dec_rows_args* a, hipStream_t st);
int omp_rows_x3_swin(const omp_swin_rows_args* a, hipStream_t st);

namespace {

#include "rows_common.inc"

constexpr int PF = 8;                  // weight fragments in flight per wave
constexpr int HC = 256;                // hidden units per FFN chunk
constexpr int A_PITCH = D * 2 + 32;    // operand tile row pitch, bytes: 264 dwords = 8 mod 64 -> the b128 fragment reads are conflict-free
constexpr int H_PITCH = HC * 2 + 32;   // hidden chunk tile row pitch: 136 dwords = 8 mod 64
constexpr int TILE_SLACK = 64;         // behind each tile: the operand prefetch of gemm_pass reads one k-step past the last row

// acc[ft][rt] += W[feature tile ft of this wave][:] . a[row tile rt][:] over KS k-steps of 32; the wave's next NFT * KS stream fragments,
// ordered (k-step, feature tile).  a_lane = operand tile + (lane & 15) * PITCH + (lane >> 4) * 16.
template <int NFT, int KS, int RTT, int PITCH>
__device__ __forceinline__ void gemm_pass(f32x4 (&acc)[NFT][RTT], const char* a_lane, u32x4 (&ring)[PF], Stream& st) {
  static_assert(PF % NFT == 0 && (NFT * KS) % PF == 0 && (PF / NFT) % 2 == 0, "a pass is a whole number of ring revolutions, an even number of k-steps each");
  constexpr int NG = NFT * KS / PF, KPG = PF / NFT;
  // the row fragments of k-step ks + 1 are requested before the matrix-core instructions of k-step ks (two register sets): a wave hides
  // its own LDS latency instead of relying on the SIMD's other wave (the read behind the last k-step lands in the tile's padding)
  bf16x8 bfr[2][RTT];
#pragma unroll
  for (int rt = 0; rt < RTT; ++rt) bfr[0][rt] = *reinterpret_cast<const bf16x8*>(a_lane + rt * 16 * PITCH);
  auto group = [&](int gi) {
    sfor<PF>([&](auto U) {
      constexpr int u = decltype(U)::value, ft = u % NFT, kk = u / NFT;
      if constexpr (ft == 0) {
        const char* ak = a_lane + (gi * KPG + kk + 1) * 64;
#pragma unroll
        for (int rt = 0; rt < RTT; ++rt) bfr[(kk + 1) & 1][rt] = *reinterpret_cast<const bf16x8*>(ak + rt * 16 * PITCH);
      }
      const bf16x8 wf = ws_take<u>(ring);
#pragma unroll
      for (int rt = 0; rt < RTT; ++rt) acc[ft][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, bfr[kk & 1][rt], acc[ft][rt], 0, 0, 0);
      ws_issue<u>(ring, st);
    });
  };
  if constexpr (NG <= 2) {
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) group(gi);
  } else {
#pragma unroll 1
    for (int gi = 0; gi < NG; ++gi) group(gi);
  }
}

// LayerNorm of the RT rows held in accumulator layout (v[ft][rt][r]: row rt * 16 + li, feature 64 w + 16 ft + 4 g + r), written as the
// bf16 operand tile.  Two-pass statistics; the waves' partial sums meet in red[2][NW][RT].  Barriers: after each partial-sum
// store (every wave has then also finished the product that read the tile: it may be overwritten) and after the tile is written.
template <int RTT>
__device__ __forceinline__ void ln_acc_to_tile(const f32x4 (&v)[4][RTT], const float* __restrict__ gam, const float* __restrict__ bet, float eps,
                                               char* tile, float* red, int wave, int li_, int g_) {
  constexpr int RT = RTT * 16;
  // opaque lane coordinates: the LDS addresses below are rebuilt per call (the compiler otherwise keeps the 40 of them alive -- spilled --
  // from one LayerNorm of a kernel to the next, across the FFN loop)
  const int li = opaque(li_), g = opaque(g_);
  float* redl = red + li;                       // + w * RT + rt * 16: immediate offsets
  float* red2l = red + NW * RT + li;
  float mean[RTT], rstd[RTT];
#pragma unroll
  for (int rt = 0; rt < RTT; ++rt) {
    float s = 0.f;
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) s += (v[ft][rt][0] + v[ft][rt][1]) + (v[ft][rt][2] + v[ft][rt][3]);
    s = quad_group_sum(s);
    if (g == 0) redl[wave * RT + rt * 16] = s;
  }
  lds_barrier();
#pragma unroll
  for (int rt = 0; rt < RTT; ++rt) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += redl[w * RT + rt * 16];
    mean[rt] = s * (1.0f / D);
  }
#pragma unroll
  for (int rt = 0; rt < RTT; ++rt) {
    float q = 0.f;
#pragma unroll
    for (int ft = 0; ft < 4; ++ft)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float d = v[ft][rt][r] - mean[rt]; q += d * d; }
    q = quad_group_sum(q);
    if (g == 0) red2l[wave * RT + rt * 16] = q;
  }
  lds_barrier();
#pragma unroll
  for (int rt = 0; rt < RTT; ++rt) {
    float q = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) q += red2l[w * RT + rt * 16];
    rstd[rt] = 1.0f / sqrtf(q * (1.0f / D) + eps);
  }
  char* tl = tile + li * A_PITCH + g * 8;       // + rt * 16 * A_PITCH + (64 w + 16 ft) * 2
#pragma unroll
  for (int ft = 0; ft < 4; ++ft) {
    const int f = wave * 64 + ft * 16 + g * 4;
    const f32x4 gg = *reinterpret_cast<const f32x4*>(gam + f), bb = *reinterpret_cast<const f32x4*>(bet + f);
#pragma unroll
    for (int rt = 0; rt < RTT; ++rt) {
      bf16x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (bf16_t)((v[ft][rt][r] - mean[rt]) * rstd[rt] * gg[r] + bb[r]);
      *reinterpret_cast<bf16x4*>(tl + rt * 16 * A_PITCH + (wave * 64 + ft * 16) * 2) = o;
    }
  }
  lds_barrier();
}

// the attention output rows of this workgroup (bf16 [R, 512]) -> operand tile, by LDS DMA: a row is 1 KB = one wave instruction (64 lanes x
// 16 bytes, LDS destination = wave-uniform base + 16 * lane), RT / 8 instructions per wave, no register staging.  Rows beyond R repeat the
// last row (computed, never stored).  The requests count in vmcnt: the caller waits vmcnt(0) (the ring's first fragments, requested earlier,
// have landed by then) before the barrier that opens the first product.
template <int RTT>
__device__ __forceinline__ void stage_rows(const bf16_t* __restrict__ att, int64_t r0, int R, char* tile, int wave, int lane) {
  constexpr int RT = RTT * 16;
#pragma unroll
  for (int i = 0; i < RT / NW; ++i) {
    const int row = wave * (RT / NW) + i;
    int64_t r = r0 + row;
    if (r > R - 1) r = R - 1;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(att + r * D + lane * 8),
                                     (__attribute__((address_space(3))) void*)(tile + row * A_PITCH), 16, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// Epilogue helpers.  FULL = the workgroup's RT rows all exist (every workgroup but the last one of a ragged launch): no per-row predicates --
// with them hipcc wraps every store in its own exec-mask branch and puts `s_waitcnt vmcnt(0)` in front of each (a store round trip per
// store: the q k v tail of a chain measured 82k cycles instead of 30k, profiles/r05h_kbench_swin_rows_trace.txt).  Callers branch once on
// nrow == RT (wave-uniform).

// acc = acc + bias[f] + x[row][f]  (the residual add of an attention sub-layer); optionally the new x goes back to memory.
// xb = x + r0 * 512 (the workgroup's first row), nrow = rows of this workgroup that exist (R - r0, >= 1)
template <int RTT, bool STORE, bool FULL>
__device__ __forceinline__ void add_bias_residual_t(f32x4 (&acc)[4][RTT], const float* __restrict__ bias, float* __restrict__ xb, int nrow,
                                                    int wave, int li_, int g_) {
  const int li = opaque(li_), g = opaque(g_);
  // the loads of FG feature tiles at a time, all issued before the first is consumed: one memory round trip per group (all four tiles at once
  // where the registers allow: 80 of them; the storing variant also holds its store addresses and takes two groups)
  constexpr int FG = STORE ? 2 : 4;
#pragma unroll
  for (int f0 = 0; f0 < 4; f0 += FG) {
    f32x4 bb[FG], xv[FG][RTT];
#pragma unroll
    for (int u = 0; u < FG; ++u) {
      const int f = wave * 64 + (f0 + u) * 16 + g * 4;
      bb[u] = *reinterpret_cast<const f32x4*>(bias + f);
#pragma unroll
      for (int rt = 0; rt < RTT; ++rt) {
        const int lr = rt * 16 + li;
        const int lc = FULL ? lr : (lr < nrow ? lr : nrow - 1);
        xv[u][rt] = *reinterpret_cast<const f32x4*>(xb + lc * D + f);
      }
    }
#pragma unroll
    for (int u = 0; u < FG; ++u) {
      const int ft = f0 + u, f = wave * 64 + ft * 16 + g * 4;
#pragma unroll
      for (int rt = 0; rt < RTT; ++rt) {
        const int lr = rt * 16 + li;
        const f32x4 v = {acc[ft][rt][0] + bb[u][0] + xv[u][rt][0], acc[ft][rt][1] + bb[u][1] + xv[u][rt][1], acc[ft][rt][2] + bb[u][2] + xv[u][rt][2],
                         acc[ft][rt][3] + bb[u][3] + xv[u][rt][3]};
        acc[ft][rt] = v;
        if constexpr (STORE) {
          if (FULL || lr < nrow) *reinterpret_cast<f32x4*>(xb + lr * D + f) = v;
        }
      }
    }
    if constexpr (FG < 4) asm volatile("" ::: "memory");   // the next group's loads stay behind this group's stores
  }
}
template <int RTT, bool STORE>
__device__ __forceinline__ void add_bias_residual(f32x4 (&acc)[4][RTT], const float* __restrict__ bias, float* __restrict__ xb, int nrow, int wave, int li, int g) {
  if (nrow == RTT * 16) add_bias_residual_t<RTT, STORE, true>(acc, bias, xb, nrow, wave, li, g);
  else add_bias_residual_t<RTT, STORE, false>(acc, bias, xb, nrow, wave, li, g);
}

// out[row][col0 + f] = T(acc + bias) (optionally ReLU): the q / q k v / logits stores.  ob = out + r0 * ld (first row of the workgroup)
template <int NFT, int RTT, typename T, bool RELU, bool FULL>
__device__ __forceinline__ void store_bias_t(const f32x4 (&acc)[NFT][RTT], const f32x4 (&bb)[NFT], T* __restrict__ ob, int ld, int nrow, int fwave, int flimit,
                                             int li_, int g_) {
  const int li = opaque(li_), g = opaque(g_);
#pragma unroll
  for (int ft = 0; ft < NFT; ++ft) {
    const int f = fwave + ft * 16 + g * 4;
    if (f < flimit) {   // wave-uniform per 16-feature tile up to the vocabulary edge (vocab % 4 == 0: a quad is live or dead as a whole)
#pragma unroll
      for (int rt = 0; rt < RTT; ++rt) {
        const int lr = rt * 16 + li;
        if (FULL || lr < nrow) {
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) { v[r] = acc[ft][rt][r] + bb[ft][r]; if (RELU) v[r] = fmaxf(v[r], 0.f); }
          if constexpr (sizeof(T) == 4) *reinterpret_cast<f32x4*>(ob + (int64_t)lr * ld + f) = f32x4{v[0], v[1], v[2], v[3]};
          else *reinterpret_cast<bf16x4*>(ob + (int64_t)lr * ld + f) = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
        }
      }
    }
  }
}
template <int NFT, int RTT, typename T, bool RELU>
__device__ __forceinline__ void store_bias(const f32x4 (&acc)[NFT][RTT], const f32x4 (&bb)[NFT], T* __restrict__ ob, int ld, int nrow, int fwave, int flimit, int li, int g) {
  if (nrow == RTT * 16) store_bias_t<NFT, RTT, T, RELU, true>(acc, bb, ob, ld, nrow, fwave, flimit, li, g);
  else store_bias_t<NFT, RTT, T, RELU, false>(acc, bb, ob, ld, nrow, fwave, flimit, li, g);
}

// The q | k | v tail of a chain: model/packing.py::_qkv_tail_rows orders the weight rows of every 64-feature group so that this lane's quads of
// tiles 0 / 1 are features 8 g .. 8 g + 7 of its rows and those of tiles 2 / 3 the same + 32 -- 16-byte stores (four lanes: 64 contiguous
// bytes of a row), half as many as store_bias_t issues.  A pass's stores share vmcnt with the weight ring, and the next pass's first takes
// wait until all but seven of the outstanding operations are done: the fewer stores, the sooner the stream resumes.
__device__ __forceinline__ void load_bias_perm(f32x4 (&bb)[4], const float* __restrict__ bias, int fwave, int g) {
#pragma unroll
  for (int ft = 0; ft < 4; ++ft) bb[ft] = *reinterpret_cast<const f32x4*>(bias + fwave + (ft >> 1) * 32 + g * 8 + (ft & 1) * 4);
}
template <int RTT, bool FULL>
__device__ __forceinline__ void store_bias_perm_t(const f32x4 (&acc)[4][RTT], const f32x4 (&bb)[4], bf16_t* __restrict__ ob, int ld, int nrow, int fwave, int li_, int g_) {
  const int li = opaque(li_), g = opaque(g_);
#pragma unroll
  for (int rt = 0; rt < RTT; ++rt) {
    const int lr = rt * 16 + li;
    if (FULL || lr < nrow) {
      bf16x8 lo, hi;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        lo[r] = (bf16_t)(acc[0][rt][r] + bb[0][r]);
        lo[4 + r] = (bf16_t)(acc[1][rt][r] + bb[1][r]);
        hi[r] = (bf16_t)(acc[2][rt][r] + bb[2][r]);
        hi[4 + r] = (bf16_t)(acc[3][rt][r] + bb[3][r]);
      }
      bf16_t* o = ob + (int64_t)lr * ld + fwave + g * 8;
      *reinterpret_cast<bf16x8*>(o) = lo;
      *reinterpret_cast<bf16x8*>(o + 32) = hi;
    }
  }
}
template <int RTT>
__device__ __forceinline__ void store_bias_perm(const f32x4 (&acc)[4][RTT], const f32x4 (&bb)[4], bf16_t* __restrict__ ob, int ld, int nrow, int fwave, int li, int g) {
  if (nrow == RTT * 16) store_bias_perm_t<RTT, true>(acc, bb, ob, ld, nrow, fwave, li, g);
  else store_bias_perm_t<RTT, false>(acc, bb, ob, ld, nrow, fwave, li, g);
}

// x[row][f] = acc: the residual stream back to memory
template <int RTT>
__device__ __forceinline__ void store_x(const f32x4 (&acc)[4][RTT], float* __restrict__ xb, int nrow, int wave, int li_, int g_) {
  const int li = opaque(li_), g = opaque(g_);
  auto body = [&](auto FULL) {
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) {
      const int f = wave * 64 + ft * 16 + g * 4;
#pragma unroll
      for (int rt = 0; rt < RTT; ++rt) {
        const int lr = rt * 16 + li;
        if (decltype(FULL)::value || lr < nrow) *reinterpret_cast<f32x4*>(xb + lr * D + f) = acc[ft][rt];
      }
    }
  };
  if (nrow == RTT * 16) body(std::true_type());
  else body(std::false_type());
}

struct RowsP {
  int R; float eps;
  const int32_t* d_pos;
  float* x;                        // [R, 512] fp32 residual stream (in place)
  const bf16_t* att;               // [R, 512] attention output feeding the first product
  const char* wstream;             // packed weight stream of this launch (model/packing.py)
  int64_t wave_stride;             // bytes between the streams of consecutive waves
  const float* out_b;              // bias of the attention out-projection
  const float *ln_g, *ln_b;        // mid: norm2; ffn: norm3
  // mid
  const float* qbias_tab;          // [Pmax, 512]: ca_q bias + position term
  bf16_t* q;                       // [R, 512]
  // ffn
  const float *ff1_b, *ff2_b;
  // embedding prologue
  const int32_t* seq; int seq_ld; const float *word_emb, *pos_tab, *emb_g, *emb_b;
  // tail
  const float *lnt_g, *lnt_b;      // next layer's norm1, or the decoder's final norm
  const float* bias_tab;           // [Pmax, 1536]: in_proj bias + position term of q and k
  bf16_t* qkv;                     // [R, 1536]
  const float *h0_b, *h1_b, *h2_b;
  float* logits; int vocab;        // [R, vocab] fp32
  unsigned long long* trace;       // development (omp_debug_swin_mlp_trace): [workgroup][16] cycle sums of wave 0 per phase of dec_rows_ffn_kernel
  int xcd_mask;                    // XCDs that run the launch's tiles (rows_common.inc xcd_tile); 0 = every block is a tile
};

// ---------------------------------------------------------------------------------------------------------------------
// x' = x + att Wo^T + bo;  q = bf16(LayerNorm2(x') Wq^T + qbias[pos])          stream: Wo (64 fragments per wave), Wq (64)
// ---------------------------------------------------------------------------------------------------------------------
template <int RTT>
__global__ __launch_bounds__(NW * 64) void dec_rows_mid_kernel(RowsP p) {
  constexpr int RT = RTT * 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* tile = smem;                                               // RT x A_PITCH
  float* red = reinterpret_cast<float*>(smem + RT * A_PITCH + TILE_SLACK);      // 2 x NW x RT
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile_id = xcd_tile(p.xcd_mask);
  if (tile_id < 0 || (int64_t)tile_id * RT >= p.R) return;   // a block of an XCD outside the mask, or beyond the last tile
  const int64_t r0 = (int64_t)tile_id * RT;
  Stream st = stream_of_wave(p.wstream, p.wave_stride, wave, lane);
  u32x4 ring[PF];
  sfor<PF>([&](auto U) { ws_issue<decltype(U)::value>(ring, st); });
  const int pos = *p.d_pos;

  stage_rows<RTT>(p.att, r0, p.R, tile, wave, lane);
  lds_barrier();
  const char* a_lane = tile + li * A_PITCH + g * 16;
  f32x4 acc[4][RTT];
  zero_acc(acc);
  gemm_pass<4, 16, RTT, A_PITCH>(acc, a_lane, ring, st);
  const int nrow = (int)((int64_t)p.R - r0 < RT ? (int64_t)p.R - r0 : RT);
  add_bias_residual<RTT, true>(acc, p.out_b, p.x + r0 * D, nrow, wave, li, g);
  ln_acc_to_tile<RTT>(acc, p.ln_g, p.ln_b, p.eps, tile, red, wave, li, g);
  zero_acc(acc);
  f32x4 qb[4];
  load_bias<4>(qb, p.qbias_tab + (int64_t)pos * D, wave * 64, D, g);
  gemm_pass<4, 16, RTT, A_PITCH>(acc, a_lane, ring, st);
  ws_drain(ring);
  store_bias<4, RTT, bf16_t, false>(acc, qb, p.q + r0 * D, D, nrow, wave * 64, D, li, g);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the ring's run-ahead requests (PF fragments of slack behind every stream)
}

// ---------------------------------------------------------------------------------------------------------------------
// PRO 0: x1 = x + att Wo^T + bo;  x' = x1 + relu(LayerNorm3(x1) W1^T + b1) W2^T + b2      stream: Wo (64), 8 x [W1 chunk (32), W2 chunk (32)]
// PRO 1: x' = LayerNorm(word[token] + position)                                            (layer 0: no stream)
// TAIL 0: qkv = bf16(LayerNorm1'(x') Win^T + bias_tab[pos])                                stream: 3 x 64
// TAIL 1: logits = h2(relu(h1(relu(h0(LayerNorm_f(x'))))))                                 stream: 64, 64, then ceil(vocab / 128) passes in 512 / 128 steps
// PRO 2 / TAIL 2 / ACT 1 (GELU): the same chains for the blocks of Swin-B's stage 2 (C = 512), see omp_swin_rows_block below:
// PRO 2: the tail's LayerNorm straight from the residual stream (the first block's norm1); TAIL 2: nothing behind the FFN (the last block)
// ---------------------------------------------------------------------------------------------------------------------
template <int RTT, int PRO, int TAIL, int ACT>
__global__ __launch_bounds__(NW * 64) void dec_rows_ffn_kernel(RowsP p) {
  constexpr int RT = RTT * 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* tile = smem;                                                   // RT x A_PITCH
  char* hbuf = smem + RT * A_PITCH + TILE_SLACK;                       // RT x H_PITCH
  float* red = reinterpret_cast<float*>(hbuf + RT * H_PITCH + TILE_SLACK);   // 2 x NW x RT
  float* b1s = red + 2 * NW * RT;                                      // d_ff floats (PRO 0)
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile_id = xcd_tile(p.xcd_mask);
  if (tile_id < 0 || (int64_t)tile_id * RT >= p.R) return;   // a block of an XCD outside the mask, or beyond the last tile
  const int64_t r0 = (int64_t)tile_id * RT;
  Stream st = stream_of_wave(p.wstream, p.wave_stride, wave, lane);
  u32x4 ring[PF];
  sfor<PF>([&](auto U) { ws_issue<decltype(U)::value>(ring, st); });
  const int pos = p.d_pos != nullptr ? *p.d_pos : 0;
  // development trace (p.trace): wave 0's cycles per phase -- 0 whole kernel, 1 prologue (rows staged / normalised), 2 out-projection product,
  // 3 residual + LayerNorm + bias, 4 linear1 products, 5 barrier before the hidden tile is rewritten, 6 activation + LDS writes, 7 barrier
  // behind them, 8 linear2 products, 9 x store, 10 the tail's LayerNorm, 11 the tail's products and stores
  const bool tracing = p.trace != nullptr && wave == 0;
  unsigned long long tr[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t_last = tracing ? __builtin_amdgcn_s_memtime() : 0ull;
  const unsigned long long t_begin = t_last;
  auto stamp = [&](auto SLOT) {
    if (tracing) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      tr[decltype(SLOT)::value] += t - t_last;
      t_last = t;
    }
  };
  typedef std::integral_constant<int, 1> S1; typedef std::integral_constant<int, 2> S2; typedef std::integral_constant<int, 3> S3;
  typedef std::integral_constant<int, 4> S4; typedef std::integral_constant<int, 5> S5; typedef std::integral_constant<int, 6> S6;
  typedef std::integral_constant<int, 7> S7; typedef std::integral_constant<int, 8> S8; typedef std::integral_constant<int, 9> S9;
  typedef std::integral_constant<int, 10> S10; typedef std::integral_constant<int, 11> S11;
  const char* a_lane = tile + li * A_PITCH + g * 16;
  const int nrow = (int)((int64_t)p.R - r0 < RT ? (int64_t)p.R - r0 : RT);
  f32x4 acc[4][RTT];

  if constexpr (PRO == 0) {
    static_assert(4 * D / 4 == NW * 64, "linear1's bias: one 16-byte piece per thread");
    const f32x4 b1v = reinterpret_cast<const f32x4*>(p.ff1_b)[tid];   // requested before the rows: one wait covers both
    stage_rows<RTT>(p.att, r0, p.R, tile, wave, lane);
    reinterpret_cast<f32x4*>(b1s)[tid] = b1v;
    lds_barrier();
    stamp(S1());
    zero_acc(acc);
    gemm_pass<4, 16, RTT, A_PITCH>(acc, a_lane, ring, st);
    stamp(S2());
    add_bias_residual<RTT, false>(acc, p.out_b, p.x + r0 * D, nrow, wave, li, g);
    ln_acc_to_tile<RTT>(acc, p.ln_g, p.ln_b, p.eps, tile, red, wave, li, g);
    // linear2's accumulators start as x1 + b2
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) {
      const f32x4 bb = *reinterpret_cast<const f32x4*>(p.ff2_b + wave * 64 + ft * 16 + g * 4);
#pragma unroll
      for (int rt = 0; rt < RTT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[ft][rt][r] += bb[r];
    }
    stamp(S3());
#pragma unroll 1
    for (int c = 0; c < 4 * D / HC; ++c) {
      f32x4 a1[2][RTT];
      zero_acc(a1);
      gemm_pass<2, 16, RTT, A_PITCH>(a1, a_lane, ring, st);   // hidden units c * 256 + 32 w + 16 t + 4 g + r of the rows
      stamp(S4());
      lds_barrier();   // everybody has left linear2 of chunk c - 1: the hidden tile may be overwritten
      stamp(S5());
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const f32x4 bb = *reinterpret_cast<const f32x4*>(b1s + c * HC + wave * 32 + t * 16 + g * 4);
#pragma unroll
        for (int rt = 0; rt < RTT; ++rt) {
          float hv[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) hv[r] = a1[t][rt][r] + bb[r];
          if constexpr (ACT == 1) {
            gelu_fast_n<4>(hv);   // the bf16 engine's GELU on every epilogue path (common.h): a value does not depend on the kernel that produced it
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) hv[r] = fmaxf(hv[r], 0.f);
          }
          *reinterpret_cast<bf16x4*>(hbuf + (rt * 16 + li) * H_PITCH + (wave * 32 + t * 16 + g * 4) * 2) = bf16x4{(bf16_t)hv[0], (bf16_t)hv[1], (bf16_t)hv[2], (bf16_t)hv[3]};
        }
      }
      stamp(S6());
      lds_barrier();   // chunk c is complete in LDS
      stamp(S7());
      gemm_pass<4, HC / 32, RTT, H_PITCH>(acc, hbuf + li * H_PITCH + g * 16, ring, st);
      stamp(S8());
    }
    if constexpr (TAIL == 2) ws_drain(ring);   // no product follows: the ring's run-ahead requests must land before its registers are reused
    // acc = x' : back to memory (the next attention sub-layer's residual), then the tail consumes it from registers
    store_x<RTT>(acc, p.x + r0 * D, nrow, wave, li, g);
    stamp(S9());
    if constexpr (TAIL != 2) ln_acc_to_tile<RTT>(acc, p.lnt_g, p.lnt_b, p.eps, tile, red, wave, li, g);
    stamp(S10());
  } else if constexpr (PRO == 1) {
    // embedding + LayerNorm -> x (fp32), then the tail's LayerNorm -> operand tile: a wave per row, RT / NW rows per wave with every
    // row's loads issued before the first row's arithmetic (one memory round trip for the lot, not one per row)
    constexpr int RPW = RT / NW;
    static_assert(RT % NW == 0, "rows per wave");
    f32x4 ev[RPW][4];
    bool live[RPW];
    int64_t rr[RPW];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      int64_t r = r0 + wave * RPW + i;
      live[i] = r < p.R;
      if (!live[i]) r = p.R - 1;
      rr[i] = r;
      const int tok = p.seq[r * p.seq_ld + pos];
      const float* we = p.word_emb + (int64_t)tok * D + lane * 8;
      const float* pe = p.pos_tab + (int64_t)pos * D + lane * 8;
      ev[i][0] = *reinterpret_cast<const f32x4*>(we); ev[i][1] = *reinterpret_cast<const f32x4*>(we + 4);
      ev[i][2] = *reinterpret_cast<const f32x4*>(pe); ev[i][3] = *reinterpret_cast<const f32x4*>(pe + 4);
    }
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[j] = ev[i][0][j] + ev[i][2][j]; v[j + 4] = ev[i][1][j] + ev[i][3][j]; }
      ln_row512(v, p.emb_g, p.emb_b, lane, p.eps);
      if (live[i]) {
        *reinterpret_cast<f32x4*>(p.x + rr[i] * D + lane * 8) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(p.x + rr[i] * D + lane * 8 + 4) = f32x4{v[4], v[5], v[6], v[7]};
      }
      ln_row512(v, p.lnt_g, p.lnt_b, lane, p.eps);
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (bf16_t)v[j];
      *reinterpret_cast<bf16x8*>(tile + (wave * RPW + i) * A_PITCH + lane * 16) = o;
    }
    lds_barrier();
  } else {
    // PRO 2: the rows of the fp32 residual stream as they are -> the tail's LayerNorm -> operand tile (a wave per row, every row's loads first)
    constexpr int RPW = RT / NW;
    static_assert(RT % NW == 0, "rows per wave");
    const float* xb = p.x + r0 * D;
    f32x4 xv[RPW][2];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      int lr = wave * RPW + i;
      if (lr > nrow - 1) lr = nrow - 1;
      xv[i][0] = *reinterpret_cast<const f32x4*>(xb + lr * D + lane * 8);
      xv[i][1] = *reinterpret_cast<const f32x4*>(xb + lr * D + lane * 8 + 4);
    }
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      float v[8] = {xv[i][0][0], xv[i][0][1], xv[i][0][2], xv[i][0][3], xv[i][1][0], xv[i][1][1], xv[i][1][2], xv[i][1][3]};
      ln_row512(v, p.lnt_g, p.lnt_b, lane, p.eps);
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (bf16_t)v[j];
      *reinterpret_cast<bf16x8*>(tile + (wave * RPW + i) * A_PITCH + lane * 16) = o;
    }
    lds_barrier();
  }

  if constexpr (TAIL == 0) {
    auto qkv_pass = [&](int ps, auto LAST) {
      zero_acc(acc);
      f32x4 bb[4];
      load_bias_perm(bb, p.bias_tab + (int64_t)pos * (3 * D) + ps * D, wave * 64, g);
      gemm_pass<4, 16, RTT, A_PITCH>(acc, a_lane, ring, st);
      if constexpr (decltype(LAST)::value) ws_drain(ring);
      store_bias_perm<RTT>(acc, bb, p.qkv + r0 * (3 * D) + ps * D, 3 * D, nrow, wave * 64, li, g);
    };
#pragma unroll 1
    for (int ps = 0; ps < 2; ++ps) qkv_pass(ps, std::false_type());
    qkv_pass(2, std::true_type());
  } else if constexpr (TAIL == 1) {
    // prediction head (block/mlp.py:11-13): two hidden layers with ReLU through the operand tile, then the vocabulary projection
#pragma unroll 1
    for (int hl = 0; hl < 2; ++hl) {
      zero_acc(acc);
      gemm_pass<4, 16, RTT, A_PITCH>(acc, a_lane, ring, st);
      lds_barrier();   // every wave has read the tile: it may be overwritten
      const float* hb_ = hl == 0 ? p.h0_b : p.h1_b;
#pragma unroll
      for (int ft = 0; ft < 4; ++ft) {
        const int f = wave * 64 + ft * 16 + g * 4;
        const f32x4 bb = *reinterpret_cast<const f32x4*>(hb_ + f);
#pragma unroll
        for (int rt = 0; rt < RTT; ++rt) {
          bf16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (bf16_t)fmaxf(acc[ft][rt][r] + bb[r], 0.f);
          *reinterpret_cast<bf16x4*>(tile + (rt * 16 + li) * A_PITCH + f * 2) = o;
        }
      }
      lds_barrier();
    }
    const int V = p.vocab, vpad = (V + 127) / 128 * 128;
    const int n512 = vpad / 512, n128 = (vpad - n512 * 512) / 128;
#pragma unroll 1
    for (int ps = 0; ps < n512; ++ps) {
      zero_acc(acc);
      f32x4 bb[4];
      load_bias<4>(bb, p.h2_b, ps * 512 + wave * 64, V, g);
      gemm_pass<4, 16, RTT, A_PITCH>(acc, a_lane, ring, st);
      ws_drain(ring);   // (after EVERY vocabulary pass: which one is the last depends on the vocabulary; once per step, ~0.5 us each)
      store_bias<4, RTT, float, false>(acc, bb, p.logits + r0 * V, V, nrow, ps * 512 + wave * 64, V, li, g);
    }
#pragma unroll 1
    for (int ps = 0; ps < n128; ++ps) {
      f32x4 a1[1][RTT];
      zero_acc(a1);
      f32x4 bb[1];
      load_bias<1>(bb, p.h2_b, n512 * 512 + ps * 128 + wave * 16, V, g);
      gemm_pass<1, 16, RTT, A_PITCH>(a1, a_lane, ring, st);
      ws_drain(ring);
      store_bias<1, RTT, float, false>(a1, bb, p.logits + r0 * V, V, nrow, n512 * 512 + ps * 128 + wave * 16, V, li, g);
    }
  }
  stamp(S11());
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (tracing && lane == 0) {
    tr[0] = __builtin_amdgcn_s_memtime() - t_begin;
#pragma unroll
    for (int i = 0; i < 12; ++i) p.trace[(int64_t)tile_id * 16 + i] = tr[i];
  }
}

constexpr int RTT_DEFAULT = 5;   // 80 rows per workgroup: the Swin chains (thousands of workgroups per launch)

// Rows per workgroup of a DECODER launch.  A workgroup streams the whole weight set whatever its rows (6.3 MB for the FFN chain: about
// 57 us at the 110 GB/s a CU draws from L2) and runs 16 x RTT rows x 3.15 M parameters on its CU's matrix cores (10-15 us per 16 rows), so a
// launch that leaves CUs idle is better cut finer -- but the polygon and the recognition decoder run side by side on two streams, a chain
// workgroup takes its CU's LDS whole, and what one decoder does not occupy is where the other's HBM-bound attention kernels run.  Measured at
// 10 240 rows (160 images x 64 instances; profiles/r05o_kbench_dec_rows_tiles.txt): the FFN chain alone 110 / 91 / 84 / 147 us at 80 / 64 /
// 48 / 32 rows per workgroup (128 / 160 / 214 / 320 workgroups), the whole phase on ONE stream 125.6 / 122.2 / 119.5 / 137.7 ms, on the
// engine's TWO streams 101.2 / 102.9 / 106.0 / 115.9 ms.  So: the smallest tile of {32, 48, 64, 80} rows that keeps a launch on half the chip
// (10 240 rows: 80; 5 120 rows: 48).  omp_debug_rows_tile forces a tile (A/B, tests).
int rows_rtt(int R, int lo = 2) {   // lo: the smallest tile the kernel is instantiated for (the mid chain also runs 16 rows: few-row phases)
  const int forced = omp_cur().rows_rtt;
  if (forced >= 2 && forced <= 5) return forced;
  const int half = omp_device_cus() / 2;   // 128 on MI355X
  for (int rtt = lo; rtt < 5; ++rtt)
    if (((int64_t)R + 16 * rtt - 1) / (16 * rtt) <= half) return rtt;
  return 5;
}

// grid of a chain launch of n_tiles tiles under an XCD mask (rows_common.inc xcd_tile): the mask applies when the tiles fit the masked XCDs' CUs
// in ONE round (32 CUs per XCD), else every block is a tile.  -> blocks; *mask_out = the mask the kernel gets
unsigned xcd_grid(int64_t n_tiles, int mask, int* mask_out) {
  mask &= 0xFF;
  const int n_sel = __builtin_popcount((unsigned)mask);
  if (mask == 0 || mask == 0xFF || n_tiles > (int64_t)n_sel * (omp_device_cus() / 8)) { *mask_out = 0; return (unsigned)n_tiles; }
  *mask_out = mask;
  return (unsigned)((n_tiles + n_sel - 1) / n_sel * 8);
}

template <typename K>
int raise_lds(K kern, const char* what) {
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
    omp_set_error("%s: cannot raise dynamic LDS limit", what);
    return OMP_ERR_LAUNCH;
  }
  return OMP_OK;
}

template <int RTT, int PRO, int TAIL, int ACT>
int launch_ffn_t(RowsP p, hipStream_t st) {
  constexpr int RT = RTT * 16;
  const size_t smem = (size_t)RT * A_PITCH + TILE_SLACK + RT * H_PITCH + TILE_SLACK + 2 * NW * RT * 4 + 4 * D * 4;
  auto kern = dec_rows_ffn_kernel<RTT, PRO, TAIL, ACT>;
  static bool done = false;   // per instantiation
  if (!done) {
    const int rc = raise_lds(kern, "omp_dec_rows_ffn");
    if (rc != OMP_OK) return rc;
    done = true;
  }
  const unsigned grid = xcd_grid(((int64_t)p.R + RT - 1) / RT, p.xcd_mask, &p.xcd_mask);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), smem, st, p);
  OMP_CHECK_LAUNCH("omp_dec_rows_ffn");
  return OMP_OK;
}

// the Swin chains (ACT 1): always 80 rows; the decoder chains (ACT 0): rows_rtt
template <int PRO, int TAIL, int ACT = 0>
int launch_ffn(const RowsP& p, hipStream_t st) {
  if constexpr (ACT == 0) {
    switch (rows_rtt(p.R)) {
      case 2: return launch_ffn_t<2, PRO, TAIL, ACT>(p, st);
      case 3: return launch_ffn_t<3, PRO, TAIL, ACT>(p, st);
      case 4: return launch_ffn_t<4, PRO, TAIL, ACT>(p, st);
      default: break;
    }
  }
  return launch_ffn_t<RTT_DEFAULT, PRO, TAIL, ACT>(p, st);
}

template <int RTT>
int launch_mid_t(RowsP p, hipStream_t st) {
  constexpr int RT = RTT * 16;
  const size_t smem = (size_t)RT * A_PITCH + TILE_SLACK + 2 * NW * RT * 4;
  auto kern = dec_rows_mid_kernel<RTT>;
  static bool done = false;
  if (!done) {
    const int rc = raise_lds(kern, "omp_dec_rows_mid");
    if (rc != OMP_OK) return rc;
    done = true;
  }
  const unsigned grid = xcd_grid(((int64_t)p.R + RT - 1) / RT, p.xcd_mask, &p.xcd_mask);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), smem, st, p);
  return OMP_OK;
}

}  // namespace

int omp_rows_tile() { return RTT_DEFAULT * 16; }

extern "C" int omp_dec_rows_mid(const omp_dec_rows_args* a, omp_stream_t s) {
  OMP_CHECK_ARG(a != nullptr, "omp_dec_rows_mid: null argument block");
  OMP_CHECK_ARG(a->R > 0 && a->d_pos && a->x && a->att && a->wstream && a->out_b && a->ln_g && a->ln_b && a->qbias_tab && a->q, "omp_dec_rows_mid: null pointer");
  const int mul = a->x3 ? 2 : 1;
  OMP_CHECK_ARG(a->wave_stride == (int64_t)mul * 128 * 1024 && ((uintptr_t)a->wstream % 16) == 0,
                "omp_dec_rows_mid: a wave's stream holds %d fragments of 1 KB (wave_stride %lld: not what model/packing.py::pack_rows_mid returns)", mul * 128, (long long)a->wave_stride);
  if (a->x3) {
    const int slot3 = omp_prof_active(OMP_PROF_ROWS) ? omp_prof_begin(OMP_PROF_ROWS, (hipStream_t)s, 3 * 4.0 * (double)a->R * D * D, (double)a->R * D * (4 + 4 + 4 + 4) + 2.0 * D * D * 4) : -1;
    const int rc3 = omp_rows_x3_mid(a, (hipStream_t)s);
    if (slot3 >= 0) omp_prof_end(OMP_PROF_ROWS, slot3, (hipStream_t)s);
    return rc3;
  }
  RowsP p{};
  p.R = a->R; p.eps = a->eps; p.d_pos = a->d_pos; p.x = a->x; p.att = reinterpret_cast<const bf16_t*>(a->att);
  p.wstream = reinterpret_cast<const char*>(a->wstream); p.wave_stride = a->wave_stride;
  p.out_b = a->out_b; p.ln_g = a->ln_g; p.ln_b = a->ln_b; p.qbias_tab = a->qbias_tab; p.q = reinterpret_cast<bf16_t*>(a->q);
  p.xcd_mask = a->xcd_mask;
  const int slot = omp_prof_active(OMP_PROF_ROWS) ? omp_prof_begin(OMP_PROF_ROWS, (hipStream_t)s, 4.0 * (double)a->R * D * D, (double)a->R * D * (2 + 4 + 4 + 2) + 2.0 * D * D * 2) : -1;
  int rc;
  switch (rows_rtt(p.R, 1)) {
    case 1: rc = launch_mid_t<1>(p, (hipStream_t)s); break;
    case 2: rc = launch_mid_t<2>(p, (hipStream_t)s); break;
    case 3: rc = launch_mid_t<3>(p, (hipStream_t)s); break;
    case 4: rc = launch_mid_t<4>(p, (hipStream_t)s); break;
    default: rc = launch_mid_t<5>(p, (hipStream_t)s); break;
  }
  if (slot >= 0) omp_prof_end(OMP_PROF_ROWS, slot, (hipStream_t)s);
  if (rc != OMP_OK) return rc;
  OMP_CHECK_LAUNCH("omp_dec_rows_mid");
  return OMP_OK;
}

extern "C" int omp_dec_rows_ffn(const omp_dec_rows_args* a, omp_stream_t s) {
  OMP_CHECK_ARG(a != nullptr, "omp_dec_rows_ffn: null argument block");
  OMP_CHECK_ARG(a->R > 0 && a->d_pos && a->x && a->wstream && a->lnt_g && a->lnt_b, "omp_dec_rows_ffn: null pointer");
  OMP_CHECK_ARG(a->prologue == 0 || a->prologue == 1, "omp_dec_rows_ffn: prologue must be 0 (attention out-projection + FFN) or 1 (embedding)");
  OMP_CHECK_ARG(a->tail == 0 || a->tail == 1, "omp_dec_rows_ffn: tail must be 0 (next layer's q | k | v) or 1 (prediction head)");
  if (a->prologue == 0) OMP_CHECK_ARG(a->att && a->out_b && a->ln_g && a->ln_b && a->ff1_b && a->ff2_b, "omp_dec_rows_ffn: null pointer (attention / FFN part)");
  else OMP_CHECK_ARG(a->seq && a->word_emb && a->pos_tab && a->emb_g && a->emb_b && a->seq_ld > 0, "omp_dec_rows_ffn: null pointer (embedding part)");
  if (a->tail == 0) OMP_CHECK_ARG(a->bias_tab && a->qkv, "omp_dec_rows_ffn: null pointer (q | k | v tail)");
  else OMP_CHECK_ARG(a->h0_b && a->h1_b && a->h2_b && a->logits && a->vocab > 0 && a->vocab % 4 == 0, "omp_dec_rows_ffn: prediction-head tail needs its biases, logits and vocab %% 4 == 0");
  const int vpad = (a->vocab + 127) / 128 * 128;
  const int64_t frags = (a->prologue == 0 ? 64 + 16 * 32 : 0) + (a->tail == 0 ? 192 : 128 + (vpad / 512) * 64 + ((vpad % 512) / 128) * 16);
  const int mul = a->x3 ? 2 : 1;
  OMP_CHECK_ARG(a->wave_stride == mul * frags * 1024 && ((uintptr_t)a->wstream % 16) == 0,
                "omp_dec_rows_ffn: a wave's stream holds %lld fragments of 1 KB here (wave_stride %lld: not what the packer of this chain returns)", (long long)(mul * frags), (long long)a->wave_stride);
  if (a->x3) {
    const double fl3 = 3 * 2.0 * (double)a->R * D * ((a->prologue == 0 ? D + 8.0 * D : 0.0) + (a->tail == 0 ? 3.0 * D : 2.0 * D + a->vocab));
    const int slot3 = omp_prof_active(OMP_PROF_ROWS) ? omp_prof_begin(OMP_PROF_ROWS, (hipStream_t)s, fl3, (double)a->R * D * (4 + 4 + 4) + (double)a->R * (a->tail == 0 ? 3 * D * 4 : a->vocab * 4) + (double)frags * 2 * 8192) : -1;
    const int rc3 = omp_rows_x3_ffn(a, (hipStream_t)s);
    if (slot3 >= 0) omp_prof_end(OMP_PROF_ROWS, slot3, (hipStream_t)s);
    return rc3;
  }
  RowsP p{};
  p.R = a->R; p.eps = a->eps; p.d_pos = a->d_pos; p.x = a->x; p.att = reinterpret_cast<const bf16_t*>(a->att);
  p.wstream = reinterpret_cast<const char*>(a->wstream); p.wave_stride = a->wave_stride;
  p.out_b = a->out_b; p.ln_g = a->ln_g; p.ln_b = a->ln_b; p.ff1_b = a->ff1_b; p.ff2_b = a->ff2_b;
  p.seq = a->seq; p.seq_ld = a->seq_ld; p.word_emb = a->word_emb; p.pos_tab = a->pos_tab; p.emb_g = a->emb_g; p.emb_b = a->emb_b;
  p.lnt_g = a->lnt_g; p.lnt_b = a->lnt_b; p.bias_tab = a->bias_tab; p.qkv = reinterpret_cast<bf16_t*>(a->qkv);
  p.h0_b = a->h0_b; p.h1_b = a->h1_b; p.h2_b = a->h2_b; p.logits = a->logits; p.vocab = a->vocab;
  p.xcd_mask = a->xcd_mask;
  hipStream_t st = (hipStream_t)s;
  const double fl = 2.0 * (double)a->R * D * ((a->prologue == 0 ? D + 8.0 * D : 0.0) + (a->tail == 0 ? 3.0 * D : 2.0 * D + a->vocab));
  const int slot = omp_prof_active(OMP_PROF_ROWS) ? omp_prof_begin(OMP_PROF_ROWS, st, fl, (double)a->R * D * (2 + 4 + 4) + (double)a->R * (a->tail == 0 ? 3 * D * 2 : a->vocab * 4) + (double)frags * 8192) : -1;
  int rc;
  if (a->prologue == 0) rc = a->tail == 0 ? launch_ffn<0, 0>(p, st) : launch_ffn<0, 1>(p, st);
  else rc = a->tail == 0 ? launch_ffn<1, 0>(p, st) : launch_ffn<1, 1>(p, st);
  if (slot >= 0) omp_prof_end(OMP_PROF_ROWS, slot, st);
  return rc;
}

extern "C" int omp_dec_rows_tile(void) { return omp_rows_tile(); }

extern "C" int omp_debug_rows_tile_choice(int R, int mid) {   // host logic: rows per workgroup a decoder chain launch of R rows would take (mid: omp_dec_rows_mid)
  if (R <= 0) return OMP_ERR_INVALID;
  return 16 * rows_rtt(R, mid ? 1 : 2);
}

extern "C" int omp_debug_rows_tile(int rtt) {   // 0 = by row count (rows_rtt), 2..5 = 16 x rtt rows per workgroup of every decoder chain launch
  OMP_CHECK_ARG(rtt == 0 || (rtt >= 2 && rtt <= 5), "omp_debug_rows_tile: 0 (automatic) or 2..5 tiles of 16 rows (got %d)", rtt);
  omp_cur().rows_rtt = rtt;
  return OMP_OK;
}


// ---------------------------------------------------------------------------------------------------------------------
// Swin-B stage 2 (C = 512, 16 heads: 18 of the 24 blocks, 60 % of the encoder) on the same chains.  Everything of a block except the
// window attention core is row-local:
//     x = x + proj(attn(norm1(x)));  x = x + fc2(GELU(fc1(norm2(x))))                  swin_transformer.py:196-253
// so a block is TWO launches -- the window attention kernel on q | k | v, then ONE chain: proj + residual, norm2, fc1 + GELU, fc2 + residual,
// and already the NEXT block's norm1 + qkv Linear -- instead of seven (LayerNorm, qkv, attention, proj, LayerNorm, fc1, fc2); the
// normalised rows, the hidden activations and the residual stream between the sub-layers never reach HBM.
// ---------------------------------------------------------------------------------------------------------------------
extern "C" int omp_swin_rows_block(const omp_swin_rows_args* a, omp_stream_t s) {
  OMP_CHECK_ARG(a != nullptr, "omp_swin_rows_block: null argument block");
  OMP_CHECK_ARG(a->M > 0 && a->M < (1ll << 31) && a->x && a->wstream, "omp_swin_rows_block: bad M / null pointer");
  OMP_CHECK_ARG(a->mode == 0 || a->mode == 1, "omp_swin_rows_block: mode 0 (norm1 + qkv) or 1 (proj + MLP [+ next norm1 + qkv])");
  const bool tail_qkv = a->n1_g != nullptr;
  if (a->mode == 0) OMP_CHECK_ARG(tail_qkv && a->n1_b && a->qkv_b && a->qkv, "omp_swin_rows_block: mode 0 needs norm1, the qkv bias and the qkv destination");
  else OMP_CHECK_ARG(a->att && a->proj_b && a->n2_g && a->n2_b && a->fc1_b && a->fc2_b && (!tail_qkv || (a->n1_b && a->qkv_b && a->qkv)), "omp_swin_rows_block: null pointer");
  const int64_t frags = (a->mode == 1 ? 64 + 8 * 64 : 0) + (tail_qkv ? 192 : 0);
  const int mul = a->x3 ? 2 : 1;
  OMP_CHECK_ARG(a->wave_stride == mul * frags * 1024 && ((uintptr_t)a->wstream % 16) == 0 && ((uintptr_t)a->x % 16) == 0,
                "omp_swin_rows_block: a wave's stream holds %lld fragments of 1 KB here (wave_stride %lld: not what the packer returns); 16-byte aligned pointers", (long long)(mul * frags), (long long)a->wave_stride);
  if (a->x3) {
    const double fl3 = 3 * 2.0 * (double)a->M * D * ((a->mode == 1 ? D + 8.0 * D : 0.0) + (tail_qkv ? 3.0 * D : 0.0));
    const double by3 = (double)a->M * D * (a->mode == 1 ? 4 + 4 + 4 : 4) + (tail_qkv ? (double)a->M * 3 * D * 4 : 0.0) + (double)frags * 2 * 8192;
    const int slot3 = omp_prof_active(OMP_PROF_MLP) ? omp_prof_begin(OMP_PROF_MLP, (hipStream_t)s, fl3, by3) : -1;
    const int rc3 = omp_rows_x3_swin(a, (hipStream_t)s);
    if (slot3 >= 0) omp_prof_end(OMP_PROF_MLP, slot3, (hipStream_t)s);
    return rc3;
  }
  RowsP p{};
  p.R = (int)a->M; p.eps = a->eps; p.d_pos = nullptr; p.x = a->x; p.att = reinterpret_cast<const bf16_t*>(a->att);
  p.wstream = reinterpret_cast<const char*>(a->wstream); p.wave_stride = a->wave_stride;
  p.out_b = a->proj_b; p.ln_g = a->n2_g; p.ln_b = a->n2_b; p.ff1_b = a->fc1_b; p.ff2_b = a->fc2_b;
  p.lnt_g = a->n1_g; p.lnt_b = a->n1_b; p.bias_tab = a->qkv_b; p.qkv = reinterpret_cast<bf16_t*>(a->qkv);
  p.trace = omp_cur().mlp_trace;   // omp_debug_swin_mlp_trace: the development buffer also takes this kernel's phase sums ([workgroup][16])
  hipStream_t st = (hipStream_t)s;
  const double fl = 2.0 * (double)a->M * D * ((a->mode == 1 ? D + 8.0 * D : 0.0) + (tail_qkv ? 3.0 * D : 0.0));
  const double by = (double)a->M * D * (a->mode == 1 ? 2 + 4 + 4 : 4) + (tail_qkv ? (double)a->M * 3 * D * 2 : 0.0) + (double)frags * 8192;
  const int slot = omp_prof_active(OMP_PROF_MLP) ? omp_prof_begin(OMP_PROF_MLP, st, fl, by) : -1;
  int rc;
  if (a->mode == 0) rc = launch_ffn<2, 0, 1>(p, st);
  else rc = tail_qkv ? launch_ffn<0, 0, 1>(p, st) : launch_ffn<0, 2, 1>(p, st);
  if (slot >= 0) omp_prof_end(OMP_PROF_MLP, slot, st);
  return rc;
}
