// Row-owner chains (csrc/dec_rows.hip) for the PARITY engine (bf16x3): fp32-grade products on the bf16 matrix cores.
//
// The parity engine stores fp32 and runs every large product as three bf16 matrix-core products of split operands (x = hi + lo with
// hi = bf16(x), lo = bf16(x - hi); x.w ~ hi.w_hi + lo.w_hi + hi.w_lo, the dropped lo.w_lo term is 2^-16 relative; include/omp355.h,
// omp_gemm_args.a_wrap).  Until round 5 its many-row decoder phases and its Swin stage-2 blocks ran one tiled GEMM launch per Linear over
// K' = 3 K, with the LayerNorms and the split conversions as launches of their own.  Here they are the same row-owner chains as the bf16
// engine's, with every operand as TWO planes:
//   * the resident row tile is [hi tile | lo tile] (bf16, the bytes of the fp32 rows): 48 rows per workgroup (two 50 KB tiles + the hidden
//     chunk's two planes fit 160 KB of LDS; 80 rows do not);
//   * the weight stream carries, per (k-step, feature tile), the fragment of w_hi then the fragment of w_lo (model/packing.py::pack_rows_*
//     with x3=True); ring of 16 fragments = 8 pairs in flight per wave;
//   * per pair and row tile three matrix-core instructions: w_hi a_hi + w_hi a_lo + w_lo a_hi, fp32 accumulation;
//   * LayerNorm outputs and hidden activations are split into the two planes where the launch-per-Linear path split them; q / q k v / logits and
//     the residual stream are fp32.
// A workgroup streams twice the bytes of the bf16 chain for 0.6 of the rows: the chain is paced by the weight stream (110 GB/s per compute
// unit, tools/probe_stream.hip), not by the matrix cores -- and still replaces ~12 launches per decoder layer at a third of their time.
#include <type_traits>
#include <utility>

#include "common.h"

namespace {

#include "rows_common.inc"

constexpr int PF3 = 16;                // fragments in flight per wave = 8 (w_hi, w_lo) pairs
constexpr int RTT3 = 3;                // 48 rows per workgroup
constexpr int RT3 = RTT3 * 16;
constexpr int HC3 = 128;               // hidden units per FFN chunk
constexpr int A_PITCH = D * 2 + 32;    // row pitch of one plane of the operand tile (conflict-free b128 fragment reads)
constexpr int H_PITCH = HC3 * 2 + 32;  // one plane of the hidden chunk tile: 72 dwords = 8 mod 64
constexpr int TILE_SLACK = 64;
constexpr int TILE_BYTES = RT3 * A_PITCH + TILE_SLACK;   // one plane
constexpr int HT_BYTES = RT3 * H_PITCH + TILE_SLACK;

// acc[ft][rt] += (w_hi + w_lo)[feature tile][:] . (a_hi + a_lo)[row tile][:] without the lo.lo term, over KS k-steps of 32; the wave's next
// 2 * NFT * KS stream fragments, ordered (k-step, feature tile, plane).  ah / al = the two planes of the operand tile at this lane's
// (row, k-chunk).  Between two takes one issue: exactly PF3 requests are outstanding at every take.
template <int NFT, int KS, int PITCH>
__device__ __forceinline__ void gemm_pass_x3(f32x4 (&acc)[NFT][RTT3], const char* ah, const char* al, u32x4 (&ring)[PF3], Stream& st) {
  constexpr int PAIRS = PF3 / 2, NG = NFT * KS / PAIRS, KPG = PAIRS / NFT;
  static_assert(PAIRS % NFT == 0 && (NFT * KS) % PAIRS == 0 && KPG % 2 == 0, "a pass is a whole number of ring revolutions, an even number of k-steps each");
  bf16x8 bh[2][RTT3], bl[2][RTT3];
#pragma unroll
  for (int rt = 0; rt < RTT3; ++rt) {
    bh[0][rt] = *reinterpret_cast<const bf16x8*>(ah + rt * 16 * PITCH);
    bl[0][rt] = *reinterpret_cast<const bf16x8*>(al + rt * 16 * PITCH);
  }
  auto group = [&](int gi) {
    sfor<PAIRS>([&](auto U) {
      constexpr int u = decltype(U)::value, ft = u % NFT, kk = u / NFT;
      if constexpr (ft == 0) {   // the next k-step's row fragments, requested before this k-step's matrix-core instructions
        const int off = (gi * KPG + kk + 1) * 64;
#pragma unroll
        for (int rt = 0; rt < RTT3; ++rt) {
          bh[(kk + 1) & 1][rt] = *reinterpret_cast<const bf16x8*>(ah + off + rt * 16 * PITCH);
          bl[(kk + 1) & 1][rt] = *reinterpret_cast<const bf16x8*>(al + off + rt * 16 * PITCH);
        }
      }
      const bf16x8 wh = ws_take<2 * u>(ring);
#pragma unroll
      for (int rt = 0; rt < RTT3; ++rt) {
        acc[ft][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, bh[kk & 1][rt], acc[ft][rt], 0, 0, 0);
        acc[ft][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, bl[kk & 1][rt], acc[ft][rt], 0, 0, 0);
      }
      ws_issue<2 * u>(ring, st);
      const bf16x8 wl = ws_take<2 * u + 1>(ring);
#pragma unroll
      for (int rt = 0; rt < RTT3; ++rt) acc[ft][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, bh[kk & 1][rt], acc[ft][rt], 0, 0, 0);
      ws_issue<2 * u + 1>(ring, st);
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

__device__ __forceinline__ void split4(const float* v, bf16x4& hi, bf16x4& lo) {
#pragma unroll
  for (int r = 0; r < 4; ++r) { hi[r] = (bf16_t)v[r]; lo[r] = (bf16_t)(v[r] - (float)hi[r]); }
}

// LayerNorm of the rows held in accumulator layout -> the two planes of the operand tile (see csrc/dec_rows.hip::ln_acc_to_tile)
__device__ __forceinline__ void ln_acc_to_tiles(const f32x4 (&v)[4][RTT3], const float* __restrict__ gam, const float* __restrict__ bet, float eps,
                                                char* tile_hi, char* tile_lo, float* red, int wave, int li_, int g_) {
  const int li = opaque(li_), g = opaque(g_);
  float* redl = red + li;
  float* red2l = red + NW * RT3 + li;
  float mean[RTT3], rstd[RTT3];
#pragma unroll
  for (int rt = 0; rt < RTT3; ++rt) {
    float s = 0.f;
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) s += (v[ft][rt][0] + v[ft][rt][1]) + (v[ft][rt][2] + v[ft][rt][3]);
    s = quad_group_sum(s);
    if (g == 0) redl[wave * RT3 + rt * 16] = s;
  }
  lds_barrier();
#pragma unroll
  for (int rt = 0; rt < RTT3; ++rt) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += redl[w * RT3 + rt * 16];
    mean[rt] = s * (1.0f / D);
  }
#pragma unroll
  for (int rt = 0; rt < RTT3; ++rt) {
    float q = 0.f;
#pragma unroll
    for (int ft = 0; ft < 4; ++ft)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float d = v[ft][rt][r] - mean[rt]; q += d * d; }
    q = quad_group_sum(q);
    if (g == 0) red2l[wave * RT3 + rt * 16] = q;
  }
  lds_barrier();
#pragma unroll
  for (int rt = 0; rt < RTT3; ++rt) {
    float q = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) q += red2l[w * RT3 + rt * 16];
    rstd[rt] = 1.0f / sqrtf(q * (1.0f / D) + eps);
  }
  const int toff = li * A_PITCH + g * 8;
#pragma unroll
  for (int ft = 0; ft < 4; ++ft) {
    const int f = wave * 64 + ft * 16 + g * 4;
    const f32x4 gg = *reinterpret_cast<const f32x4*>(gam + f), bb = *reinterpret_cast<const f32x4*>(bet + f);
#pragma unroll
    for (int rt = 0; rt < RTT3; ++rt) {
      float y[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) y[r] = (v[ft][rt][r] - mean[rt]) * rstd[rt] * gg[r] + bb[r];
      bf16x4 hi, lo;
      split4(y, hi, lo);
      const int o = toff + rt * 16 * A_PITCH + (wave * 64 + ft * 16) * 2;
      *reinterpret_cast<bf16x4*>(tile_hi + o) = hi;
      *reinterpret_cast<bf16x4*>(tile_lo + o) = lo;
    }
  }
  lds_barrier();
}

// one 512-wide row (8 values per lane) -> the two planes of the operand tile
__device__ __forceinline__ void row_to_tiles(const float* v, char* tile_hi, char* tile_lo, int lr, int lane) {
  bf16x8 hi, lo;
#pragma unroll
  for (int j = 0; j < 8; ++j) { hi[j] = (bf16_t)v[j]; lo[j] = (bf16_t)(v[j] - (float)hi[j]); }
  *reinterpret_cast<bf16x8*>(tile_hi + lr * A_PITCH + lane * 16) = hi;
  *reinterpret_cast<bf16x8*>(tile_lo + lr * A_PITCH + lane * 16) = lo;
}

// the attention output rows of this workgroup, split pairs bf16 [R, 1024] = [hi | lo], -> the two planes by LDS DMA (a plane row = 1 KB = one
// wave instruction); rows beyond R repeat the last row
__device__ __forceinline__ void stage_pairs(const bf16_t* __restrict__ att, int64_t r0, int R, char* tile_hi, char* tile_lo, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < RT3 / NW; ++i) {
    const int row = wave * (RT3 / NW) + i;
    int64_t r = r0 + row;
    if (r > R - 1) r = R - 1;
    const bf16_t* src = att + r * (2 * D) + lane * 8;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(tile_hi + row * A_PITCH), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + D), (__attribute__((address_space(3))) void*)(tile_lo + row * A_PITCH), 16, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <bool STORE, bool FULL>
__device__ __forceinline__ void add_bias_residual_t(f32x4 (&acc)[4][RTT3], const float* __restrict__ bias, float* __restrict__ xb, int nrow, int wave, int li_, int g_) {
  const int li = opaque(li_), g = opaque(g_);
  f32x4 bb[4], xv[4][RTT3];
#pragma unroll
  for (int ft = 0; ft < 4; ++ft) {
    const int f = wave * 64 + ft * 16 + g * 4;
    bb[ft] = *reinterpret_cast<const f32x4*>(bias + f);
#pragma unroll
    for (int rt = 0; rt < RTT3; ++rt) {
      const int lr = rt * 16 + li;
      const int lc = FULL ? lr : (lr < nrow ? lr : nrow - 1);
      xv[ft][rt] = *reinterpret_cast<const f32x4*>(xb + lc * D + f);
    }
  }
#pragma unroll
  for (int ft = 0; ft < 4; ++ft) {
    const int f = wave * 64 + ft * 16 + g * 4;
#pragma unroll
    for (int rt = 0; rt < RTT3; ++rt) {
      const int lr = rt * 16 + li;
      const f32x4 v = {acc[ft][rt][0] + bb[ft][0] + xv[ft][rt][0], acc[ft][rt][1] + bb[ft][1] + xv[ft][rt][1], acc[ft][rt][2] + bb[ft][2] + xv[ft][rt][2],
                       acc[ft][rt][3] + bb[ft][3] + xv[ft][rt][3]};
      acc[ft][rt] = v;
      if constexpr (STORE) {
        if (FULL || lr < nrow) *reinterpret_cast<f32x4*>(xb + lr * D + f) = v;
      }
    }
  }
}
template <bool STORE>
__device__ __forceinline__ void add_bias_residual(f32x4 (&acc)[4][RTT3], const float* __restrict__ bias, float* __restrict__ xb, int nrow, int wave, int li, int g) {
  if (nrow == RT3) add_bias_residual_t<STORE, true>(acc, bias, xb, nrow, wave, li, g);
  else add_bias_residual_t<STORE, false>(acc, bias, xb, nrow, wave, li, g);
}

// out[row][f] = acc + bias, fp32 (q / q k v / logits)
template <int NFT, bool FULL>
__device__ __forceinline__ void store_bias_t(const f32x4 (&acc)[NFT][RTT3], const f32x4 (&bb)[NFT], float* __restrict__ ob, int ld, int nrow, int fwave, int flimit, int li_, int g_) {
  const int li = opaque(li_), g = opaque(g_);
#pragma unroll
  for (int ft = 0; ft < NFT; ++ft) {
    const int f = fwave + ft * 16 + g * 4;
    if (f < flimit) {
#pragma unroll
      for (int rt = 0; rt < RTT3; ++rt) {
        const int lr = rt * 16 + li;
        if (FULL || lr < nrow)
          *reinterpret_cast<f32x4*>(ob + (int64_t)lr * ld + f) = f32x4{acc[ft][rt][0] + bb[ft][0], acc[ft][rt][1] + bb[ft][1], acc[ft][rt][2] + bb[ft][2], acc[ft][rt][3] + bb[ft][3]};
      }
    }
  }
}
template <int NFT>
__device__ __forceinline__ void store_bias(const f32x4 (&acc)[NFT][RTT3], const f32x4 (&bb)[NFT], float* __restrict__ ob, int ld, int nrow, int fwave, int flimit, int li, int g) {
  if (nrow == RT3) store_bias_t<NFT, true>(acc, bb, ob, ld, nrow, fwave, flimit, li, g);
  else store_bias_t<NFT, false>(acc, bb, ob, ld, nrow, fwave, flimit, li, g);
}

__device__ __forceinline__ void store_x(const f32x4 (&acc)[4][RTT3], float* __restrict__ xb, int nrow, int wave, int li_, int g_) {
  const int li = opaque(li_), g = opaque(g_);
  auto body = [&](auto FULL) {
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) {
      const int f = wave * 64 + ft * 16 + g * 4;
#pragma unroll
      for (int rt = 0; rt < RTT3; ++rt) {
        const int lr = rt * 16 + li;
        if (decltype(FULL)::value || lr < nrow) *reinterpret_cast<f32x4*>(xb + lr * D + f) = acc[ft][rt];
      }
    }
  };
  if (nrow == RT3) body(std::true_type());
  else body(std::false_type());
}

struct Rows3P {
  int R; float eps;
  const int32_t* d_pos;
  float* x;
  const bf16_t* att;               // split pairs [R, 1024] = [hi | lo]
  const char* wstream; int64_t wave_stride;
  const float* out_b;
  const float *ln_g, *ln_b;
  const float* qbias_tab; float* q;
  const float *ff1_b, *ff2_b;
  const int32_t* seq; int seq_ld; const float *word_emb, *pos_tab, *emb_g, *emb_b;
  const float *lnt_g, *lnt_b;
  const float* bias_tab; float* qkv;
  const float *h0_b, *h1_b, *h2_b;
  float* logits; int vocab;
};

// x' = x + att Wo^T + bo;  q = LayerNorm2(x') Wq^T + qbias[pos]  (fp32)          stream: Wo (64 pairs per wave), Wq (64 pairs)
__global__ __launch_bounds__(NW * 64) void dec_rows_x3_mid_kernel(Rows3P p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* tile_hi = smem;
  char* tile_lo = smem + TILE_BYTES;
  float* red = reinterpret_cast<float*>(smem + 2 * TILE_BYTES);
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t r0 = (int64_t)blockIdx.x * RT3;
  Stream st = stream_of_wave(p.wstream, p.wave_stride, wave, lane);
  u32x4 ring[PF3];
  sfor<PF3>([&](auto U) { ws_issue<decltype(U)::value>(ring, st); });
  const int pos = p.d_pos != nullptr ? *p.d_pos : 0;
  stage_pairs(p.att, r0, p.R, tile_hi, tile_lo, wave, lane);
  lds_barrier();
  const int loff = li * A_PITCH + g * 16;
  const int nrow = (int)((int64_t)p.R - r0 < RT3 ? (int64_t)p.R - r0 : RT3);
  f32x4 acc[4][RTT3];
  zero_acc(acc);
  gemm_pass_x3<4, 16, A_PITCH>(acc, tile_hi + loff, tile_lo + loff, ring, st);
  add_bias_residual<true>(acc, p.out_b, p.x + r0 * D, nrow, wave, li, g);
  ln_acc_to_tiles(acc, p.ln_g, p.ln_b, p.eps, tile_hi, tile_lo, red, wave, li, g);
  zero_acc(acc);
  f32x4 qb[4];
  load_bias<4>(qb, p.qbias_tab + (int64_t)pos * D, wave * 64, D, g);
  gemm_pass_x3<4, 16, A_PITCH>(acc, tile_hi + loff, tile_lo + loff, ring, st);
  ws_drain(ring);
  store_bias<4>(acc, qb, p.q + r0 * D, D, nrow, wave * 64, D, li, g);
}

// the chain behind the cross-attention (PRO 0), the embedding (PRO 1) or the bare residual stream (PRO 2), with the next layer's q k v (TAIL 0),
// the prediction head (TAIL 1) or nothing (TAIL 2) behind it; ACT 0 = ReLU (decoders), 1 = GELU (Swin).  See csrc/dec_rows.hip::dec_rows_ffn_kernel.
template <int PRO, int TAIL, int ACT>
__global__ __launch_bounds__(NW * 64) void dec_rows_x3_ffn_kernel(Rows3P p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* tile_hi = smem;
  char* tile_lo = smem + TILE_BYTES;
  char* h_hi = smem + 2 * TILE_BYTES;
  char* h_lo = h_hi + HT_BYTES;
  float* red = reinterpret_cast<float*>(h_lo + HT_BYTES);        // 2 x NW x RT3
  float* b1s = red + 2 * NW * RT3;                                // d_ff floats (PRO 0)
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t r0 = (int64_t)blockIdx.x * RT3;
  Stream st = stream_of_wave(p.wstream, p.wave_stride, wave, lane);
  u32x4 ring[PF3];
  sfor<PF3>([&](auto U) { ws_issue<decltype(U)::value>(ring, st); });
  const int pos = p.d_pos != nullptr ? *p.d_pos : 0;
  const int loff = li * A_PITCH + g * 16;
  const int nrow = (int)((int64_t)p.R - r0 < RT3 ? (int64_t)p.R - r0 : RT3);
  f32x4 acc[4][RTT3];

  if constexpr (PRO == 0) {
    static_assert(4 * D / 4 == NW * 64, "linear1's bias: one 16-byte piece per thread");
    const f32x4 b1v = reinterpret_cast<const f32x4*>(p.ff1_b)[tid];
    stage_pairs(p.att, r0, p.R, tile_hi, tile_lo, wave, lane);
    reinterpret_cast<f32x4*>(b1s)[tid] = b1v;
    lds_barrier();
    zero_acc(acc);
    gemm_pass_x3<4, 16, A_PITCH>(acc, tile_hi + loff, tile_lo + loff, ring, st);
    add_bias_residual<false>(acc, p.out_b, p.x + r0 * D, nrow, wave, li, g);
    ln_acc_to_tiles(acc, p.ln_g, p.ln_b, p.eps, tile_hi, tile_lo, red, wave, li, g);
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) {   // linear2's accumulators start as x1 + b2
      const f32x4 bb = *reinterpret_cast<const f32x4*>(p.ff2_b + wave * 64 + ft * 16 + g * 4);
#pragma unroll
      for (int rt = 0; rt < RTT3; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[ft][rt][r] += bb[r];
    }
    const int hoff = li * H_PITCH + g * 16;
#pragma unroll 1
    for (int c = 0; c < 4 * D / HC3; ++c) {
      f32x4 a1[1][RTT3];
      zero_acc(a1);
      gemm_pass_x3<1, 16, A_PITCH>(a1, tile_hi + loff, tile_lo + loff, ring, st);   // hidden units c * 128 + 16 w + 4 g + r of the rows
      lds_barrier();   // everybody has left linear2 of chunk c - 1: the hidden planes may be overwritten
      const f32x4 bb = *reinterpret_cast<const f32x4*>(b1s + c * HC3 + wave * 16 + g * 4);
#pragma unroll
      for (int rt = 0; rt < RTT3; ++rt) {
        float hv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) hv[r] = a1[0][rt][r] + bb[r];
        if constexpr (ACT == 1) {
          gelu_fast_n<4>(hv);   // the packed GELU of the split-pair fc1 epilogue (csrc/gemm.hip): 3.2e-7 absolute, below the pairs' resolution
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) hv[r] = fmaxf(hv[r], 0.f);
        }
        bf16x4 hi, lo;
        split4(hv, hi, lo);
        const int o = (rt * 16 + li) * H_PITCH + (wave * 16 + g * 4) * 2;
        *reinterpret_cast<bf16x4*>(h_hi + o) = hi;
        *reinterpret_cast<bf16x4*>(h_lo + o) = lo;
      }
      lds_barrier();   // chunk c is complete in LDS
      gemm_pass_x3<4, HC3 / 32, H_PITCH>(acc, h_hi + hoff, h_lo + hoff, ring, st);
    }
    if constexpr (TAIL == 2) ws_drain(ring);
    store_x(acc, p.x + r0 * D, nrow, wave, li, g);
    if constexpr (TAIL != 2) ln_acc_to_tiles(acc, p.lnt_g, p.lnt_b, p.eps, tile_hi, tile_lo, red, wave, li, g);
  } else if constexpr (PRO == 1) {
    constexpr int RPW = RT3 / NW;
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
      row_to_tiles(v, tile_hi, tile_lo, wave * RPW + i, lane);
    }
    lds_barrier();
  } else {
    constexpr int RPW = RT3 / NW;
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
      row_to_tiles(v, tile_hi, tile_lo, wave * RPW + i, lane);
    }
    lds_barrier();
  }

  if constexpr (TAIL == 0) {
    auto qkv_pass = [&](int ps, auto LAST) {
      zero_acc(acc);
      f32x4 bb[4];
      load_bias<4>(bb, p.bias_tab + (int64_t)pos * (3 * D) + ps * D, wave * 64, D, g);
      gemm_pass_x3<4, 16, A_PITCH>(acc, tile_hi + loff, tile_lo + loff, ring, st);
      if constexpr (decltype(LAST)::value) ws_drain(ring);
      store_bias<4>(acc, bb, p.qkv + r0 * (3 * D) + ps * D, 3 * D, nrow, wave * 64, D, li, g);
    };
#pragma unroll 1
    for (int ps = 0; ps < 2; ++ps) qkv_pass(ps, std::false_type());
    qkv_pass(2, std::true_type());
  } else if constexpr (TAIL == 1) {
#pragma unroll 1
    for (int hl = 0; hl < 2; ++hl) {
      zero_acc(acc);
      gemm_pass_x3<4, 16, A_PITCH>(acc, tile_hi + loff, tile_lo + loff, ring, st);
      lds_barrier();   // every wave has read the tile: it may be overwritten
      const float* hb_ = hl == 0 ? p.h0_b : p.h1_b;
#pragma unroll
      for (int ft = 0; ft < 4; ++ft) {
        const int f = wave * 64 + ft * 16 + g * 4;
        const f32x4 bb = *reinterpret_cast<const f32x4*>(hb_ + f);
#pragma unroll
        for (int rt = 0; rt < RTT3; ++rt) {
          float hv[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) hv[r] = fmaxf(acc[ft][rt][r] + bb[r], 0.f);
          bf16x4 hi, lo;
          split4(hv, hi, lo);
          const int o = (rt * 16 + li) * A_PITCH + f * 2;
          *reinterpret_cast<bf16x4*>(tile_hi + o) = hi;
          *reinterpret_cast<bf16x4*>(tile_lo + o) = lo;
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
      gemm_pass_x3<4, 16, A_PITCH>(acc, tile_hi + loff, tile_lo + loff, ring, st);
      ws_drain(ring);
      store_bias<4>(acc, bb, p.logits + r0 * V, V, nrow, ps * 512 + wave * 64, V, li, g);
    }
#pragma unroll 1
    for (int ps = 0; ps < n128; ++ps) {
      f32x4 a1[1][RTT3];
      zero_acc(a1);
      f32x4 bb[1];
      load_bias<1>(bb, p.h2_b, n512 * 512 + ps * 128 + wave * 16, V, g);
      gemm_pass_x3<1, 16, A_PITCH>(a1, tile_hi + loff, tile_lo + loff, ring, st);
      ws_drain(ring);
      store_bias<1>(a1, bb, p.logits + r0 * V, V, nrow, n512 * 512 + ps * 128 + wave * 16, V, li, g);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <typename K>
int raise_lds(K kern, const char* what) {
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
    omp_set_error("%s: cannot raise dynamic LDS limit", what);
    return OMP_ERR_LAUNCH;
  }
  return OMP_OK;
}

template <int PRO, int TAIL, int ACT>
int launch_ffn3(const Rows3P& p, hipStream_t st) {
  const size_t smem = (size_t)2 * TILE_BYTES + 2 * HT_BYTES + 2 * NW * RT3 * 4 + 4 * D * 4;
  auto kern = dec_rows_x3_ffn_kernel<PRO, TAIL, ACT>;
  static bool done = false;   // per instantiation
  if (!done) {
    const int rc = raise_lds(kern, "row-owner chain (bf16x3)");
    if (rc != OMP_OK) return rc;
    done = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(((int64_t)p.R + RT3 - 1) / RT3)), dim3(NW * 64), smem, st, p);
  OMP_CHECK_LAUNCH("row-owner chain (bf16x3)");
  return OMP_OK;
}

}  // namespace

// entry points for csrc/dec_rows.hip (omp_dec_rows_mid / omp_dec_rows_ffn / omp_swin_rows_block with x3 = 1): arguments already checked there
int omp_rows_x3_tile() { return RT3; }

int omp_rows_x3_mid(const omp_dec_rows_args* a, hipStream_t st) {
  Rows3P p{};
  p.R = a->R; p.eps = a->eps; p.d_pos = a->d_pos; p.x = a->x; p.att = reinterpret_cast<const bf16_t*>(a->att);
  p.wstream = reinterpret_cast<const char*>(a->wstream); p.wave_stride = a->wave_stride;
  p.out_b = a->out_b; p.ln_g = a->ln_g; p.ln_b = a->ln_b; p.qbias_tab = a->qbias_tab; p.q = reinterpret_cast<float*>(a->q);
  const size_t smem = (size_t)2 * TILE_BYTES + 2 * NW * RT3 * 4;
  static bool done = false;
  if (!done) {
    const int rc = raise_lds(dec_rows_x3_mid_kernel, "omp_dec_rows_mid(bf16x3)");
    if (rc != OMP_OK) return rc;
    done = true;
  }
  hipLaunchKernelGGL(dec_rows_x3_mid_kernel, dim3((unsigned)(((int64_t)p.R + RT3 - 1) / RT3)), dim3(NW * 64), smem, st, p);
  OMP_CHECK_LAUNCH("omp_dec_rows_mid(bf16x3)");
  return OMP_OK;
}

int omp_rows_x3_ffn(const omp_dec_rows_args* a, hipStream_t st) {
  Rows3P p{};
  p.R = a->R; p.eps = a->eps; p.d_pos = a->d_pos; p.x = a->x; p.att = reinterpret_cast<const bf16_t*>(a->att);
  p.wstream = reinterpret_cast<const char*>(a->wstream); p.wave_stride = a->wave_stride;
  p.out_b = a->out_b; p.ln_g = a->ln_g; p.ln_b = a->ln_b; p.ff1_b = a->ff1_b; p.ff2_b = a->ff2_b;
  p.seq = a->seq; p.seq_ld = a->seq_ld; p.word_emb = a->word_emb; p.pos_tab = a->pos_tab; p.emb_g = a->emb_g; p.emb_b = a->emb_b;
  p.lnt_g = a->lnt_g; p.lnt_b = a->lnt_b; p.bias_tab = a->bias_tab; p.qkv = reinterpret_cast<float*>(a->qkv);
  p.h0_b = a->h0_b; p.h1_b = a->h1_b; p.h2_b = a->h2_b; p.logits = a->logits; p.vocab = a->vocab;
  if (a->prologue == 0) return a->tail == 0 ? launch_ffn3<0, 0, 0>(p, st) : launch_ffn3<0, 1, 0>(p, st);
  return a->tail == 0 ? launch_ffn3<1, 0, 0>(p, st) : launch_ffn3<1, 1, 0>(p, st);
}

int omp_rows_x3_swin(const omp_swin_rows_args* a, hipStream_t st) {
  Rows3P p{};
  p.R = (int)a->M; p.eps = a->eps; p.d_pos = nullptr; p.x = a->x; p.att = reinterpret_cast<const bf16_t*>(a->att);
  p.wstream = reinterpret_cast<const char*>(a->wstream); p.wave_stride = a->wave_stride;
  p.out_b = a->proj_b; p.ln_g = a->n2_g; p.ln_b = a->n2_b; p.ff1_b = a->fc1_b; p.ff2_b = a->fc2_b;
  p.lnt_g = a->n1_g; p.lnt_b = a->n1_b; p.bias_tab = a->qkv_b; p.qkv = reinterpret_cast<float*>(a->qkv);
  if (a->mode == 0) return launch_ffn3<2, 0, 1>(p, st);
  return a->n1_g != nullptr ? launch_ffn3<0, 0, 1>(p, st) : launch_ffn3<0, 2, 1>(p, st);
}
