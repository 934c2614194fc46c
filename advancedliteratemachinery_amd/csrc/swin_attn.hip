// Fused Swin (shifted-)window attention core for gfx950.
//
// One workgroup = one 7x7 window x 4 heads (one wave per head, head_dim 32).  Everything the
// reference does between the qkv and proj Linears happens here with no intermediate tensors:
//   * F.pad to a multiple of 7 AFTER norm1: padded tokens do not exist in memory; their q/k/v are
//     the qkv bias (0 @ W + b), read from `qkv_bias`               (swin_transformer.py:209-215)
//   * torch.roll(-shift) / roll back: pure index arithmetic, token s of the shifted grid is
//     original token (s + shift) mod Hp                               (:218-221, :238-241)
//   * window_partition / window_reverse: index arithmetic             (:226-235)
//   * q*scale, q@k^T + relative_position_bias[(dy+6)*13+(dx+6)], SW-MSA mask (-100 where the
//     region ids {0: s<L-7, 1: s<L-3, 2: rest} of the two tokens differ), softmax, @v
//                                                                     (:129-147, :369-387)
//   * crop back to HxW: padded query rows are simply not stored       (:243-244)
// Lane i (< 49) owns query row i: q[32], the 49 scores and o[32] live in registers, so softmax
// needs no cross-lane traffic; K and V of the window/head are staged once in LDS (fp32) and read
// as wave-wide broadcasts (conflict-free ds_read_b128).
#include "common.h"
#include <atomic>

namespace {

constexpr int WS = 7, WT = 49, HD = 32;

template <typename T>
__global__ __launch_bounds__(256) void swin_attn_kernel(const T* __restrict__ qkv,
                                                         const float* __restrict__ qkv_bias,
                                                         const float* __restrict__ table,
                                                         T* __restrict__ out, int B, int H, int W, int C,
                                                         int nH, int shift, int nWy, int nWx) {
  constexpr int NV = Vec16<T>::N;
  constexpr int CPR = HD / NV;  // 16-byte chunks per 32-dim head row
  __shared__ __attribute__((aligned(16))) float kv[4][2][WT * HD];
  __shared__ float tab[4][176];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int head = blockIdx.y * 4 + wave;
  const bool hact = head < nH;
  int widx = blockIdx.x;
  const int wx = widx % nWx; widx /= nWx;
  const int wy = widx % nWy;
  const int b = widx / nWy;
  const int Hp = nWy * WS, Wp = nWx * WS;
  const int C3 = 3 * C;
  float* ks = kv[wave][0];
  float* vs = kv[wave][1];

  if (hact) {
    // stage K and V of this (window, head) in LDS as fp32
    for (int idx = lane; idx < WT * CPR; idx += 64) {
      const int t = idx / CPR, cc = idx - t * CPR;
      int py = wy * WS + t / WS + shift, px = wx * WS + t % WS + shift;
      if (py >= Hp) py -= Hp;
      if (px >= Wp) px -= Wp;
      float kk[NV], vv[NV];
      if (py < H && px < W) {
        const T* row = qkv + (((int64_t)b * H + py) * W + px) * C3 + head * HD + cc * NV;
        unpack16(ld16<T>(row + C), kk);
        unpack16(ld16<T>(row + 2 * C), vv);
      } else {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          kk[i] = qkv_bias[C + head * HD + cc * NV + i];
          vv[i] = qkv_bias[2 * C + head * HD + cc * NV + i];
        }
      }
#pragma unroll
      for (int i = 0; i < NV; i += 4) {
        *reinterpret_cast<f32x4*>(ks + t * HD + cc * NV + i) = f32x4{kk[i], kk[i + 1], kk[i + 2], kk[i + 3]};
        *reinterpret_cast<f32x4*>(vs + t * HD + cc * NV + i) = f32x4{vv[i], vv[i + 1], vv[i + 2], vv[i + 3]};
      }
    }
    for (int idx = lane; idx < 169; idx += 64) tab[wave][idx] = table[idx * nH + head];
  }
  __syncthreads();
  if (!hact || lane >= WT) return;

  // this lane's query token
  const int ty = lane / WS, tx = lane % WS;
  const int sy = wy * WS + ty, sx = wx * WS + tx;
  int py = sy + shift, px = sx + shift;
  if (py >= Hp) py -= Hp;
  if (px >= Wp) px -= Wp;
  const bool qvalid = py < H && px < W;
  const int64_t tok = ((int64_t)b * H + py) * W + px;
  const float scale = 0.17677669529663687f;  // 32^-0.5
  const float scale2 = 0.17677669529663687f * 1.4426950408889634f;   // scale * log2(e): base-2 softmax (EXPB)
  // SW-MSA: only the last window row / column of the padded grid mixes regions (swin_transformer.py:369-387)
  const bool edge = shift > 0 && (wy == nWy - 1 || wx == nWx - 1);
  float q[HD];
  if (qvalid) {
    const T* row = qkv + tok * C3 + head * HD;
#pragma unroll
    for (int c = 0; c < CPR; ++c) unpack16(ld16<T>(row + c * NV), q + c * NV);
  } else {
#pragma unroll
    for (int d = 0; d < HD; ++d) q[d] = qkv_bias[head * HD + d];
  }
#pragma unroll
  for (int d = 0; d < HD; ++d) q[d] *= scale;

  int rid_i = 0;
  if (shift > 0) {
    const int ry = sy < Hp - WS ? 0 : (sy < Hp - shift ? 1 : 2);
    const int rx = sx < Wp - WS ? 0 : (sx < Wp - shift ? 1 : 2);
    rid_i = ry * 3 + rx;
  }

  float sc[WT];
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < WT; ++j) {
    const int jy = j / WS, jx = j % WS;
    float a = 0.f;
#pragma unroll
    for (int d = 0; d < HD; d += 4) {
      const f32x4 k4 = *reinterpret_cast<const f32x4*>(ks + j * HD + d);
      a = fmaf(q[d], k4[0], a);
      a = fmaf(q[d + 1], k4[1], a);
      a = fmaf(q[d + 2], k4[2], a);
      a = fmaf(q[d + 3], k4[3], a);
    }
    a += tab[wave][(ty - jy + WS - 1) * (2 * WS - 1) + (tx - jx + WS - 1)];
    if (shift > 0) {
      const int ssy = wy * WS + jy, ssx = wx * WS + jx;
      const int ry = ssy < Hp - WS ? 0 : (ssy < Hp - shift ? 1 : 2);
      const int rx = ssx < Wp - WS ? 0 : (ssx < Wp - shift ? 1 : 2);
      if (ry * 3 + rx != rid_i) a += -100.0f;
    }
    sc[j] = a;
    mx = fmaxf(mx, a);
  }
  float l = 0.f;
#pragma unroll
  for (int j = 0; j < WT; ++j) {
    sc[j] = expf(sc[j] - mx);
    l += sc[j];
  }
  const float inv = 1.0f / l;
  float o[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) o[d] = 0.f;
#pragma unroll
  for (int j = 0; j < WT; ++j) {
    const float pj = sc[j] * inv;
#pragma unroll
    for (int d = 0; d < HD; d += 4) {
      const f32x4 v4 = *reinterpret_cast<const f32x4*>(vs + j * HD + d);
      o[d] = fmaf(pj, v4[0], o[d]);
      o[d + 1] = fmaf(pj, v4[1], o[d + 1]);
      o[d + 2] = fmaf(pj, v4[2], o[d + 2]);
      o[d + 3] = fmaf(pj, v4[3], o[d + 3]);
    }
  }
  if (qvalid) {
    T* dst = out + tok * C + head * HD;
#pragma unroll
    for (int c = 0; c < CPR; ++c) {
      typename Vec16<T>::type pv;
      pack16(o + c * NV, pv);
      st16<T>(dst + c * NV, pv);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Matrix-core version (the one that runs; the scalar kernel above is kept as an independent
// cross-check, selectable with omp_debug_swin_attn_impl).
//
// One wave = one (window, head).  The 49 tokens are padded to 64 = 4 tiles of 16:
//   S^T[key][query] = K[key,:] . Q[query,:]       head_dim 32 = ONE k-step of the 16x16x32 bf16 MFMA
//                                                  (two 16x16x16-equivalent steps of 4 f32 MFMAs)
//   softmax over keys in registers: a lane owns query (l & 15) of each query tile and 4 keys
//   {4*(l>>4) + r} of each key tile -> 16 scores per query tile in-lane + two xor-shuffles
//   O^T[d][query] += V^T[d][key] * P[key][query]   P comes straight from the S^T accumulators
// Q and K fragments are 16-byte loads straight from the qkv rows in HBM (8 per lane); only V needs a
// transpose and goes through LDS once (written as [dim][key] in the k-slot order the second product
// wants, pitch padded so the ds_read_b128 of a fragment is conflict-free).  Everything else the
// reference does between the qkv and proj Linears is index arithmetic exactly as in the scalar kernel.
// Per wave: 32 MFMAs for 12.5 KB of HBM traffic -> the kernel is bound by HBM, not by LDS/VALU.
// ---------------------------------------------------------------------------------------------
template <typename T>
struct SwinTraits;
template <>
struct SwinTraits<bf16_t> {
  static constexpr int QS = 1;    // k-steps over head_dim 32 in QK^T
  static constexpr int PS = 2;    // k-steps over the 64 (padded) keys in PV
  static constexpr int KPS = 32;  // keys per PV k-step
  static constexpr int VP = 72;   // V^T row pitch in elements (64 keys + pad: 144 B)
  __device__ static __forceinline__ int slot(int j) {   // key j -> position inside its 32-key step
    const int kl = j & 31;
    return (j & 32) + ((kl & 15) >> 2) * 8 + (kl >> 4) * 4 + (kl & 3);
  }
  __device__ static __forceinline__ bf16x8 pfrag(const float* a, const float* b) {
    bf16x8 f = {(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3], (bf16_t)b[0], (bf16_t)b[1], (bf16_t)b[2], (bf16_t)b[3]};
    return f;
  }
};
template <>
struct SwinTraits<float> {
  static constexpr int QS = 2;
  static constexpr int PS = 4;
  static constexpr int KPS = 16;
  static constexpr int VP = 68;   // 272 B
  __device__ static __forceinline__ int slot(int j) { return j; }
  __device__ static __forceinline__ f32x4 pfrag(const float* a, const float*) { return f32x4{a[0], a[1], a[2], a[3]}; }
};

// EXPB: the relative-position bias arrives EXPANDED per head as fp32 [64 queries][64 keys] (omp_swin_expand_bias: bias / scale,
// -inf on the 15 padding key slots), so it seeds the S^T accumulators instead of 64 per-score table lookups; the softmax then runs
// in base 2 (one v_exp_f32 per score) and the SW-MSA mask arithmetic is skipped for the interior windows, whose tokens all sit in
// region 0.
//
// Round 5 -- what paces this kernel is neither bytes (0.37 of the HBM floor) nor VALU issue (a third fewer vector instructions bought
// 3 %): wave 0's s_memtime stamps (TRACE, profiles/r05u_kbench_swin_attn_trace.txt) show 10 k of a (window, head)'s 25 k cycles passing
// while its twelve q / k / v loads ISSUE -- the request rate of a CU's L1 towards L2, ~330 cache-line requests per (window, head).  So:
//   * PERSISTENT waves: the launcher sizes grid.x to what is co-resident and a wave walks windows blockIdx.x, + gridDim.x, ... of ITS head,
//     whose expanded bias tile (64 registers per lane) is fetched once -- it was 128 of the 330 requests;
//   * ONE token per lane: lane t resolves window token min(t, 48) (row of qkv or -1: padding token; SW-MSA region), the (tile, lane) -> token
//     lookups the loads and the mask need are ds_bpermute reads of that register instead of eight copies of the divide / wrap / bounds arithmetic;
//   * padding tokens (q / k / v = bias: the last window row / column of an image whose size is not a multiple of 7 -- 19 of the 100 windows of
//     a 64 x 64 map, 36 under SW-MSA) read token 0 and get the bias SELECTED in once everything is in flight; the substitution used to be a
//     branch with its own loads behind each load, and the waits inside it made an edge window's twelve loads twelve serial round trips;
//   * bf16 (PIPE): the NEXT window's loads are issued before this window is computed (two register sets, the loop unrolled by two), and two
//     workgroups per CU instead of four: 195 -> 150 us per 131 072 tokens with the first three, -> see profiles/r05y_* with the prefetch.
// TRACE (development, omp_debug_swin_mlp_trace): wave 0 of every workgroup writes s_memtime sums over its windows [workgroup][8]: 0 all of
// 1..4, 1 token resolution + load issue (of the next window under PIPE), 2 until this window's q / k / v have landed and V^T sits in LDS,
// 3 the four query tiles (S^T, softmax, PV, store), 5 = windows.
template <typename T, bool EXPB, bool TRACE = false>
__global__ __launch_bounds__(256, 2) void swin_attn_mfma_kernel(const T* __restrict__ qkv,
                                                              const float* __restrict__ qkv_bias,
                                                              const float* __restrict__ table,
                                                              T* __restrict__ out, int B, int H, int W, int C,
                                                              int nH, int shift, int nWy, int nWx, int out_split,
                                                              unsigned long long* __restrict__ trace = nullptr) {
  typedef Mma<T> MM;
  typedef SwinTraits<T> ST;
  typedef typename MM::frag frag;
  constexpr int NV = Vec16<T>::N;          // elements per 16-byte chunk (8 / 4)
  constexpr int VP = ST::VP;
  constexpr int CPR = HD / NV;             // chunks per 32-dim row: 4 (bf16) / 8 (f32)
  constexpr int KPI = 64 / CPR;            // keys per V iteration
  constexpr int NIT = 64 / KPI;            // V iterations: 4 (bf16) / 8 (f32)
  constexpr bool PIPE = EXPB && sizeof(T) == 2;   // fp32 operands: two register sets do not fit beside the bias tile
  __shared__ __attribute__((aligned(16))) T vt[4][HD * VP];   // per wave: V^T [32 dims][64 key slots (+pad)]
  __shared__ float tab[4][176];
  __shared__ __attribute__((aligned(16))) T padb[4][2 * ST::QS + 1][64 * NV];   // per wave and lane: the q / k / v bias chunks a padding token takes

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, g = lane >> 4;
  const int head = blockIdx.y * 4 + wave;
  if (head >= nH) return;                  // whole wave; no block-wide barrier below
  const int Hp = nWy * WS, Wp = nWx * WS;
  const int C3 = 3 * C;
  T* vs = vt[wave];
  const float scale = 0.17677669529663687f;  // 32^-0.5
  const float scale2 = 0.17677669529663687f * 1.4426950408889634f;   // scale * log2(e): base-2 softmax (EXPB)
  const T* qkv_head = qkv + head * HD;
  auto from_lane = [](int v, int src) -> int { return __builtin_amdgcn_ds_bpermute(src << 2, v); };

  const float* be = table + (int64_t)head * 4096 + li * 64 + g * 4;   // [query][key] of this head (EXPB)
  f32x4 bias_all[4][4];   // [query tile][key tile]
  if constexpr (EXPB) {
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) bias_all[t4][kt] = *reinterpret_cast<const f32x4*>(be + t4 * 16 * 64 + kt * 16);
  } else {
    for (int idx = lane; idx < 169; idx += 64) tab[wave][idx] = table[idx * nH + head];
  }

  // The bias chunks this lane substitutes for padding tokens, once per wave, parked in LDS: reading them back costs no vmcnt wait, so the
  // substitution never drains the prefetched window's loads.
  {
    auto bias_chunk = [&](int sel, int d0) -> frag {
      float t[NV];
#pragma unroll
      for (int i = 0; i < NV; ++i) t[i] = qkv_bias[sel * C + head * HD + d0 + i];
      frag f;
      pack16(t, f);
      return f;
    };
#pragma unroll
    for (int s = 0; s < ST::QS; ++s) {
      *reinterpret_cast<frag*>(&padb[wave][2 * s][lane * NV]) = bias_chunk(0, s * MM::KSTEP + g * MM::KPL);
      *reinterpret_cast<frag*>(&padb[wave][2 * s + 1][lane * NV]) = bias_chunk(1, s * MM::KSTEP + g * MM::KPL);
    }
    *reinterpret_cast<frag*>(&padb[wave][2 * ST::QS][lane * NV]) = bias_chunk(2, (lane % CPR) * NV);
  }

  // everything of a window that is in flight or pending between its load issue and its computation
  struct Win {
    frag qf[4][ST::QS], kf[4][ST::QS], vchunk[NIT];
    int qtok[4];       // row of the lane's query of tile t4 (or -1: padding token / beyond the 49)
    int my_rid;        // SW-MSA region of the token this LANE resolved
    unsigned pad;      // bit t4: q / k of tile t4 belong to a padding token; bit 4 + it: v chunk it does
    bool all_real, edge, masked;   // wave-uniform
  };
  unsigned long long t_sum[5] = {0ull, 0ull, 0ull, 0ull, 0ull};
  unsigned long long t_last = 0ull;
  auto lap = [&](int i, bool drain) {
    if constexpr (TRACE) {
      if (drain) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      else asm volatile("" ::: "memory");
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      t_sum[i] += t - t_last;
      t_last = t;
    }
  };

  // ---- token resolution + every global load of window wcur ----------------------------------------
  auto issue = [&](int wcur, Win& w) {
    int widx = wcur;
    const int wx = widx % nWx; widx /= nWx;
    const int wy = widx % nWy;
    const int b = widx / nWy;
    // SW-MSA: only the last window row / column of the padded grid mixes regions (swin_transformer.py:369-387)
    w.edge = shift > 0 && (wy == nWy - 1 || wx == nWx - 1);
    w.masked = EXPB ? w.edge : shift > 0;
    const int sy_hi = wy * WS + WS - 1 + shift, sx_hi = wx * WS + WS - 1 + shift;
    w.all_real = (sy_hi < Hp ? sy_hi : Hp - 1) < H && (sx_hi < Wp ? sx_hi : Wp - 1) < W;
    int my_tok;
    w.my_rid = 0;
    {
      const int t = lane < WT ? lane : WT - 1;
      const int ty = (t * 37) >> 8, tx = t - ty * WS;   // t / 7 for t < 64
      const int sy = wy * WS + ty, sx = wx * WS + tx;
      int py = sy + shift, px = sx + shift;
      if (py >= Hp) py -= Hp;
      if (px >= Wp) px -= Wp;
      my_tok = (py < H && px < W) ? (b * H + py) * W + px : -1;   // token rows fit 32 bits (checked by the launcher)
      if (w.masked) {
        const int ry = sy < Hp - WS ? 0 : (sy < Hp - shift ? 1 : 2);
        const int rx = sx < Wp - WS ? 0 : (sx < Wp - shift ? 1 : 2);
        w.my_rid = ry * 3 + rx;
      }
    }
    auto load_chunk = [&](int tok, int sel, int d0) -> frag {   // 16 bytes of q (sel 0) / k (1) / v (2) of token tok at head dims [d0, d0 + NV)
      return ld16<T>(qkv_head + (uint64_t)(uint32_t)(tok >= 0 ? tok : 0) * (uint32_t)C3 + (sel * C + d0));
    };
    w.pad = 0u;
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) {
      int i = t4 * 16 + li; if (i > WT - 1) i = WT - 1;      // clamped rows are never stored / are masked
      const int tok = from_lane(my_tok, i);
      w.qtok[t4] = (t4 * 16 + li < WT) ? tok : -1;
      w.pad |= tok < 0 ? (1u << t4) : 0u;
#pragma unroll
      for (int s = 0; s < ST::QS; ++s) {
        w.qf[t4][s] = load_chunk(tok, 0, s * MM::KSTEP + g * MM::KPL);
        w.kf[t4][s] = load_chunk(tok, 1, s * MM::KSTEP + g * MM::KPL);
      }
    }
    // V: lane (key = it*KPI + lane/CPR, chunk = lane % CPR) -> 16-byte row pieces, coalesced per token
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int j = it * KPI + lane / CPR, dc = lane % CPR;
      const int tok = from_lane(my_tok, j < WT ? j : WT - 1);   // every lane takes part in the permute
      w.pad |= (j < WT && tok < 0) ? (16u << it) : 0u;
      w.vchunk[it] = load_chunk(tok, 2, dc * NV);   // every lane loads (key slots >= 49 re-read token 48 and are zeroed in compute): no
    }                                                // conditional load, so the count of loads in flight is the same on every path
  };

  // ---- S^T = K Q^T, softmax, O^T = V^T P, store -------------------------------------------------
  auto compute = [&](Win& w) {
    if (!w.all_real) {   // padding tokens: the lane's bias chunks (LDS) selected in
#pragma unroll
      for (int s = 0; s < ST::QS; ++s) {
        const frag bq = *reinterpret_cast<const frag*>(&padb[wave][2 * s][lane * NV]);
        const frag bk = *reinterpret_cast<const frag*>(&padb[wave][2 * s + 1][lane * NV]);
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
          w.qf[t4][s] = (w.pad >> t4) & 1u ? bq : w.qf[t4][s];
          w.kf[t4][s] = (w.pad >> t4) & 1u ? bk : w.kf[t4][s];
        }
      }
      const frag bv = *reinterpret_cast<const frag*>(&padb[wave][2 * ST::QS][lane * NV]);
#pragma unroll
      for (int it = 0; it < NIT; ++it) w.vchunk[it] = (w.pad >> (4 + it)) & 1u ? bv : w.vchunk[it];
    }
    {   // key slots beyond the 49 tokens must be finite: P = 0 there (only the last V iteration holds any)
      float z[NV];
#pragma unroll
      for (int i = 0; i < NV; ++i) z[i] = 0.f;
      frag zf;
      pack16(z, zf);
#pragma unroll
      for (int it = 0; it < NIT; ++it)
        if ((it + 1) * KPI > WT) w.vchunk[it] = (it * KPI + lane / CPR < WT) ? w.vchunk[it] : zf;
    }
    // V^T -> LDS (the previous window's fragment reads are older LDS operations of this wave: in order)
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int j = it * KPI + lane / CPR, dc = lane % CPR;
      const int pos = ST::slot(j);
#pragma unroll
      for (int e = 0; e < NV; ++e) vs[(dc * NV + e) * VP + pos] = w.vchunk[it][e];
    }
    lap(2, true);
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wave's V^T and table stores have landed (own data only)
    // a query tile at a time, stored as soon as it is done: 8 accumulator registers live instead of 32 (the prefetched window needs the room)
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) {
      f32x4 oacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};   // [dim tile]
      // geometry of this lane's query of tile t4 (table variant)
      const int i = t4 * 16 + li;
      const int ic = i < WT ? i : WT - 1;
      const int ity = (i * 37) >> 8, itx = i - ity * WS;
      float sc[16];
      float mx = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        f32x4 st = {0.f, 0.f, 0.f, 0.f};
        if constexpr (EXPB) st = bias_all[t4][kt];
#pragma unroll
        for (int s = 0; s < ST::QS; ++s) MM::mma(st, w.kf[kt][s], w.qf[t4][s]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = kt * 16 + g * 4 + r;   // acc[r] of key tile kt is key j = kt*16 + 4g + r
          float a;
          if constexpr (EXPB) {
            a = st[r] * scale2;   // (q.k + bias / scale) * scale * log2 e; padding keys are -inf through the seed
          } else {
            a = st[r] * scale;
            if (j < WT && i < WT) {
              const int jy = (j * 37) >> 8, jx = j - jy * WS;
              a += tab[wave][(ity - jy + WS - 1) * (2 * WS - 1) + (itx - jx + WS - 1)];
            } else if (j >= WT) {
              a = -INFINITY;
            }
          }
          sc[kt * 4 + r] = a;
        }
      }
      if (w.masked) {   // wave-uniform: the SW-MSA mask costs the interior windows no instruction; regions by permute from the lanes that own the tokens
        const int rid_i = from_lane(w.my_rid, ic);
        const float m = EXPB ? -100.0f * 1.4426950408889634f : -100.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          const int j = (k >> 2) * 16 + g * 4 + (k & 3);
          const int rid_j = from_lane(w.my_rid, j < WT ? j : WT - 1);
          if (EXPB || (j < WT && i < WT)) sc[k] += rid_j != rid_i ? m : 0.f;
        }
      }
#pragma unroll
      for (int k = 0; k < 16; ++k) mx = fmaxf(mx, sc[k]);
      mx = quad_group_max(mx);
      float l = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        if constexpr (EXPB) sc[k] = __builtin_amdgcn_exp2f(sc[k] - mx);
        else sc[k] = expf(sc[k] - mx);
        l += sc[k];
      }
      l = quad_group_sum(l);
      const float inv = 1.0f / l;
#pragma unroll
      for (int k = 0; k < 16; ++k) sc[k] *= inv;
      // O^T += V^T P : k-step ps covers keys [ps*KPS, +KPS)
#pragma unroll
      for (int ps = 0; ps < ST::PS; ++ps) {
        const frag pf = ST::pfrag(sc + ps * (ST::KPS / 4), sc + ps * (ST::KPS / 4) + 4);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const frag vf = *reinterpret_cast<const frag*>(vs + (dt * 16 + li) * VP + ps * ST::KPS + g * MM::KPL);
          MM::mma(oacc[dt], vf, pf);
        }
      }
      // store: acc[r] <-> (dim = dt*16 + 4g + r, query = t4*16 + li)
      if (w.qtok[t4] >= 0) {
        T* dst = out + (uint64_t)(uint32_t)w.qtok[t4] * (uint32_t)C + (head * HD + g * 4);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const f32x4 o = oacc[dt];
          if constexpr (sizeof(T) == 4) {
            if (out_split) {   // fp32 engine with bf16x3 products: the proj GEMM reads split pairs [hi | lo] (OMP_BF16X2)
              bf16_t* ds = reinterpret_cast<bf16_t*>(out) + (uint64_t)(uint32_t)w.qtok[t4] * (uint32_t)(2 * C) + (head * HD + g * 4 + dt * 16);
              bf16x4 hi, lo;
#pragma unroll
              for (int e = 0; e < 4; ++e) { hi[e] = (bf16_t)o[e]; lo[e] = (bf16_t)(o[e] - (float)hi[e]); }
              *reinterpret_cast<bf16x4*>(ds) = hi;
              *reinterpret_cast<bf16x4*>(ds + C) = lo;
              continue;
            }
            *reinterpret_cast<f32x4*>(dst + dt * 16) = o;
          } else {
            bf16x4 ov = {(bf16_t)o[0], (bf16_t)o[1], (bf16_t)o[2], (bf16_t)o[3]};
            *reinterpret_cast<bf16x4*>(dst + dt * 16) = ov;
          }
        }
      }
    }
    lap(3, false);
  };

  const int nWin = B * nWy * nWx;
  const int step = (int)gridDim.x;
  int nwalked = 0;
  if constexpr (TRACE) t_last = __builtin_amdgcn_s_memtime();
  if constexpr (PIPE) {
    // the next window's loads are issued UNCONDITIONALLY (the last one re-reads its own window): a conditional issue makes the number of
    // loads in flight path-dependent and the compiler's waits fall back to vmcnt(0), which drains the prefetch
    Win wa, wb;
    int wcur = blockIdx.x;
    if (wcur < nWin) issue(wcur, wa);
    lap(1, false);
    while (wcur < nWin) {
      issue(wcur + step < nWin ? wcur + step : wcur, wb);
      lap(1, false);
      compute(wa);
      ++nwalked;
      wcur += step;
      if (wcur >= nWin) break;
      issue(wcur + step < nWin ? wcur + step : wcur, wa);
      lap(1, false);
      compute(wb);
      ++nwalked;
      wcur += step;
    }
  } else {
    for (int wcur = blockIdx.x; wcur < nWin; wcur += step) {
      Win w;
      issue(wcur, w);
      lap(1, false);
      compute(w);
      ++nwalked;
    }
  }
  if constexpr (TRACE) {
    if (wave == 0 && lane == 0 && trace != nullptr) {
      unsigned long long* tr = trace + ((int64_t)blockIdx.x * gridDim.y + blockIdx.y) * 8;
      tr[0] = t_sum[1] + t_sum[2] + t_sum[3] + t_sum[4];
#pragma unroll
      for (int i = 1; i < 5; ++i) tr[i] = t_sum[i];
      tr[5] = (unsigned long long)nwalked;
    }
  }
}


// ---------------------------------------------------------------------------------------------
// fp32-grade window attention on the bf16 matrix cores (round 4, the parity engine): fp32 q / k / v in, every operand split
// x = hi + lo (hi = bf16(x), lo = bf16(x - hi): 16 mantissa bits) and both products run as three bf16 MFMAs --
//   S^T = K_hi Q_hi^T + K_lo Q_hi^T + K_hi Q_lo^T,   O^T = V^T_hi P_hi + V^T_lo P_hi + V^T_hi P_lo
// (the lo x lo terms are 2^-16 relative).  swin_attn_mfma_kernel<float> spends 256 fp32 matrix-core instructions of 32 cycles per
// (window, head) and is paced by them (400 us per stage-2 launch of 32 images against 206 us for the bf16 kernel,
// profiles/r04a_kernel_shapes_parity_engine_graph0.txt); this one issues 96 of 16 cycles.  Same decomposition, index arithmetic,
// expanded bias, base-2 softmax and mask handling as swin_attn_mfma_kernel<bf16_t, true>; fp32 or split-pair rows out.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void swin_attn_x3_kernel(const float* __restrict__ qkv, const float* __restrict__ qkv_bias,
                                                              const float* __restrict__ bias_exp, float* __restrict__ out, int B, int H,
                                                              int W, int C, int nH, int shift, int nWy, int nWx, int out_split) {
  typedef Mma<bf16_t> MM;
  typedef SwinTraits<bf16_t> ST;
  typedef bf16x8 frag;
  constexpr int VP = ST::VP;
  __shared__ __attribute__((aligned(16))) bf16_t vt[4][2][HD * VP];   // per wave: V^T hi / lo planes [32 dims][64 key slots (+pad)]

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, g = lane >> 4;
  const int head = blockIdx.y * 4 + wave;
  if (head >= nH) return;
  const int Hp = nWy * WS, Wp = nWx * WS;
  const int C3 = 3 * C;
  bf16_t* vh = vt[wave][0];
  bf16_t* vl = vt[wave][1];
  const float scale2 = 0.17677669529663687f * 1.4426950408889634f;   // 32^-0.5 * log2(e): base-2 softmax
  auto split8 = [](const float* v, frag& hi, frag& lo) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { hi[i] = (bf16_t)v[i]; lo[i] = (bf16_t)(v[i] - (float)hi[i]); }
  };
  auto from_lane = [](int v, int src) -> int { return __builtin_amdgcn_ds_bpermute(src << 2, v); };
  // persistent over windows, the head's expanded bias tile in registers, one token resolved per lane: as swin_attn_mfma_kernel (round 5)
  const float* be = bias_exp + (int64_t)head * 4096 + li * 64 + g * 4;   // [query][key] of this head
  f32x4 bias_all[4][4];
#pragma unroll
  for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) bias_all[t4][kt] = *reinterpret_cast<const f32x4*>(be + t4 * 16 * 64 + kt * 16);
  const float* qkv_head = qkv + head * HD;
  const int nWin = B * nWy * nWx;
  for (int wcur = blockIdx.x; wcur < nWin; wcur += gridDim.x) {
    int widx = wcur;
    const int wx = widx % nWx; widx /= nWx;
    const int wy = widx % nWy;
    const int b = widx / nWy;
    const bool edge = shift > 0 && (wy == nWy - 1 || wx == nWx - 1);   // wave-uniform
    int my_tok, my_rid = 0;
    {
      const int t = lane < WT ? lane : WT - 1;
      const int ty = (t * 37) >> 8, tx = t - ty * WS;
      const int sy = wy * WS + ty, sx = wx * WS + tx;
      int py = sy + shift, px = sx + shift;
      if (py >= Hp) py -= Hp;
      if (px >= Wp) px -= Wp;
      my_tok = (py < H && px < W) ? (b * H + py) * W + px : -1;
      if (edge) {
        const int ry = sy < Hp - WS ? 0 : (sy < Hp - shift ? 1 : 2);
        const int rx = sx < Wp - WS ? 0 : (sx < Wp - shift ? 1 : 2);
        my_rid = ry * 3 + rx;
      }
    }
    const int sy_hi = wy * WS + WS - 1 + shift, sx_hi = wx * WS + WS - 1 + shift;
    const bool all_real = (sy_hi < Hp ? sy_hi : Hp - 1) < H && (sx_hi < Wp ? sx_hi : Wp - 1) < W;
    // 8 fp32 values of q (sel 0) / k (1) / v (2) of token tok at head dims [d0, d0 + 8); padding tokens read token 0 and get the bias selected
    // in once every load of the wave is in flight (no loads, hence no waits, inside a branch between the loads)
    auto load8 = [&](int tok, int sel, int d0, float* v) {
      const float* src = qkv_head + (uint64_t)(uint32_t)(tok >= 0 ? tok : 0) * (uint32_t)C3 + (sel * C + d0);
      unpack16(ld16<float>(src), v);
      unpack16(ld16<float>(src + 4), v + 4);
    };

    // ---- every global load of the wave first: q / k of the four 16-token tiles (lane: token li, dims 8 g .. 8 g + 7), v (lane: key it * 16 + lane / 4,
    //      dims 8 (lane % 4) ..) ----------------------------------------------------------------------------------------
    float qraw[4][8], kraw[4][8], vraw[4][8];
    int qtok[4], qrid[4];
    bool qpad[4], vpad[4];
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) {
      int i = t4 * 16 + li; if (i > WT - 1) i = WT - 1;
      const int tok = from_lane(my_tok, i);
      qtok[t4] = (t4 * 16 + li < WT) ? tok : -1;
      qpad[t4] = tok < 0;
      qrid[t4] = 0;
      if (edge) qrid[t4] = from_lane(my_rid, i);
      load8(tok, 0, g * 8, qraw[t4]);
      load8(tok, 1, g * 8, kraw[t4]);
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int j = it * 16 + (lane >> 2), dc = lane & 3;
      const int tok = from_lane(my_tok, j < WT ? j : WT - 1);
      vpad[it] = j < WT && tok < 0;
      if (j < WT) {
        load8(tok, 2, dc * 8, vraw[it]);
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) vraw[it][i] = 0.f;   // padded key slots must be finite: P = 0 there
      }
    }
    if (!all_real) {
      float bq[8], bk[8], bv[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        bq[i] = qkv_bias[head * HD + g * 8 + i];
        bk[i] = qkv_bias[C + head * HD + g * 8 + i];
        bv[i] = qkv_bias[2 * C + head * HD + (lane & 3) * 8 + i];
      }
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          qraw[t4][i] = qpad[t4] ? bq[i] : qraw[t4][i];
          kraw[t4][i] = qpad[t4] ? bk[i] : kraw[t4][i];
          vraw[t4][i] = vpad[t4] ? bv[i] : vraw[t4][i];
        }
    }
    frag qh[4], ql[4], kh[4], kl[4];
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) {
      split8(qraw[t4], qh[t4], ql[t4]);
      split8(kraw[t4], kh[t4], kl[t4]);
    }
    // ---- V^T -> LDS (both planes) ---------------------------------------------------------------------------------------
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int j = it * 16 + (lane >> 2), dc = lane & 3;
      const int pos = ST::slot(j);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const bf16_t hi = (bf16_t)vraw[it][e];
        vh[(dc * 8 + e) * VP + pos] = hi;
        vl[(dc * 8 + e) * VP + pos] = (bf16_t)(vraw[it][e] - (float)hi);
      }
    }
    int krid[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int j = (k >> 2) * 16 + g * 4 + (k & 3);   // acc[r] of key tile kt is key kt * 16 + 4 g + r
      krid[k] = 0;
      if (edge) krid[k] = from_lane(my_rid, j < WT ? j : WT - 1);
    }

    f32x4 oacc[2][4];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4) oacc[dt][t4] = f32x4{0.f, 0.f, 0.f, 0.f};

    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wave's V^T stores have landed (own data only)
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) {
      const int rid_i = qrid[t4];
      float sc[16];
      float mx = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        f32x4 st = bias_all[t4][kt];
        MM::mma(st, kl[kt], qh[t4]);
        MM::mma(st, kh[kt], ql[t4]);
        MM::mma(st, kh[kt], qh[t4]);
#pragma unroll
        for (int r = 0; r < 4; ++r) sc[kt * 4 + r] = st[r] * scale2;   // (q.k + bias / scale) * scale * log2 e; padding keys are -inf through the seed
      }
      if (edge) {   // wave-uniform: the SW-MSA mask costs the interior windows no instruction
#pragma unroll
        for (int k = 0; k < 16; ++k) sc[k] += krid[k] != rid_i ? -100.0f * 1.4426950408889634f : 0.f;
      }
#pragma unroll
      for (int k = 0; k < 16; ++k) mx = fmaxf(mx, sc[k]);
      mx = quad_group_max(mx);
      float l = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) { sc[k] = __builtin_amdgcn_exp2f(sc[k] - mx); l += sc[k]; }
      l = quad_group_sum(l);
      const float inv = 1.0f / l;
#pragma unroll
      for (int k = 0; k < 16; ++k) sc[k] *= inv;
#pragma unroll
      for (int ps = 0; ps < 2; ++ps) {   // 32 keys per step
        frag ph, pl;
        split8(sc + ps * 8, ph, pl);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const int off = (dt * 16 + li) * VP + ps * 32 + g * 8;
          const frag vfh = *reinterpret_cast<const frag*>(vh + off);
          const frag vfl = *reinterpret_cast<const frag*>(vl + off);
          MM::mma(oacc[dt][t4], vfl, ph);
          MM::mma(oacc[dt][t4], vfh, pl);
          MM::mma(oacc[dt][t4], vfh, ph);
        }
      }
    }

    // ---- store: acc[r] <-> (dim = dt*16 + 4g + r, query = t4*16 + li) -----------------------------------
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) {
      if (qtok[t4] >= 0) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const f32x4 o = oacc[dt][t4];
          if (out_split) {
            bf16_t* ds = reinterpret_cast<bf16_t*>(out) + (uint64_t)(uint32_t)qtok[t4] * (uint32_t)(2 * C) + (head * HD + g * 4 + dt * 16);
            bf16x4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) { hi[e] = (bf16_t)o[e]; lo[e] = (bf16_t)(o[e] - (float)hi[e]); }
            *reinterpret_cast<bf16x4*>(ds) = hi;
            *reinterpret_cast<bf16x4*>(ds + C) = lo;
          } else {
            *reinterpret_cast<f32x4*>(out + (uint64_t)(uint32_t)qtok[t4] * (uint32_t)C + (head * HD + g * 4 + dt * 16)) = o;
          }
        }
      }
    }
  }   // windows of this wave
}

// relative_position_bias_table [169, nH] -> per head [64 queries][64 keys] fp32: bias[(dy + 6) * 13 + (dx + 6)] / scale for
// real tokens, -inf for the 15 padding key slots, 0 for padding query rows (never stored)
__global__ __launch_bounds__(256) void swin_expand_bias_kernel(const float* __restrict__ table, float* __restrict__ out, int nH) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= nH * 4096) return;
  const int j = idx & 63, i = (idx >> 6) & 63, h = idx >> 12;
  float v;
  if (j >= WT) v = -INFINITY;
  else if (i >= WT) v = 0.f;
  else {
    const int iy = i / WS, ix = i % WS, jy = j / WS, jx = j % WS;
    v = table[((iy - jy + WS - 1) * (2 * WS - 1) + (ix - jx + WS - 1)) * nH + h] / 0.17677669529663687f;
  }
  out[idx] = v;
}

}  // namespace

extern "C" int omp_swin_expand_bias(const float* rel_bias_table, int nH, float* out, omp_stream_t s) {
  OMP_CHECK_ARG(rel_bias_table && out && nH > 0, "omp_swin_expand_bias: bad arguments");
  hipLaunchKernelGGL(swin_expand_bias_kernel, dim3((nH * 4096 + 255) / 256), dim3(256), 0, (hipStream_t)s, rel_bias_table, out, nH);
  OMP_CHECK_LAUNCH("omp_swin_expand_bias");
  return OMP_OK;
}

namespace {
int device_cus() {   // compute units of the current device (cached per device)
  static std::atomic<int> cus[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  int v = cus[dev].load(std::memory_order_relaxed);
  if (v == 0) {
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    cus[dev].store(v, std::memory_order_relaxed);
  }
  return v;
}
}  // namespace

extern "C" int omp_swin_window_attn(const void* qkv, const float* qkv_bias, const float* rel_bias_table,
                                    void* out, int dtype, int B, int H, int W, int C, int nH, int window,
                                    int shift, omp_stream_t s) {
  return omp_swin_window_attn2(qkv, qkv_bias, rel_bias_table, nullptr, out, dtype, dtype, B, H, W, C, nH, window, shift, s);
}

extern "C" int omp_swin_window_attn2(const void* qkv, const float* qkv_bias, const float* rel_bias_table,
                                     const float* bias_expanded, void* out, int dtype, int out_dtype, int B, int H, int W, int C, int nH,
                                     int window, int shift, omp_stream_t s) {
  OMP_CHECK_ARG(out_dtype == dtype || (out_dtype == OMP_BF16X2 && dtype == OMP_F32),
                "omp_swin_window_attn: out_dtype must equal dtype (or be split-bf16 pairs for fp32 inputs)");
  const int out_split = out_dtype == OMP_BF16X2 ? 1 : 0;
  OMP_CHECK_ARG(!out_split || omp_cur().swin_impl != 1, "omp_swin_window_attn: the scalar cross-check kernel has no split-bf16 output");
  OMP_CHECK_ARG(qkv && qkv_bias && (rel_bias_table || bias_expanded) && out, "omp_swin_window_attn: null pointer");
  OMP_CHECK_ARG(window == WS, "omp_swin_window_attn: only window 7 is built (got %d)", window);
  OMP_CHECK_ARG(shift >= 0 && shift < WS, "omp_swin_window_attn: bad shift %d", shift);
  OMP_CHECK_ARG(nH > 0 && C == nH * HD, "omp_swin_window_attn: head_dim must be 32 (C=%d nH=%d)", C, nH);
  OMP_CHECK_ARG(B > 0 && H > 0 && W > 0, "omp_swin_window_attn: bad shape");
  const int nWy = (H + WS - 1) / WS, nWx = (W + WS - 1) / WS;
  OMP_CHECK_ARG((int64_t)B * nWy * nWx < (1ll << 31) && (int64_t)B * H * W < (1ll << 31), "omp_swin_window_attn: more than 2^31 windows / tokens");
  dim3 grid((unsigned)((int64_t)B * nWy * nWx), (unsigned)((nH + 3) / 4));
  if (dtype != OMP_F32 && dtype != OMP_BF16) { omp_set_error("omp_swin_window_attn: bad dtype %d", dtype); return OMP_ERR_INVALID; }
  const int swin_impl = omp_cur().swin_impl;
  // swin_attn_mfma_kernel / swin_attn_x3_kernel walk windows: as many workgroups as are co-resident (two per CU: their launch bounds)
  dim3 pgrid = grid;
  {
    const int per_cu = 2;
    const int64_t co = (int64_t)device_cus() * per_cu / grid.y;
    if (co >= 1 && co < (int64_t)grid.x) pgrid.x = (unsigned)co;
  }
  if (swin_impl == 1) {
    if (dtype == OMP_F32)
      hipLaunchKernelGGL((swin_attn_kernel<float>), grid, dim3(256), 0, (hipStream_t)s, (const float*)qkv,
                         qkv_bias, rel_bias_table, (float*)out, B, H, W, C, nH, shift, nWy, nWx);
    else
      hipLaunchKernelGGL((swin_attn_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)s, (const bf16_t*)qkv,
                         qkv_bias, rel_bias_table, (bf16_t*)out, B, H, W, C, nH, shift, nWy, nWx);
  } else {
    const bool expb = bias_expanded != nullptr && swin_impl != 2;
    OMP_CHECK_ARG(expb || rel_bias_table != nullptr, "omp_swin_window_attn: the table form of the bias is needed for this path");
    const float* tb = expb ? bias_expanded : rel_bias_table;
    if (dtype == OMP_F32 && expb && out_split && swin_impl != 3) {
      // the parity engine's call (fp32 qkv from a bf16x3 product, split-pair rows for the next one): three bf16 MFMAs per product
      // instead of fp32 ones (swin_attn_x3_kernel); selector 3 keeps the fp32 matrix-core kernel for A/B
      hipLaunchKernelGGL(swin_attn_x3_kernel, pgrid, dim3(256), 0, (hipStream_t)s, (const float*)qkv, qkv_bias, tb, (float*)out, B, H, W, C, nH, shift, nWy, nWx, 1);
    } else if (dtype == OMP_F32 && expb && swin_impl == 4) {   // development: the same kernel with fp32 rows out (tests compare it with the reference)
      hipLaunchKernelGGL(swin_attn_x3_kernel, pgrid, dim3(256), 0, (hipStream_t)s, (const float*)qkv, qkv_bias, tb, (float*)out, B, H, W, C, nH, shift, nWy, nWx, out_split);
    } else if (dtype == OMP_F32) {
      if (expb) hipLaunchKernelGGL((swin_attn_mfma_kernel<float, true>), pgrid, dim3(256), 0, (hipStream_t)s, (const float*)qkv, qkv_bias, tb, (float*)out, B, H, W, C, nH, shift, nWy, nWx, out_split, (unsigned long long*)nullptr);
      else hipLaunchKernelGGL((swin_attn_mfma_kernel<float, false>), pgrid, dim3(256), 0, (hipStream_t)s, (const float*)qkv, qkv_bias, tb, (float*)out, B, H, W, C, nH, shift, nWy, nWx, out_split, (unsigned long long*)nullptr);
    } else {
      unsigned long long* tr = omp_cur().mlp_trace;   // development: the traced instantiation writes [workgroup][8] phase cycles of wave 0
      if (expb && tr != nullptr) hipLaunchKernelGGL((swin_attn_mfma_kernel<bf16_t, true, true>), pgrid, dim3(256), 0, (hipStream_t)s, (const bf16_t*)qkv, qkv_bias, tb, (bf16_t*)out, B, H, W, C, nH, shift, nWy, nWx, 0, tr);
      else if (expb) hipLaunchKernelGGL((swin_attn_mfma_kernel<bf16_t, true>), pgrid, dim3(256), 0, (hipStream_t)s, (const bf16_t*)qkv, qkv_bias, tb, (bf16_t*)out, B, H, W, C, nH, shift, nWy, nWx, 0, (unsigned long long*)nullptr);
      else hipLaunchKernelGGL((swin_attn_mfma_kernel<bf16_t, false>), pgrid, dim3(256), 0, (hipStream_t)s, (const bf16_t*)qkv, qkv_bias, tb, (bf16_t*)out, B, H, W, C, nH, shift, nWy, nWx, 0, (unsigned long long*)nullptr);
    }
  }
  OMP_CHECK_LAUNCH("omp_swin_window_attn");
  return OMP_OK;
}

extern "C" int omp_debug_swin_attn_impl(int which) {
  omp_cur().swin_impl = (which >= 1 && which <= 4) ? which : 0;   // 3: fp32 matrix cores also for split-pair output, 4: the split-product kernel also for fp32 output
  return OMP_OK;
}
