// GEMM with fused epilogue for gfx950:  C = act(A @ W^T + bias) + residual
//
// Kernels, all on the 16x16 MFMA (bf16: v_mfma_f32_16x16x32_bf16, f32: v_mfma_f32_16x16x4_f32):
//
//  gemm_dma<T,TOut,BM,BN,NS>   LDS-tiled, 4 waves (2x2), operands global -> LDS by DMA into an NS-stage ring,
//                              one raw barrier per 128-byte K tile, epilogue through LDS.  128x128x2 for the
//                              big Swin / FPN / K-V projection GEMMs (M = tokens), 64x64 rings for mid sizes.
//  gemm_rows<T,TOut>           no LDS; each wave streams a 16-row slab of W straight into MFMA
//                              fragments (decoder GEMMs with <= 64 rows, no LayerNorm prologue).
//  gemm_small<T,TOut,MF,LN>    split-K over the waves of a workgroup for <= 64 rows with the preceding
//                              LayerNorm fused into the prologue (decoder steps).
//  (mlp.hip holds the fused fc1 + GELU + fc2 kernel of the Swin MLP.)
//
// Operand orientation: the matrix core computes D[i][j] with i = output feature n (A operand =
// rows of W) and j = token m (B operand = rows of A).  A lane then owns 4 CONSECUTIVE output
// features of one token (acc[r] <-> n = 4*(lane>>4)+r, m = lane&15), so bias loads and C stores
// are 8/16-byte vectors instead of 4 scalar stores.
#include <type_traits>

#include <atomic>
#include "common.h"

namespace {


struct GemmP {
  const void* A; int64_t lda;
  const void* W; int64_t ldw;
  const float* bias; const int32_t* bias_row; int64_t bias_row_stride;
  const void* residual; int64_t ldr;
  void* C; int64_t ldc;
  int64_t M; int N; int K;
  int act; int trans_out; int64_t trans_rows, trans_ld;
  int tiles_m, tiles_n; int small_hint;
  const float* ln_g; const float* ln_b; float ln_eps;   // optional LayerNorm prologue (A is fp32)
  int store_mode; int bias_m;                           // OMP_STORE_*; bias indexed by m instead of n
  int kv_B, kv_tok, kv_mpad, kv_nH, kv_kb;              // blocked K / V^T destination geometry
  void* C2; int64_t ldc2;                               // optional second destination without the residual
  int a_wrap;                                           // > 0: A rows are split-bf16 pairs [hi | lo] of a_wrap elements, K columns beyond wrap back (bf16x3)
  int split_out;                                        // bf16 destination written as split pairs: hi at column n, lo at column N + n
  unsigned long long* trace;                            // debug: per-workgroup phase timestamps (gemm_dma<..., TRACE>)
};

// bf16 destinations take the bf16 engine's GELU everywhere (vectorised or not: a value must not depend on which
// kernel or epilogue path produced it), fp32 destinations the < 1 ulp erf form
template <typename TOut>
__device__ __forceinline__ float apply_act(float v, int act, bool precise = false) {
  if (act == OMP_ACT_GELU) {
    if constexpr (std::is_same<TOut, bf16_t>::value) return precise ? gelu_erf(v) : gelu_fast2(f32x2{v, 0.0f})[0];
    else return gelu_erf(v);
  }
  if (act == OMP_ACT_RELU) return fmaxf(v, 0.0f);
  return v;
}

// Store 4 consecutive-n values v (bias and activation already applied) of token m: adds the residual and
// honours the destination layout (plain / transposed / blocked K / blocked V^T).
template <typename TOut>
__device__ __forceinline__ void store4(const GemmP& p, int64_t m, int n, const float* vin) {
  if (m >= p.M || n >= p.N) return;
  const TOut* res = reinterpret_cast<const TOut*>(p.residual);
  TOut* C = reinterpret_cast<TOut*>(p.C);
  float v[4] = {vin[0], vin[1], vin[2], vin[3]};
  if (p.store_mode == OMP_STORE_KBLK) {
    // m = memory token (image b, key ml), n..n+3 = 4 dims of one head of one (decoder, layer) slab:
    // K slab [nl][b][h][Mpad][64]
    const int d = p.kv_nH * 64;
    const int b = (int)(m / p.kv_tok), ml = (int)(m % p.kv_tok);
    const int nl = n / d, h = (n % d) >> 6, dd = n & 63;
    if constexpr (sizeof(TOut) == 2) {
      if (p.split_out) {   // split planes: block ml / 32 = [hi plane 32 x 64 | lo plane]
        TOut* dst = C + (((((int64_t)nl * p.kv_B + b) * p.kv_nH + h) * (p.kv_mpad >> 5) + (ml >> 5)) * 2) * 2048 + (ml & 31) * 64 + dd;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n + r < p.N) {
            const bf16_t hi = (bf16_t)v[r];
            dst[r] = hi;
            dst[2048 + r] = (bf16_t)(v[r] - (float)hi);
          }
        return;
      }
    }
    TOut* dst = C + ((((int64_t)nl * p.kv_B + b) * p.kv_nH + h) * p.kv_mpad + ml) * 64 + dd;
    if (n + 3 < p.N) {
      if constexpr (sizeof(TOut) == 4) *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
      else *reinterpret_cast<bf16x4*>(dst) = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (n + r < p.N) dst[r] = from_f32<TOut>(v[r]);
    }
    return;
  }
  if (p.store_mode == OMP_STORE_VBLK) {
    // swapped operands: m = value feature (decoder-layer nl, head h, dim dd), n..n+3 = 4 memory tokens.
    // V^T slab [nl][b][h][Mpad/KB][64][KB]; inside a block the KB keys sit in the order the PV matrix-core
    // product consumes them (bf16: slot 8g + 4*half + r <-> key 16*half + 4g + r; f32: natural order).
    const int d = p.kv_nH * 64, KB = p.kv_kb;
    const int nl = (int)(m / d), h = (int)(m % d) >> 6, dd = (int)m & 63;
    const int PLN = (sizeof(TOut) == 2 && p.split_out) ? 2 : 1;   // split planes: a block is [hi plane 64 x KB | lo plane]
    auto slot = [&](int tok, int& b) -> int64_t {
      b = tok / p.kv_tok;
      const int ml = tok - b * p.kv_tok;
      const int blk = ml / KB, kl = ml - blk * KB;
      const int pos = (KB == 32) ? (((kl & 15) >> 2) * 8 + (kl >> 4) * 4 + (kl & 3)) : kl;
      return ((((int64_t)nl * p.kv_B + b) * p.kv_nH + h) * (p.kv_mpad / KB) + blk) * (64 * KB * PLN) + dd * KB + pos;
    };
    // element by element (split planes; tokens per image not a multiple of 4: MGP-STR's 257): ONE division for the 4 tokens -- (image, key) of the
    // first, then walked -- instead of three per element (round 6: the V^T projection of a ViT block spent 540 us here against 200 for the product)
    auto walk = [&](auto&& put) {
      int b = n / p.kv_tok, ml = n - b * p.kv_tok;
      const int64_t per_head = (int64_t)(p.kv_mpad / KB) * (64 * KB * PLN);
      int64_t base = (((int64_t)nl * p.kv_B + b) * p.kv_nH + h) * per_head + dd * KB;
      const int sh = KB == 32 ? 5 : 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (n + r < p.N) {
          const int blk = ml >> sh, kl = ml & (KB - 1);
          const int pos = (KB == 32) ? (((kl & 15) >> 2) * 8 + (kl >> 4) * 4 + (kl & 3)) : kl;
          put(base + (int64_t)blk * (64 * KB * PLN) + pos, r);
        }
        if (++ml == p.kv_tok) { ml = 0; base += (int64_t)p.kv_nH * per_head; }
      }
    };
    if constexpr (sizeof(TOut) == 2) {
      if (p.split_out) {
        walk([&](int64_t i, int r) {
          const bf16_t hi = (bf16_t)v[r];
          C[i] = hi;
          C[i + 64 * KB] = (bf16_t)(v[r] - (float)hi);
        });
        return;
      }
    }
    if (n + 3 < p.N && (p.kv_tok & 3) == 0) {   // 4 tokens of one image, contiguous slots
      int b0;
      const int64_t i0 = slot(n, b0);
      if constexpr (sizeof(TOut) == 4) *reinterpret_cast<f32x4*>(C + i0) = f32x4{v[0], v[1], v[2], v[3]};
      else *reinterpret_cast<bf16x4*>(C + i0) = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
    } else {
      walk([&](int64_t i, int r) { C[i] = from_f32<TOut>(v[r]); });
    }
    return;
  }
  if (p.trans_out) {
    int64_t bidx = m / p.trans_rows, mi = m % p.trans_rows;
    TOut* base = C + bidx * (int64_t)p.N * p.trans_ld + mi;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (n + r < p.N) base[(int64_t)(n + r) * p.trans_ld] = from_f32<TOut>(v[r]);
    return;
  }
  if constexpr (sizeof(TOut) == 2) {
    if (p.split_out) {   // split-bf16 pair rows (no residual: checked on the host)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (n + r < p.N) {
          const bf16_t hi = (bf16_t)v[r];
          C[m * p.ldc + n + r] = hi;
          C[m * p.ldc + p.N + n + r] = (bf16_t)(v[r] - (float)hi);
        }
      return;
    }
  }
  const bool full = (n + 3 < p.N) && ((p.ldc & 3) == 0) && (res == nullptr || (p.ldr & 3) == 0);
  if (full) {
    if (res != nullptr) {
      if constexpr (sizeof(TOut) == 4) {
        f32x4 rv = *reinterpret_cast<const f32x4*>(res + m * p.ldr + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += rv[r];
      } else {
        bf16x4 rv = *reinterpret_cast<const bf16x4*>(res + m * p.ldr + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += (float)rv[r];
      }
    }
    if constexpr (sizeof(TOut) == 4) {
      f32x4 o = {v[0], v[1], v[2], v[3]};
      *reinterpret_cast<f32x4*>(C + m * p.ldc + n) = o;
    } else {
      bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
      *reinterpret_cast<bf16x4*>(C + m * p.ldc + n) = o;
    }
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (n + r < p.N) {
        float o = v[r];
        if (res != nullptr) o += to_f32(res[m * p.ldr + n + r]);
        C[m * p.ldc + n + r] = from_f32<TOut>(o);
      }
    }
  }
}

// bias + activation of the 4 consecutive-n values a lane holds for token m, then store4.
template <typename TOut>
__device__ __forceinline__ void epilogue_store(const GemmP& p, const float* bias, int64_t m, int n,
                                               f32x4 acc) {
  if (m >= p.M || n >= p.N) return;
  float v[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float b = 0.0f;
    if (bias != nullptr) b = p.bias_m ? bias[m] : (n + r < p.N ? bias[n + r] : 0.0f);
    v[r] = apply_act<TOut>(acc[r] + b, p.act);
  }
  store4<TOut>(p, m, n, v);
}

// bijective XCD remap: consecutive logical tile ids land on the same XCD (block b runs on XCD b%8)
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + idx;
}

// ---------------------------------------------------------------------------------------------
// gemm_dma<T,TOut,BM,BN>: the large-M GEMM (Swin / FPN / projection / K-V slabs).
//   * operand tiles go global -> LDS by DMA (global_load_lds_dwordx4, no staging registers); the LDS image
//     is lane-linear per wave instruction (8 rows x 128 B), so the XOR swizzle that makes the fragment
//     ds_read_b128 conflict-free is applied to the per-lane SOURCE address (chunk c of row r lives in
//     slot c ^ (r & 7));
//   * two LDS stages, ONE raw s_barrier per K tile: wait own DMA of tile t -> barrier -> issue DMA of tile
//     t+1 into the other stage -> MFMAs of tile t (the next tile's loads fly under them);
//   * epilogue through LDS: accumulators (+bias, activation) are written as fp32 rows, read back row-
//     contiguous and leave as 16-byte stores of full 128/256-byte row segments, the residual arriving the
//     same way -- instead of 8-byte stores scattered over 16 rows per instruction.
// ---------------------------------------------------------------------------------------------
// wait until at most `tiles` of this wave's most recently issued K tiles (IPT DMA instructions each) are
// still in flight; the count must be an immediate, hence the switch (wave-uniform, so one scalar branch)
template <int IPT>
__device__ __forceinline__ void wait_dma_tiles(int tiles) {
  static_assert(IPT * 7 <= 63, "vmcnt immediate out of range");
  switch (tiles) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPT * 1) : "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPT * 2) : "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPT * 3) : "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPT * 4) : "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPT * 5) : "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPT * 6) : "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPT * 7) : "memory"); break;
  }
}

template <typename T, typename TOut, int BM, int BN, int NS, bool TRACE = false>
__global__ __launch_bounds__(256) void gemm_dma(GemmP p) {
  static_assert(NS >= 2 && NS <= 8, "2..8 LDS stages");
  // TRACE (selector 15, development only): wave 0 / lane 0 stamps s_memtime at the phase boundaries into p.trace[blockIdx][8]
  unsigned long long tr[6];
  auto stamp = [&](int i) {
    if constexpr (TRACE) tr[i] = __builtin_amdgcn_s_memtime();
  };
  stamp(0);
  typedef Mma<T> MM;
  typedef typename MM::frag frag;
  constexpr int ROWB = 128;                       // bytes of K per LDS row
  constexpr int KT = ROWB / (int)sizeof(T);       // k elements per tile
  constexpr int STEPS = KT / MM::KSTEP;           // 2
  constexpr int EPC = 16 / (int)sizeof(T);        // elements per 16-byte chunk
  constexpr int FM = BM / 32, FN = BN / 32;       // frags per wave (wave tile = BM/2 x BN/2)
  constexpr int AI = BM / 32, WI = BN / 32;       // DMA instructions per wave per tile (8 rows each)
  constexpr int STAGE = (BM + BN) * ROWB;
  constexpr int ES = BN + 4;                      // epilogue row pitch in floats (bank spread)
  extern __shared__ __attribute__((aligned(16))) char smem[];   // max(NS * STAGE, BM * ES * 4)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int nwg = p.tiles_m * p.tiles_n;
  const int lid = xcd_remap(blockIdx.x, nwg);
  const int tm = lid / p.tiles_n, tn = lid % p.tiles_n;
  const int64_t m0 = (int64_t)tm * BM;
  const int n0 = tn * BN;

  // DMA source pointers: lane l of a wave instruction fills slot (l & 7) of row (l >> 3) of an 8-row piece
  const int lr = lane >> 3, lc = (lane & 7) ^ lr;
  const T* a_src[AI];
  const T* w_src[WI];
#pragma unroll
  for (int j = 0; j < AI; ++j) {
    int64_t gm = m0 + wave * (BM / 4) + j * 8 + lr; if (gm > p.M - 1) gm = p.M - 1;
    a_src[j] = reinterpret_cast<const T*>(p.A) + gm * p.lda + lc * EPC;
  }
#pragma unroll
  for (int j = 0; j < WI; ++j) {
    int gn = n0 + wave * (BN / 4) + j * 8 + lr; if (gn > p.N - 1) gn = p.N - 1;
    w_src[j] = reinterpret_cast<const T*>(p.W) + (int64_t)gn * p.ldw + lc * EPC;
  }
  auto issue = [&](int kt, int buf) {
    const int koff = kt * KT;
    int kaoff = koff;
    if (p.a_wrap > 0 && kaoff >= p.a_wrap) kaoff -= p.a_wrap;   // bf16x3: [hi | lo] rows read as [hi | lo | hi]
    char* abase = smem + buf * STAGE + (wave * (BM / 4)) * ROWB;
    char* wbase = smem + buf * STAGE + BM * ROWB + (wave * (BN / 4)) * ROWB;
#pragma unroll
    for (int j = 0; j < AI; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[j] + kaoff),
                                       (__attribute__((address_space(3))) void*)(abase + j * 8 * ROWB), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < WI; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_src[j] + koff),
                                       (__attribute__((address_space(3))) void*)(wbase + j * 8 * ROWB), 16, 0, 0);
  };

  // this thread's role on the way OUT (epilogue): column chunk cidx of rows rsub, rsub + RPP, ...; its bias
  // values are requested here, before the first DMA, so they have long arrived when the K loop ends
  constexpr int CH = 16 / (int)sizeof(TOut);   // output elements per 16-byte chunk
  constexpr int CPR = BN / CH;                 // chunks per tile row
  constexpr int RPP = 256 / CPR;               // rows per pass
  const TOut* res = reinterpret_cast<const TOut*>(p.residual);
  TOut* C = reinterpret_cast<TOut*>(p.C);
  const bool vec_ok = p.store_mode == OMP_STORE_PLAIN && !p.trans_out && (p.ldc % CH) == 0 &&
                      (res == nullptr || (p.ldr % CH) == 0);
  const int cidx = tid % CPR, rsub = tid / CPR;
  const int n = n0 + cidx * CH;
  const bool vec_path = vec_ok && n + CH <= p.N;
  const float* bias = p.bias;
  if (bias != nullptr && p.bias_row != nullptr) bias += (int64_t)(*p.bias_row) * p.bias_row_stride;
  // unconditional 16-byte loads (a branch here would make the compiler wait for them on the spot): threads
  // without a vector-loadable column bias read the first bytes of W instead and discard them
  const bool bias_vec = bias != nullptr && !p.bias_m && n + CH <= p.N && (reinterpret_cast<uintptr_t>(bias) & 15) == 0;
  const float* bsrc = bias_vec ? bias + n : reinterpret_cast<const float*>(p.W);
  f32x4 braw[CH / 4];
#pragma unroll
  for (int q = 0; q < CH / 4; ++q) braw[q] = *reinterpret_cast<const f32x4*>(bsrc + 4 * q);

  f32x4 acc[FN][FM];
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int lrow = lane & 15, lg = lane >> 4;
  auto compute = [&](int buf) {
    const char* as = smem + buf * STAGE + (wm * (BM / 2)) * ROWB;
    const char* ws = smem + buf * STAGE + BM * ROWB + (wn * (BN / 2)) * ROWB;
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      frag fw[FN], fx[FM];
      const int c = s * 4 + lg;
#pragma unroll
      for (int i = 0; i < FN; ++i) {
        const int row = i * 16 + lrow;   // (row & 7) == (lrow & 7): fragment tiles are 16-row aligned
        fw[i] = *reinterpret_cast<const frag*>(ws + row * ROWB + ((c ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < FM; ++j) {
        const int row = j * 16 + lrow;
        fx[j] = *reinterpret_cast<const frag*>(as + row * ROWB + ((c ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) MM::mma(acc[i][j], fw[i], fx[j]);
    }
  };

  // Ring of NS stages.  Prologue: tiles 0..NS-2 in flight.  Iteration kt: wait for this wave's DMA of tile kt
  // (everything issued after it may stay in flight: min(nk-1-kt, NS-2) tiles), barrier -- now everybody's
  // part of tile kt has landed AND everybody is done reading tile kt-1, whose stage (kt-1) % NS =
  // (kt+NS-1) % NS is refilled with tile kt+NS-1 while tile kt is multiplied.  NS = 2 is the plain double
  // buffer; with NS*KT >= K (decoder GEMMs, K = 512) the whole K extent is requested up front and the kernel
  // costs one memory round trip instead of one per K tile.
  const int nk = p.K / KT;
#pragma unroll
  for (int t = 0; t < NS - 1; ++t)
    if (t < nk) issue(t, t);
  int st_c = 0, st_i = NS - 1;   // stage of tile kt / of tile kt+NS-1
  for (int kt = 0; kt < nk; ++kt) {
    const int after = nk - 1 - kt;
    wait_dma_tiles<AI + WI>(after < NS - 2 ? after : NS - 2);
    __builtin_amdgcn_s_barrier();
    if (kt == 0) stamp(1);                       // first K tile landed everywhere
    if (kt + NS - 1 < nk) issue(kt + NS - 1, st_i);
    compute(st_c);
    st_c = (st_c + 1 == NS) ? 0 : st_c + 1;
    st_i = (st_i + 1 == NS) ? 0 : st_i + 1;
  }
  stamp(2);                                      // K loop done (this wave)

  // ---- epilogue ------------------------------------------------------------------------------------------
  // The accumulators go through LDS RAW; bias, activation and residual are applied on the way out, where a thread
  // owns one 16-byte column chunk of RPP-strided rows: its CH bias values were requested before the K loop
  // (round 1 loaded them per fragment between the K loop and the LDS pass -- four serialised memory round trips,
  // 37 % of a workgroup's life in the stage-2 QKV GEMM, profiles/r02a_gemm_trace.txt), the activation is a
  // compile-time branch around the whole store loop, and the residual rows are requested before the LDS pass so
  // that their round trip runs under it.
  float bcol[CH];
#pragma unroll
  for (int q = 0; q < CH; ++q) bcol[q] = bias_vec ? braw[q >> 2][q & 3] : 0.f;
  if (bias != nullptr && !p.bias_m && !bias_vec) {   // ragged N edge
#pragma unroll
    for (int q = 0; q < CH; ++q)
      if (n + q < p.N) bcol[q] = bias[n + q];
  }
  typename Vec16<TOut>::type rres[BM / RPP];
  if (vec_path && res != nullptr) {
#pragma unroll
    for (int pass = 0; pass < BM / RPP; ++pass) {
      int64_t m = m0 + pass * RPP + rsub;
      if (m > p.M - 1) m = p.M - 1;            // clamped rows are loaded but never stored
      rres[pass] = *reinterpret_cast<const typename Vec16<TOut>::type*>(res + m * p.ldr + n);
    }
  }
  __builtin_amdgcn_s_barrier();   // all waves are done with the operand stages
  float* E = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < FN; ++i) {
    const int nl = wn * (BN / 2) + i * 16 + lg * 4;
#pragma unroll
    for (int j = 0; j < FM; ++j) {
      const int ml = wm * (BM / 2) + j * 16 + lrow;
      *reinterpret_cast<f32x4*>(E + ml * ES + nl) = acc[i][j];
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // LDS writes only: the residual loads keep flying
  __builtin_amdgcn_s_barrier();
  stamp(3);                                      // accumulators of every wave are in LDS
  if (p.store_mode == OMP_STORE_ROWSTAT) {
    // greedy decoding without the logits (include/omp355.h, omp_row_stat_merge): per row and 64-column HALF of the tile the maximum of logit + bias,
    // its column and the sum of exp(logit - maximum) over the valid columns.  A thread owns one row's half (two serial passes over 64 LDS values: no
    // cross-lane reduction chains -- a first version reduced a row over 32 lanes with ten dependent shuffles per pass and took longer than the
    // logits store it replaced); columns beyond N count as -inf, a half without a valid column stores {-inf, 0}.
    if constexpr (std::is_same<TOut, float>::value && BM == 128 && BN == 128) {
      const int row = tid & 127, half = tid >> 7;
      const float* er = E + row * ES + half * 64;
      const int nc0 = n0 + half * 64;
      const bool bvec = bias != nullptr && !p.bias_m && (reinterpret_cast<uintptr_t>(bias) & 15) == 0 && nc0 + 64 <= p.N;
      auto col = [&](int q, float (&v)[4]) {   // columns nc0 + 4 q .. + 3 of this row: logit + bias, -inf beyond N
        const f32x4 t = *reinterpret_cast<const f32x4*>(er + 4 * q);
        if (bvec) {
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + nc0 + 4 * q);
#pragma unroll
          for (int u = 0; u < 4; ++u) v[u] = t[u] + b4[u];
        } else {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int c = nc0 + 4 * q + u;
            v[u] = c < p.N ? t[u] + ((bias != nullptr && !p.bias_m) ? bias[c] : 0.f) : -INFINITY;
          }
        }
      };
      float best = -INFINITY;
      int bi = 0x7fffffff;
#pragma unroll 4
      for (int q = 0; q < 16; ++q) {
        float v[4];
        col(q, v);
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (v[u] > best) { best = v[u]; bi = nc0 + 4 * q + u; }   // ascending columns, strict >: the lowest index wins a tie
      }
      float sum = 0.f;
      if (best > -INFINITY) {
#pragma unroll 4
        for (int q = 0; q < 16; ++q) {
          float v[4];
          col(q, v);
#pragma unroll
          for (int u = 0; u < 4; ++u) sum += expf(v[u] - best);   // exp(-inf) = 0
        }
      }
      if (m0 + row < p.M) reinterpret_cast<f32x4*>(p.C)[((int64_t)(m0 + row) * p.tiles_n + tn) * 2 + half] = f32x4{best, sum, __int_as_float(bi), 0.f};
    }
    return;
  }
  if (n >= p.N) return;
  if (vec_path) {
    auto store_rows = [&](auto ACT) {
#pragma unroll
      for (int pass = 0; pass < BM / RPP; ++pass) {
        const int r = pass * RPP + rsub;
        const int64_t m = m0 + r;
        if (m < p.M) {
          float v[CH];
#pragma unroll
          for (int q = 0; q < CH; q += 4) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(E + r * ES + cidx * CH + q);
            v[q] = t[0]; v[q + 1] = t[1]; v[q + 2] = t[2]; v[q + 3] = t[3];
          }
#pragma unroll
          for (int q = 0; q < CH; ++q) {
            v[q] += bcol[q];
            if constexpr (decltype(ACT)::value == OMP_ACT_GELU && !std::is_same<TOut, bf16_t>::value) v[q] = gelu_erf(v[q]);
            if constexpr (decltype(ACT)::value == OMP_ACT_RELU) v[q] = fmaxf(v[q], 0.0f);
          }
          if constexpr (decltype(ACT)::value == OMP_ACT_GELU && std::is_same<TOut, bf16_t>::value) gelu_fast_n<CH>(v);   // bf16 and split pairs alike
          if constexpr (std::is_same<TOut, bf16_t>::value) {
            if (p.split_out) {
              bf16x8 hi, lo;
#pragma unroll
              for (int q = 0; q < 8; ++q) { hi[q] = (bf16_t)v[q]; lo[q] = (bf16_t)(v[q] - (float)hi[q]); }
              *reinterpret_cast<bf16x8*>(C + m * p.ldc + n) = hi;
              *reinterpret_cast<bf16x8*>(C + m * p.ldc + p.N + n) = lo;
              continue;
            }
          }
          if (res != nullptr) {
            if (p.C2 != nullptr) {   // memory and memory + pos from one product
              typename Vec16<TOut>::type o2;
              pack16(v, o2);
              *reinterpret_cast<typename Vec16<TOut>::type*>(reinterpret_cast<TOut*>(p.C2) + m * p.ldc2 + n) = o2;
            }
            float rv[CH];
            unpack16(rres[pass], rv);
#pragma unroll
            for (int q = 0; q < CH; ++q) v[q] += rv[q];
          }
          typename Vec16<TOut>::type o;
          pack16(v, o);
          *reinterpret_cast<typename Vec16<TOut>::type*>(C + m * p.ldc + n) = o;
        }
      }
    };
    if (p.act == OMP_ACT_GELU) store_rows(std::integral_constant<int, OMP_ACT_GELU>());
    else if (p.act == OMP_ACT_RELU) store_rows(std::integral_constant<int, OMP_ACT_RELU>());
    else store_rows(std::integral_constant<int, OMP_ACT_NONE>());
    if constexpr (TRACE) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // stores retired
      stamp(4);
      if (tid == 0 && p.trace != nullptr) {
        unsigned long long* t = p.trace + (long long)blockIdx.x * 8;
        t[0] = tr[0]; t[1] = tr[1]; t[2] = tr[2]; t[3] = tr[3]; t[4] = tr[4];
        t[5] = (unsigned long long)__builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20);   // XCC_ID (HW_REG 20), low 4 bits
      }
    }
  } else {
    // ragged N edge, odd pitches and the transposed / blocked K / blocked V^T destinations: 4 values at a time
#pragma unroll 1
    for (int it = 0; it < (BM / RPP) * (CH / 4); ++it) {
      const int pass = it / (CH / 4), q = (it % (CH / 4)) * 4;
      const int r = pass * RPP + rsub;
      const f32x4 t = *reinterpret_cast<const f32x4*>(E + r * ES + cidx * CH + q);
      float bm = 0.f;
      if (bias != nullptr && p.bias_m && m0 + r < p.M) bm = bias[m0 + r];
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = apply_act<TOut>(t[u] + bcol[q + u] + bm, p.act);
      store4<TOut>(p, m0 + r, n + q, v);
    }
  }
}

template <typename T, typename TOut, int BM, int BN, int NS, bool TRACE = false>
int launch_dma(GemmP& p, hipStream_t st) {
  constexpr int STAGE = (BM + BN) * 128;
  constexpr int EBYTES = BM * (BN + 4) * 4;
  constexpr size_t smem = (NS * STAGE > EBYTES) ? NS * STAGE : EBYTES;
  auto kern = gemm_dma<T, TOut, BM, BN, NS, TRACE>;
  static bool done = false;   // per template instantiation
  if (!done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
      omp_set_error("omp_gemm_bias_act: cannot raise dynamic LDS limit");
      return OMP_ERR_LAUNCH;
    }
    done = true;
  }
  p.tiles_m = (int)ceil_div64(p.M, BM); p.tiles_n = (int)ceil_div64(p.N, BN);
  hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n), dim3(256), smem, st, p);
  return OMP_OK;
}

#include "gemm256.inc"
#include "gemm4w.inc"
#include "gemm4wr.inc"
#include "gemm4wp.inc"

template <typename T, typename TOut>
__global__ __launch_bounds__(256) void gemm_rows(GemmP p) {
  typedef Mma<T> MM;
  typedef typename MM::frag frag;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lrow = lane & 15, lg = lane >> 4;
  const int nb = blockIdx.x * 64 + wave * 16;
  const int64_t mb = (int64_t)blockIdx.y * 16;
  if (nb >= p.N) return;
  int gn = nb + lrow; if (gn > p.N - 1) gn = p.N - 1;
  int64_t gm = mb + lrow; if (gm > p.M - 1) gm = p.M - 1;
  const T* wp = reinterpret_cast<const T*>(p.W) + (int64_t)gn * p.ldw + lg * MM::KPL;
  const T* xp = reinterpret_cast<const T*>(p.A) + gm * p.lda + lg * MM::KPL;
  // ONE accumulator chain in ascending k: bit-identical to gemm_tiled's per-element summation order,
  // so a row's result does not depend on which kernel (i.e. which batch size) computed it.
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int nsteps = p.K / MM::KSTEP;
  auto xk = [&](int st) -> int {   // A-side K offset of step st (bf16x3: [hi | lo] rows read as [hi | lo | hi])
    const int k = st * MM::KSTEP;
    return (p.a_wrap > 0 && k >= p.a_wrap) ? k - p.a_wrap : k;
  };
  int s = 0;
  for (; s + 8 <= nsteps; s += 8) {
    frag fw[8], fx[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      fw[u] = ld16<T>(wp + (s + u) * MM::KSTEP);
      fx[u] = ld16<T>(xp + xk(s + u));
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) MM::mma(acc, fw[u], fx[u]);
  }
  for (; s < nsteps; ++s) MM::mma(acc, ld16<T>(wp + s * MM::KSTEP), ld16<T>(xp + xk(s)));
  const float* bias = p.bias;
  if (bias != nullptr && p.bias_row != nullptr) bias += (int64_t)(*p.bias_row) * p.bias_row_stride;
  epilogue_store<TOut>(p, bias, mb + lrow, nb + lg * 4, acc);
}


// Small-M split-K kernel (decoder steps: M = rows of one phase <= 64, weight-streaming bound).
// grid = ceil(N/16) workgroups of 4 waves; a workgroup owns 16 output features x ALL rows and its waves
// split K four ways (so even N = 512 gives 32 workgroups x 4 waves, each streaming a 16 x K/4 slab of W
// with a handful of 16-byte loads in flight); partial accumulators meet in LDS in a fixed order.
// LN = true fuses the preceding LayerNorm: A is the fp32 residual stream, each workgroup normalises
// the (<= 64) rows once into LDS (as T) and the MFMA B-operand is read from there.
template <typename T, typename TOut, int MF, bool LN>
__global__ __launch_bounds__(256) void gemm_small(GemmP p) {
  typedef Mma<T> MM;
  typedef typename MM::frag frag;
  extern __shared__ __attribute__((aligned(16))) char sm_small[];
  f32x4* red = reinterpret_cast<f32x4*>(sm_small);           // [4 waves][MF][64 lanes]
  char* alds = sm_small + 4 * MF * 64 * sizeof(f32x4);        // [MF*16 rows][K*sizeof(T) + 16]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lrow = lane & 15, lg = lane >> 4;
  const int nb = blockIdx.x * 16;
  int gn = nb + lrow; if (gn > p.N - 1) gn = p.N - 1;
  const T* wp = reinterpret_cast<const T*>(p.W) + (int64_t)gn * p.ldw + lg * MM::KPL;
  const int KQ = p.K / 4, kbeg = wave * KQ;
  const int astride = p.K * (int)sizeof(T) + 16;
  const int nsteps = KQ / MM::KSTEP;   // 4 (K=512 bf16) .. 16 (K=2048 bf16 / K=1024 f32)
  constexpr int PF = 8;                // W fragments in flight per wave
  // the weight stream does not depend on anything computed here: issue it first so its latency hides
  // under the LayerNorm prologue / activation loads
  frag fw[PF];
#pragma unroll
  for (int u = 0; u < PF; ++u)
    if (u < nsteps) fw[u] = ld16<T>(wp + kbeg + u * MM::KSTEP);

  if constexpr (LN) {
    // LayerNorm prologue: wave w normalises rows w, w+4, ... of the (<= MF*16) rows into LDS.  Rows are taken
    // RB at a time with ALL their loads issued before the first reduction, so a wave pays one memory round
    // trip and RB interleaved shuffle chains per batch instead of one of each per row.
    const float* X = reinterpret_cast<const float*>(p.A);
    const int nch = p.K / 4;  // float4 chunks per row (<= 256)
    constexpr int RB = 4;
    const float invK = 1.0f / (float)p.K;
#pragma unroll 1
    for (int rb = 0; rb < MF * 4; rb += RB) {
      f32x4 v[RB][4];
#pragma unroll
      for (int j = 0; j < RB; ++j) {
        const int r = wave + 4 * (rb + j);
        const float* xr = X + (int64_t)(r < p.M ? r : 0) * p.lda;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int c = lane + 64 * it;
          v[j][it] = (c < nch && r < p.M) ? *reinterpret_cast<const f32x4*>(xr + c * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
      float mean[RB], rstd[RB];
#pragma unroll
      for (int j = 0; j < RB; ++j) {
        float s = 0.f;
#pragma unroll
        for (int it = 0; it < 4; ++it) s += (v[j][it][0] + v[j][it][1]) + (v[j][it][2] + v[j][it][3]);
        mean[j] = s;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int j = 0; j < RB; ++j) mean[j] += __shfl_xor(mean[j], o, 64);
#pragma unroll
      for (int j = 0; j < RB; ++j) {
        mean[j] *= invK;
        float q = 0.f;
#pragma unroll
        for (int it = 0; it < 4; ++it)
          if (lane + 64 * it < nch) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { const float d = v[j][it][i] - mean[j]; q += d * d; }
          }
        rstd[j] = q;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int j = 0; j < RB; ++j) rstd[j] += __shfl_xor(rstd[j], o, 64);
#pragma unroll
      for (int j = 0; j < RB; ++j) {
        const int r = wave + 4 * (rb + j);
        const bool live = r < p.M;
        const float rs = 1.0f / sqrtf(rstd[j] * invK + p.ln_eps);
        char* dst = alds + r * astride;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int c = lane + 64 * it;
          if (c < nch) {
            const f32x4 gg = *reinterpret_cast<const f32x4*>(p.ln_g + c * 4);
            const f32x4 bb = *reinterpret_cast<const f32x4*>(p.ln_b + c * 4);
            T* o = reinterpret_cast<T*>(dst) + c * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = from_f32<T>(live ? (v[j][it][i] - mean[j]) * rs * gg[i] + bb[i] : 0.f);
          }
        }
      }
    }
    __syncthreads();
  }

  const T* xp[MF];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf) {
    int64_t gm = mf * 16 + lrow; if (gm > p.M - 1) gm = p.M - 1;
    xp[mf] = reinterpret_cast<const T*>(p.A) + gm * p.lda + lg * MM::KPL;
  }
  f32x4 acc[MF];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf) acc[mf] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int s0 = 0; s0 < nsteps; s0 += PF) {
    if (s0 > 0) {
#pragma unroll
      for (int u = 0; u < PF; ++u)
        if (s0 + u < nsteps) fw[u] = ld16<T>(wp + kbeg + (s0 + u) * MM::KSTEP);
    }
    frag fx[PF][MF];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      if (s0 + u < nsteps) {
        const int k = kbeg + (s0 + u) * MM::KSTEP;
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
          if constexpr (LN) fx[u][mf] = *reinterpret_cast<const frag*>(alds + (mf * 16 + lrow) * astride + (k + lg * MM::KPL) * (int)sizeof(T));
          else fx[u][mf] = ld16<T>(xp[mf] + k);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      if (s0 + u < nsteps) {
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) MM::mma(acc[mf], fw[u], fx[u][mf]);
      }
    }
  }
#pragma unroll
  for (int mf = 0; mf < MF; ++mf) red[(wave * MF + mf) * 64 + lane] = acc[mf];
  __syncthreads();
  if (wave < MF) {
    f32x4 a = red[(0 * MF + wave) * 64 + lane];
#pragma unroll
    for (int w = 1; w < 4; ++w) a += red[(w * MF + wave) * 64 + lane];
    const float* bias = p.bias;
    if (bias != nullptr && p.bias_row != nullptr) bias += (int64_t)(*p.bias_row) * p.bias_row_stride;
    epilogue_store<TOut>(p, bias, (int64_t)wave * 16 + lrow, nb + lg * 4, a);
  }
}

template <typename T, typename TOut, int MF, bool LN>
int launch_small_t(const GemmP& p, hipStream_t st) {
  const size_t red = 4 * MF * 64 * sizeof(f32x4);
  const size_t ab = LN ? (size_t)MF * 16 * (p.K * sizeof(T) + 16) : 0;
  const size_t smem = red + ab;
  auto kern = gemm_small<T, TOut, MF, LN>;
  if (smem > 48 * 1024) {
    static bool done = false;   // per template instantiation
    if (!done) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
        omp_set_error("omp_gemm_bias_act: cannot raise dynamic LDS limit");
        return OMP_ERR_LAUNCH;
      }
      done = true;
    }
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)ceil_div64(p.N, 16)), dim3(256), smem, st, p);
  OMP_CHECK_LAUNCH("omp_gemm_bias_act(small)");
  return OMP_OK;
}

template <typename T, typename TOut>
int launch_small(const GemmP& p, hipStream_t st) {
  const int mf = (int)ceil_div64(p.M, 16);
  const bool ln = p.ln_g != nullptr;
  if (mf <= 1) return ln ? launch_small_t<T, TOut, 1, true>(p, st) : launch_small_t<T, TOut, 1, false>(p, st);
  if (mf <= 2) return ln ? launch_small_t<T, TOut, 2, true>(p, st) : launch_small_t<T, TOut, 2, false>(p, st);
  return ln ? launch_small_t<T, TOut, 4, true>(p, st) : launch_small_t<T, TOut, 4, false>(p, st);
}

// minimum HBM traffic of one product: A and W once, every destination once, the residual once
inline double gemm_alg_bytes(const GemmP& p, size_t esz, size_t osz) {
  return ((double)p.M * p.K + (double)p.N * p.K) * (double)esz +
         (double)p.M * p.N * (double)osz * (1.0 + (p.residual != nullptr ? 1.0 : 0.0) + (p.C2 != nullptr ? 1.0 : 0.0));
}

template <typename T, typename TOut>
int launch_gemm(const GemmP& p0, hipStream_t st) {
  GemmP p = p0;
  omp_ctx& cx = omp_cur();
  int which = cx.force_gemm;
  const int kq = 4 * Mma<T>::KSTEP;
  if (p.ln_g != nullptr || which == 4 || (which == 0 && p.small_hint && p.M <= 64 && p.K % kq == 0)) {
    if (p.a_wrap > 0 || p.split_out) {
      omp_set_error("omp_gemm_bias_act: the split-K small-M kernel takes no split-bf16 operands / destinations");
      return OMP_ERR_UNSUPPORTED;
    }
    if (p.M > 64 || p.K % kq != 0 || p.trans_out) {
      omp_set_error("omp_gemm_bias_act: split-K small-M kernel needs M <= 64, K %% %d == 0, no trans_out", kq);
      return OMP_ERR_UNSUPPORTED;
    }
    if (p.ln_g != nullptr && p.K > 1024) {
      omp_set_error("omp_gemm_bias_act: fused LayerNorm supports K <= 1024");
      return OMP_ERR_UNSUPPORTED;
    }
    if (cx.gemm_choice_only) { cx.gemm_last_choice = 4; return OMP_OK; }
    return launch_small<T, TOut>(p, st);
  }
  if (p.store_mode == OMP_STORE_ROWSTAT) which = 5;   // the row statistics live in the 128 x 128 kernel's epilogue
  if (which == 0) {
    // round 4 (profiles/r04b_kbench_dec_rows*.txt, the 10240-row polygon / recognition phases of a 160-image engine call): 64x64
    // tiles only below 256 tiles of 128x128 (they lose 10-25 % to the 128x128 kernel at 320 tiles, at K = 512 and at the bf16x3
    // engine's K = 1536 / 6144 alike)
    if (p.M <= 64) which = 3;
    else if (ceil_div64(p.M, 128) * ceil_div64(p.N, 128) < 256) which = 6;
    else which = 5;
    // 256x256 phase-interleaved tiles once they fill the chip and the output is wide enough for a 256-column tile to pay
    // (profiles/r02g_kbench_gemm_256.txt: wins on every Swin qkv / fc1 / fc2 / proj shape of stages 1-3 with >= 256 tiles, loses
    // at N = 128 and on half-empty grids).  "Fill the chip" = the last round of tiles over the 256 CUs is not mostly idle: 240
    // tiles (one round, 94 %) win, 320 tiles (two rounds, 62 %) lose to the 128x128 kernel (r04b)
    if constexpr (std::is_same<T, bf16_t>::value) {
      const int64_t t256 = ceil_div64(p.M, 256) * ceil_div64(p.N, 256);
      const bool fills = t256 >= 192 && 4 * t256 >= 3 * 256 * ceil_div64(t256, 256);
      if (which == 5 && p.N >= 256 && fills && gemm256_ok(p, true, std::is_same<TOut, bf16_t>::value)) which = 9;
      // the four-wave kernel (gemm4w.inc) produces the same bits; measured against gemm_256 per shape (profiles/r04f_kbench_gemm_4w_v2_*):
      // equal within the box-to-box spread on most, 7-13 % faster on the half-million-row products of stage 1 without an activation,
      // 7-13 % slower behind a GELU epilogue (four waves instead of eight do the vector work) -- it takes the former
      if (which == 9 && p.act == OMP_ACT_NONE && p.store_mode == OMP_STORE_PLAIN && p.M >= 262144 && (p.K >= 1024 || p.N >= 768)) which = 10;
      // the persistent four-wave kernel (gemm4wp.inc; same bits again) where its register-only epilogue and its missing prologues pay
      // (profiles/r04s_kbench_gemm_4w_p_*.txt, r04v_kbench_ab_k9_k20.txt: three boxes): behind a GELU on a bf16 destination (+9 % at
      // K = 512, +3-13 % at K = 768, +23-30 % at K = 256, equal at K = 1024) and on the K = 768 products without a residual (ViT-B qkv
      // +7 %, the bf16x3 stage-1 qkv +7 % over gemm_4w); equal or behind gemm_256 elsewhere (K = 256 without GELU: +12 % on one box,
      // -7 % on another; long K: the operand stream paces both kernels)
      if ((which == 9 || which == 10) && !p.split_out && gemm4wp_ok(p, true, std::is_same<TOut, bf16_t>::value) &&
          ((p.act == OMP_ACT_GELU && std::is_same<TOut, bf16_t>::value) || (p.K == 768 && p.residual == nullptr && p.N >= 768 && p.act == OMP_ACT_NONE)))
        which = 20;
      // bf16x3 operands: the three products fused over shared operand tiles (gemm4wp.inc, X3) wherever a 256x256-tile kernel was chosen
      // and the shape has no ragged edge -- 2/3 of the operand bytes; faster on every encoder product of the parity engine
      // (profiles/r04y_kbench_gemm_x3_fused.txt: -1...-17 %).  Its summation order is chunk by chunk, not plane by plane: equal to the
      // three-pass kernels within fp32 rounding, not bit for bit.
      if ((which == 9 || which == 10 || which == 20) && gemm4wx3_ok(p, true, std::is_same<TOut, bf16_t>::value)) which = 22;
    }
  }
  if (cx.gemm_choice_only) {
    // omp_debug_gemm_choice: the dispatch table is host logic, testable without a GPU.  It reports a selector only for arguments the
    // launch path below accepts -- a forced selector that does not take the product (no second destination, a shape its tiles do not
    // cover, non-bf16 operands) is the launch path's OMP_ERR_UNSUPPORTED here too
    bool ok = true;
    constexpr bool BO = std::is_same<TOut, bf16_t>::value;
    if (p.C2 != nullptr && which != 5 && which != 6 && which != 9 && which != 10 && which != 11 && which != 15 && which != 16 && which != 18 && which != 20 && which != 21 && which != 22) ok = false;
    else if (which == 9 || (which >= 10 && which <= 14) || (which >= 16 && which <= 18) || (which >= 20 && which <= 22)) {
      if constexpr (std::is_same<T, bf16_t>::value) {
        if (which == 9) ok = gemm256_ok(p, true, BO);
        else if (which == 10 || which == 11) ok = gemm4w_ok(p, true, BO);
        else if (which >= 12 && which <= 14) ok = BO && gemm4w_ok(p, true, true) && p.store_mode == OMP_STORE_PLAIN && !p.split_out;
        else if (which >= 16 && which <= 18) ok = gemm4wr_ok(p, true, BO) && (which != 17 || (BO && !p.split_out));
        else if (which == 20 || which == 21) ok = gemm4wp_ok(p, true, BO) && (which != 21 || !BO);
        else ok = gemm4wx3_ok(p, true, BO);
      } else {
        ok = false;
      }
    } else if (which != 3 && which != 5 && which != 6 && which != 15) {
      ok = false;
    }
    if (!ok) {
      omp_set_error("omp_gemm_bias_act: kernel selector %d does not take this product", which);
      return OMP_ERR_UNSUPPORTED;
    }
    cx.gemm_last_choice = which;
    return OMP_OK;
  }
  if (p.C2 != nullptr && which != 5 && which != 6 && which != 9 && which != 10 && which != 11 && which != 15 && which != 16 && which != 18 && which != 20 && which != 21 && which != 22) {
    omp_set_error("omp_gemm_bias_act: kernel selector %d has no second destination (C2)", which);
    return OMP_ERR_UNSUPPORTED;
  }
  if (which == 5) {
    // bench.py's matrix-core roofline leg: hipEvent bracket + flop count of the large-M GEMMs
    const int pcls = p.M >= 32768 ? OMP_PROF_GEMM : OMP_PROF_GEMM_DEC;
    const int slot = omp_prof_active(pcls) ? omp_prof_begin(pcls, st, 2.0 * (double)p.M * p.N * p.K, gemm_alg_bytes(p, sizeof(T), sizeof(TOut))) : -1;
    int rc = launch_dma<T, TOut, 128, 128, 2>(p, st);
    if (slot >= 0) omp_prof_end(pcls, slot, st);
    if (rc != OMP_OK) return rc;
  } else if (which == 6) {
    // mid-size problems (decoder phases with 65..~4000 rows, small-image encoders): 64x64 tiles and a deep
    // ring so that the K extent is in flight at once; a shallower ring (2 workgroups per CU) once the grid
    // is large enough to want the occupancy instead
    int rc = (ceil_div64(p.M, 64) * ceil_div64(p.N, 64) <= 512) ? launch_dma<T, TOut, 64, 64, 8>(p, st)
                                                                 : launch_dma<T, TOut, 64, 64, 4>(p, st);
    if (rc != OMP_OK) return rc;
  } else if (which == 9) {           // 256x256 phase-interleaved kernel (gemm256.inc)
    if constexpr (std::is_same<T, bf16_t>::value) {
      if (!gemm256_ok(p, true, std::is_same<TOut, bf16_t>::value)) {
        omp_set_error("omp_gemm_bias_act: selector 9 (256x256 tiles) needs bf16 operands, K %% 64 == 0, K >= 128, N %% 8 == 0 (blocked K / V^T slabs: bf16 only)");
        return OMP_ERR_UNSUPPORTED;
      }
      const int pcls = p.M >= 32768 ? OMP_PROF_GEMM : OMP_PROF_GEMM_DEC;
    const int slot = omp_prof_active(pcls) ? omp_prof_begin(pcls, st, 2.0 * (double)p.M * p.N * p.K, gemm_alg_bytes(p, sizeof(T), sizeof(TOut))) : -1;
      int rc = launch_256<TOut>(p, st);
      if (slot >= 0) omp_prof_end(pcls, slot, st);
      if (rc != OMP_OK) return rc;
    } else {
      omp_set_error("omp_gemm_bias_act: selector 9 (256x256 tiles) is bf16-only");
      return OMP_ERR_UNSUPPORTED;
    }
  } else if (which == 10 || which == 11) {   // 256x256 tiles on four waves (gemm4w.inc); 11 is kept as an alias of 10
    if constexpr (std::is_same<T, bf16_t>::value) {
      if (!gemm4w_ok(p, true, std::is_same<TOut, bf16_t>::value)) {
        omp_set_error("omp_gemm_bias_act: selector %d (256x256 tiles on four waves) takes the shapes of selector 9", which);
        return OMP_ERR_UNSUPPORTED;
      }
      const int pcls = p.M >= 32768 ? OMP_PROF_GEMM : OMP_PROF_GEMM_DEC;
    const int slot = omp_prof_active(pcls) ? omp_prof_begin(pcls, st, 2.0 * (double)p.M * p.N * p.K, gemm_alg_bytes(p, sizeof(T), sizeof(TOut))) : -1;
      int rc = launch_4w<TOut, 5>(p, st);
      if (slot >= 0) omp_prof_end(pcls, slot, st);
      if (rc != OMP_OK) return rc;
    } else {
      omp_set_error("omp_gemm_bias_act: selector %d (256x256 tiles on four waves) is bf16-only", which);
      return OMP_ERR_UNSUPPORTED;
    }
  } else if (which >= 12 && which <= 14) {   // development: gemm_4w ablations (wrong results, valid timing), plain bf16 destination only
    if constexpr (std::is_same<T, bf16_t>::value && std::is_same<TOut, bf16_t>::value) {
      if (!gemm4w_ok(p, true, true) || p.store_mode != OMP_STORE_PLAIN || p.split_out) { omp_set_error("omp_gemm_bias_act: selectors 12..14 take plain bf16 products"); return OMP_ERR_UNSUPPORTED; }
      int rc = which == 12 ? launch_4w_sm<TOut, OMP_STORE_PLAIN, false, 5, 1>(p, st) : which == 13 ? launch_4w_sm<TOut, OMP_STORE_PLAIN, false, 5, 2>(p, st)
                                                                                                     : launch_4w_sm<TOut, OMP_STORE_PLAIN, false, 5, 3>(p, st);
      if (rc != OMP_OK) return rc;
    } else {
      omp_set_error("omp_gemm_bias_act: selectors 12..14 take plain bf16 products");
      return OMP_ERR_UNSUPPORTED;
    }
  } else if (which >= 16 && which <= 18) {   // 256x256 tiles on four waves, weights streamed into registers (gemm4wr.inc); 17: without its MFMAs (wrong results, valid timing)
    if constexpr (std::is_same<T, bf16_t>::value) {
      if (!gemm4wr_ok(p, true, std::is_same<TOut, bf16_t>::value)) {
        omp_set_error("omp_gemm_bias_act: selector %d (register-streamed weights) takes row-major destinations of selector 9 with K %% 256 == 0", which);
        return OMP_ERR_UNSUPPORTED;
      }
      const int pcls = p.M >= 32768 ? OMP_PROF_GEMM : OMP_PROF_GEMM_DEC;
      const int slot = omp_prof_active(pcls) ? omp_prof_begin(pcls, st, 2.0 * (double)p.M * p.N * p.K, gemm_alg_bytes(p, sizeof(T), sizeof(TOut))) : -1;
      int rc = OMP_OK;
      if (which == 16) rc = launch_4wr<TOut>(p, st);
      else if (which == 18) {   // development: per-workgroup phase timestamps (tools/gemm4wr_trace.py)
        if (p.split_out || cx.gemm_trace == nullptr || (long long)ceil_div64(p.M, 256) * ceil_div64(p.N, 256) > cx.gemm_trace_cap) {
          omp_set_error("omp_gemm_bias_act: selector 18 needs omp_debug_set_gemm_trace(buffer for every 256x256 tile) and a plain destination");
          return OMP_ERR_INVALID;
        }
        p.trace = cx.gemm_trace;
        rc = launch_4wr_t<TOut, false, 4>(p, st);
      } else if constexpr (std::is_same<TOut, bf16_t>::value) {
        if (p.split_out) { omp_set_error("omp_gemm_bias_act: selector 17 takes plain bf16 products"); return OMP_ERR_UNSUPPORTED; }
        rc = launch_4wr_t<TOut, false, 3>(p, st);
      } else {
        omp_set_error("omp_gemm_bias_act: selector 17 takes plain bf16 products");
        return OMP_ERR_UNSUPPORTED;
      }
      if (slot >= 0) omp_prof_end(pcls, slot, st);
      if (rc != OMP_OK) return rc;
    } else {
      omp_set_error("omp_gemm_bias_act: selector %d (register-streamed weights) is bf16-only", which);
      return OMP_ERR_UNSUPPORTED;
    }
  } else if (which == 20 || which == 21) {   // persistent 256x256 tiles on four waves, register-only epilogue (gemm4wp.inc); 21: 2/3 of its operand bytes (wrong results, valid timing)
    if constexpr (std::is_same<T, bf16_t>::value) {
      if (!gemm4wp_ok(p, true, std::is_same<TOut, bf16_t>::value)) {
        omp_set_error("omp_gemm_bias_act: selector 20 (persistent four-wave tiles) takes row-major destinations with M, N, K multiples of 256 and 16-byte aligned bias / rows");
        return OMP_ERR_UNSUPPORTED;
      }
      const int pcls = p.M >= 32768 ? OMP_PROF_GEMM : OMP_PROF_GEMM_DEC;
      const int slot = omp_prof_active(pcls) ? omp_prof_begin(pcls, st, 2.0 * (double)p.M * p.N * p.K, gemm_alg_bytes(p, sizeof(T), sizeof(TOut))) : -1;
      int rc = OMP_OK;
      if (which == 20) rc = launch_4wp<TOut>(p, st);
      else if constexpr (!std::is_same<TOut, bf16_t>::value) rc = launch_4wp_t<TOut, false, 5>(p, st);
      else { omp_set_error("omp_gemm_bias_act: selector 21 takes fp32 destinations"); return OMP_ERR_UNSUPPORTED; }
      if (slot >= 0) omp_prof_end(pcls, slot, st);
      if (rc != OMP_OK) return rc;
    } else {
      omp_set_error("omp_gemm_bias_act: selector 20 (persistent four-wave tiles) is bf16-only");
      return OMP_ERR_UNSUPPORTED;
    }
  } else if (which == 22) {          // bf16x3: the three products fused over shared operand tiles (gemm4wp.inc, X3)
    if constexpr (std::is_same<T, bf16_t>::value) {
      if (!gemm4wx3_ok(p, true, std::is_same<TOut, bf16_t>::value)) {
        omp_set_error("omp_gemm_bias_act: selector 22 takes bf16x3 operands (a_wrap = 2 K0, K = 3 K0, K0 %% 128 == 0), M and N multiples of 256, fp32 or split destination");
        return OMP_ERR_UNSUPPORTED;
      }
      const int pcls = p.M >= 32768 ? OMP_PROF_GEMM : OMP_PROF_GEMM_DEC;
      const int slot = omp_prof_active(pcls) ? omp_prof_begin(pcls, st, 2.0 * (double)p.M * p.N * p.K, gemm_alg_bytes(p, sizeof(T), sizeof(TOut))) : -1;
      int rc = launch_4wx3<TOut>(p, st);
      if (slot >= 0) omp_prof_end(pcls, slot, st);
      if (rc != OMP_OK) return rc;
    } else {
      omp_set_error("omp_gemm_bias_act: selector 22 is bf16-only");
      return OMP_ERR_UNSUPPORTED;
    }
  } else if (which == 15) {          // development: gemm_dma<128,128,2> with per-workgroup phase timestamps
    p.tiles_m = (int)ceil_div64(p.M, 128); p.tiles_n = (int)ceil_div64(p.N, 128);
    if (cx.gemm_trace == nullptr || (long long)p.tiles_m * p.tiles_n > cx.gemm_trace_cap) {
      omp_set_error("omp_gemm_bias_act: selector 15 needs omp_debug_set_gemm_trace(buffer for >= %d workgroups)", p.tiles_m * p.tiles_n);
      return OMP_ERR_INVALID;
    }
    p.trace = cx.gemm_trace;
    int rc = launch_dma<T, TOut, 128, 128, 2, true>(p, st);
    if (rc != OMP_OK) return rc;
  } else if (which == 3) {
    dim3 grid((unsigned)ceil_div64(p.N, 64), (unsigned)ceil_div64(p.M, 16));
    hipLaunchKernelGGL((gemm_rows<T, TOut>), grid, dim3(256), 0, st, p);
  } else {
    omp_set_error("omp_gemm_bias_act: unknown kernel selector %d", which);
    return OMP_ERR_INVALID;
  }
  OMP_CHECK_LAUNCH("omp_gemm_bias_act");
  return OMP_OK;
}

}  // namespace

extern "C" int omp_debug_set_gemm_trace(void* buffer, int64_t n_workgroups) {
  omp_cur().gemm_trace = reinterpret_cast<unsigned long long*>(buffer);
  omp_cur().gemm_trace_cap = buffer ? n_workgroups : 0;
  return OMP_OK;
}

// The kernel selector omp_gemm_bias_act would take for these arguments (after its argument checks), without launching anything and
// without touching a device: pointers are only tested for null / alignment.  > 0: selector (omp355_debug.h), < 0: the error code.
extern "C" int omp_debug_gemm_choice(const omp_gemm_args* a) {
  omp_ctx& cx = omp_cur();
  cx.gemm_choice_only = 1;
  cx.gemm_last_choice = 0;
  const int rc = omp_gemm_bias_act(a, nullptr);
  cx.gemm_choice_only = 0;
  return rc == OMP_OK ? cx.gemm_last_choice : rc;
}

extern "C" int omp_debug_force_gemm_kernel(int which) {
  omp_cur().force_gemm = which;
  return OMP_OK;
}

extern "C" int omp_gemm_bias_act(const omp_gemm_args* a, omp_stream_t s) {
  OMP_CHECK_ARG(a != nullptr, "omp_gemm_bias_act: null args");
  OMP_CHECK_ARG(a->A && a->W && a->C, "omp_gemm_bias_act: null A/W/C");
  OMP_CHECK_ARG(a->M > 0 && a->N > 0 && a->K > 0, "omp_gemm_bias_act: bad shape M=%lld N=%d K=%d",
                (long long)a->M, a->N, a->K);
  OMP_CHECK_ARG(a->dtype == OMP_F32 || a->dtype == OMP_BF16, "omp_gemm_bias_act: bad dtype %d", a->dtype);
  OMP_CHECK_ARG(a->out_dtype == a->dtype || a->out_dtype == OMP_F32 || (a->out_dtype == OMP_BF16X2 && a->dtype == OMP_BF16),
                "omp_gemm_bias_act: out_dtype must equal dtype, be f32, or be split-bf16 pairs with bf16 operands");
  const bool split_out = a->out_dtype == OMP_BF16X2;
  OMP_CHECK_ARG(!split_out || (a->residual == nullptr && a->C2 == nullptr && !a->trans_out &&
                               (a->store_mode == OMP_STORE_PLAIN ? (a->ldc >= 2 * (int64_t)a->N && a->N % 8 == 0) : a->kv_key_block == 32)),
                "omp_gemm_bias_act: split-bf16 destinations are plain [M, 2N] rows (N %% 8 == 0) or split-plane K / V^T slabs of 32-key blocks, "
                "without residual / second destination");
  OMP_CHECK_ARG(a->a_wrap >= 0 && (a->a_wrap == 0 || (a->dtype == OMP_BF16 && a->a_wrap % 64 == 0 && a->a_wrap < a->K && a->K <= 2 * a->a_wrap &&
                                                      a->ln_gamma == nullptr)),
                "omp_gemm_bias_act: a_wrap needs bf16 operands, a_wrap %% 64 == 0 and a_wrap < K <= 2 * a_wrap");
  const int esz = a->dtype == OMP_F32 ? 4 : 2;
  const int ktile = 128 / esz;
  OMP_CHECK_ARG(a->K % ktile == 0, "omp_gemm_bias_act: K=%d must be a multiple of %d", a->K, ktile);
  OMP_CHECK_ARG((a->lda * (a->ln_gamma ? 4 : esz)) % 16 == 0 && (a->ldw * esz) % 16 == 0,
                "omp_gemm_bias_act: lda/ldw rows must be 16-byte aligned");
  OMP_CHECK_ARG(((uintptr_t)a->A % 16) == 0 && ((uintptr_t)a->W % 16) == 0 && ((uintptr_t)a->C % 16) == 0,
                "omp_gemm_bias_act: A/W/C must be 16-byte aligned");
  OMP_CHECK_ARG(!a->trans_out || (a->trans_rows > 0 && a->trans_ld >= a->trans_rows),
                "omp_gemm_bias_act: trans_rows must be > 0 and trans_ld >= trans_rows");
  OMP_CHECK_ARG(a->M < (1ll << 31), "omp_gemm_bias_act: M too large");
  GemmP p;
  p.A = a->A; p.lda = a->lda; p.W = a->W; p.ldw = a->ldw;
  p.bias = a->bias; p.bias_row = a->bias_row; p.bias_row_stride = a->bias_row_stride;
  p.residual = a->residual; p.ldr = a->ldr; p.C = a->C; p.ldc = a->ldc;
  p.M = a->M; p.N = a->N; p.K = a->K; p.act = a->act;
  p.trans_out = a->trans_out; p.trans_rows = a->trans_rows; p.trans_ld = a->trans_ld; p.tiles_m = p.tiles_n = 0;
  p.ln_g = a->ln_gamma; p.ln_b = a->ln_beta; p.ln_eps = a->ln_eps; p.small_hint = a->small_m_splitk;
  p.store_mode = a->store_mode; p.bias_m = a->bias_along_m; p.trace = nullptr;
  p.C2 = a->C2; p.ldc2 = a->ldc2;
  p.a_wrap = a->a_wrap; p.split_out = split_out ? 1 : 0;
  if (p.C2 != nullptr) {
    OMP_CHECK_ARG(a->residual != nullptr && a->store_mode == OMP_STORE_PLAIN && !a->trans_out && a->ln_gamma == nullptr && a->M > 64 &&
                      a->N % (16 / (a->out_dtype == OMP_F32 ? 4 : 2)) == 0 && a->ldc2 % (16 / (a->out_dtype == OMP_F32 ? 4 : 2)) == 0 &&
                      a->ldc % (16 / (a->out_dtype == OMP_F32 ? 4 : 2)) == 0 && a->ldr % (16 / (a->out_dtype == OMP_F32 ? 4 : 2)) == 0 &&
                      ((uintptr_t)a->C2 % 16) == 0,
                  "omp_gemm_bias_act: C2 needs a residual, a plain 16-byte-aligned destination, M > 64 and N a multiple of the 16-byte chunk");
  }
  p.kv_B = a->kv_images; p.kv_tok = a->kv_tokens; p.kv_mpad = a->kv_mpad; p.kv_nH = a->kv_heads; p.kv_kb = a->kv_key_block;
  if (p.store_mode == OMP_STORE_ROWSTAT) {
    OMP_CHECK_ARG(a->out_dtype == OMP_F32 && !a->trans_out && a->residual == nullptr && a->C2 == nullptr && a->ln_gamma == nullptr && !a->small_m_splitk &&
                      a->act == OMP_ACT_NONE && !a->bias_along_m,
                  "omp_gemm_bias_act: OMP_STORE_ROWSTAT takes an fp32 destination [M][2 ceil(N / 128)][4] and no residual / activation / C2 / trans_out / LayerNorm prologue");
  } else if (p.store_mode != OMP_STORE_PLAIN) {
    OMP_CHECK_ARG(p.store_mode == OMP_STORE_KBLK || p.store_mode == OMP_STORE_VBLK, "omp_gemm_bias_act: bad store_mode %d", p.store_mode);
    OMP_CHECK_ARG(!a->trans_out && a->residual == nullptr && a->ln_gamma == nullptr && !a->small_m_splitk,
                  "omp_gemm_bias_act: blocked K/V stores take no residual / trans_out / LayerNorm prologue");
    OMP_CHECK_ARG(p.kv_B > 0 && p.kv_tok > 0 && p.kv_nH > 0 && (p.kv_kb == 16 || p.kv_kb == 32) && p.kv_mpad >= p.kv_tok &&
                      p.kv_mpad % p.kv_kb == 0,
                  "omp_gemm_bias_act: bad blocked K/V geometry");
    const int64_t toks = (int64_t)p.kv_B * p.kv_tok;
    OMP_CHECK_ARG(p.store_mode == OMP_STORE_KBLK ? (a->M == toks && a->N % (p.kv_nH * 64) == 0)
                                                 : (a->N == toks && a->M % (p.kv_nH * 64) == 0),
                  "omp_gemm_bias_act: blocked K/V store shape mismatch (M=%lld N=%d)", (long long)a->M, a->N);
  }
  OMP_CHECK_ARG(!p.bias_m || (a->bias_row == nullptr), "omp_gemm_bias_act: bias_along_m excludes bias_row");
  OMP_CHECK_ARG((a->ln_gamma == nullptr) == (a->ln_beta == nullptr), "omp_gemm_bias_act: ln_gamma and ln_beta go together");
  hipStream_t st = (hipStream_t)s;
  if (a->dtype == OMP_F32) return launch_gemm<float, float>(p, st);
  if (a->out_dtype == OMP_F32) return launch_gemm<bf16_t, float>(p, st);
  return launch_gemm<bf16_t, bf16_t>(p, st);
}
