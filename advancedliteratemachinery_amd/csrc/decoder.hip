// Point-conditioned autoregressive decoders: KV-cached greedy decoding on gfx950.
//
// The reference re-runs the whole prefix through 4 layers at every greedy step and physically
// replicates the image memory once per text instance (transformer.py:74-100).  Algebraically the
// same computation is done here incrementally:
//   * self-attention K/V of past positions live in a cache (the causal mask makes them immutable);
//   * cross-attention K = (memory+pos) Wk^T + bk and V = memory Wv^T + bv are computed ONCE per
//     image and shared by every query row of that image (rows are grouped in tiles of <= 16
//     consecutive rows of one image);
//   * (y + qpos) Wq^T == y Wq^T + (qpos Wq^T): the position term is a per-position bias table;
//   * the 3-layer head runs on the newest position only.
// All rows of a phase sit at the same sequence position, kept in device memory (*d_pos) so the
// step is position-independent on the host side and can be replayed as a hipGraph.
#include <stdlib.h>

#include <mutex>
#include <vector>

#include "common.h"

namespace {

constexpr int DH = 64;  // decoder head_dim (512 / 8)

// ---------------------------------------------------------------------------------------------
// embedding + LayerNorm  (one wave per row)
// ---------------------------------------------------------------------------------------------
template <typename TO>
__global__ __launch_bounds__(256) void dec_embed_ln_kernel(const int32_t* __restrict__ seq, int seq_ld,
                                                           const int32_t* __restrict__ d_pos,
                                                           const float* __restrict__ word,
                                                           const float* __restrict__ postab,
                                                           const float* __restrict__ g,
                                                           const float* __restrict__ be, float* __restrict__ x,
                                                           TO* __restrict__ y, int R, int d, float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = blockIdx.x * 4 + wave;
  if (r >= R) return;
  const int p = *d_pos;
  const int tok = seq[(int64_t)r * seq_ld + p];
  const float* we = word + (int64_t)tok * d;
  const float* pe = postab + (int64_t)p * d;
  float v[4][4];
  float s = 0.f;
  const int nch = d / 4;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int c = lane + it * 64;
    if (c < nch) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(we + c * 4);
      const f32x4 b = *reinterpret_cast<const f32x4*>(pe + c * 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) { v[it][i] = a[i] + b[i]; s += v[it][i]; }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[it][i] = 0.f;
    }
  }
  s = wave_sum(s);
  const float mean = s / (float)d;
  float q = 0.f;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    if (lane + it * 64 < nch) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { const float t = v[it][i] - mean; q += t * t; }
    }
  }
  q = wave_sum(q);
  const float rstd = 1.0f / sqrtf(q / (float)d + eps);
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int c = lane + it * 64;
    if (c < nch) {
      const f32x4 gg = *reinterpret_cast<const f32x4*>(g + c * 4);
      const f32x4 bb = *reinterpret_cast<const f32x4*>(be + c * 4);
      float o[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = (v[it][i] - mean) * rstd * gg[i] + bb[i];
      if (x != nullptr) *reinterpret_cast<f32x4*>(x + (int64_t)r * d + c * 4) = f32x4{o[0], o[1], o[2], o[3]};
      if (y != nullptr) {
#pragma unroll
        for (int i = 0; i < 4; ++i) y[(int64_t)r * d + c * 4 + i] = from_f32<TO>(o[i]);
      }
    }
  }
}

// load 8 consecutive elements as floats
__device__ __forceinline__ void load8(const float* p, float* o) {
  unpack16(ld16<float>(p), o);
  unpack16(ld16<float>(p + 4), o + 4);
}
__device__ __forceinline__ void load8(const bf16_t* p, float* o) { unpack16(ld16<bf16_t>(p), o); }
__device__ __forceinline__ void store8(float* p, const float* v) {
  *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
  *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
}
__device__ __forceinline__ void store8(bf16_t* p, const float* v) {
  bf16x8 o;
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = (bf16_t)v[i];
  *reinterpret_cast<bf16x8*>(p) = o;
}

// ---------------------------------------------------------------------------------------------
// causal self-attention step with KV cache: one wave per (row, head).
// lane = (js = lane>>3 : key slot, dc = lane&7 : 8-dim chunk); 8 keys per iteration, each key row
// (64 dims) read as 8 x 16 B (bf16) -- full 128-byte lines.  Online softmax per key slot, merged
// across the 8 slots at the end.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void dec_self_attn_kernel(const T* __restrict__ qkv, T* __restrict__ kc,
                                                            T* __restrict__ vc, T* __restrict__ out,
                                                            const int32_t* __restrict__ d_pos, int R, int nH,
                                                            int d, int Lmax) {
  // 4 heads of one row per workgroup (waves are independent; fewer, fatter workgroups dispatch faster)
  const int r = blockIdx.x, h = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (h >= nH) return;
  const int lane = threadIdx.x & 63, js = lane >> 3, dc = lane & 7;
  const int p = *d_pos;
  const T* row = qkv + (int64_t)r * 3 * d + h * DH + dc * 8;
  float q[8], kn[8], vn[8];
  load8(row, q);
  load8(row + d, kn);
  load8(row + 2 * d, vn);
#pragma unroll
  for (int i = 0; i < 8; ++i) q[i] *= 0.125f;  // 1/sqrt(64), applied to q like nn.MultiheadAttention
  T* kbase = kc + ((int64_t)r * Lmax) * d + h * DH + dc * 8;
  T* vbase = vc + ((int64_t)r * Lmax) * d + h * DH + dc * 8;
  if (js == 0) {
    store8(kbase + (int64_t)p * d, kn);
    store8(vbase + (int64_t)p * d, vn);
  }
  float m = -INFINITY, l = 0.f, o[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = 0.f;
  // 4 key blocks (32 keys) per outer iteration: all 8 loads are issued before the dependent online-softmax
  // updates, so one memory round trip covers 32 keys instead of 8.
  for (int j0 = 0; j0 <= p; j0 += 32) {
    float kk[4][8], vv[4][8];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = j0 + u * 8 + js;
      if (j < p) {
        load8(kbase + (int64_t)j * d, kk[u]);
        load8(vbase + (int64_t)j * d, vv[u]);
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) { kk[u][i] = kn[i]; vv[u][i] = vn[i]; }
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = j0 + u * 8 + js;
      float sdot = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) sdot = fmaf(q[i], kk[u][i], sdot);
      sdot += __shfl_xor(sdot, 1, 64);
      sdot += __shfl_xor(sdot, 2, 64);
      sdot += __shfl_xor(sdot, 4, 64);
      if (j <= p) {
        const float mn = fmaxf(m, sdot);
        const float a = expf(m - mn);  // exp(-inf) = 0 on the first key
        const float pj = expf(sdot - mn);
        l = l * a + pj;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = o[i] * a + pj * vv[u][i];
        m = mn;
      }
    }
  }
  // merge the 8 key slots (lanes differing in bits 3..5)
  float mall = m;
  mall = fmaxf(mall, __shfl_xor(mall, 8, 64));
  mall = fmaxf(mall, __shfl_xor(mall, 16, 64));
  mall = fmaxf(mall, __shfl_xor(mall, 32, 64));
  const float sc = (m == -INFINITY) ? 0.f : expf(m - mall);
  l *= sc;
  l += __shfl_xor(l, 8, 64); l += __shfl_xor(l, 16, 64); l += __shfl_xor(l, 32, 64);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float t = o[i] * sc;
    t += __shfl_xor(t, 8, 64); t += __shfl_xor(t, 16, 64); t += __shfl_xor(t, 32, 64);
    o[i] = t / l;
  }
  if (js == 0) store8(out + (int64_t)r * d + h * DH + dc * 8, o);
}

// ---------------------------------------------------------------------------------------------
// The same step for MANY rows (polygon / recognition decoders: thousands of rows, <= ~40 cached positions): one wave per
// ROW, all 8 heads at once.  Lane l holds dims (l & 7) * 8 .. + 8 of head l >> 3, so one cache position of a row
// ([Lmax][512] layout, 1 KB in bf16) is ONE fully coalesced 16-byte-per-lane load, the append of the new k / v is one
// coalesced store, and a row costs one wave instead of eight (round 1: one wave per (row, head), 16 384 workgroups at
// R = 8192, 2.3-2.9 TB/s in isolation and 1.1 TB/s in the bench).  Keys are visited in order with an online softmax
// (base 2: q carries 1/sqrt(64) * log2 e); the score of a key is an 8-lane xor-shuffle reduction.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256, sizeof(T) == 2 ? 4 : 2) void dec_self_attn_row_kernel(const T* __restrict__ qkv, T* __restrict__ kc, T* __restrict__ vc,
                                                                T* __restrict__ out, const int32_t* __restrict__ d_pos, int R,
                                                                int d, int Lmax, bf16_t* __restrict__ out_split = nullptr) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const int p = *d_pos;
  const T* row = qkv + (int64_t)r * 3 * d + lane * 8;
  float q[8], kn[8], vn[8];
  load8(row, q);
  load8(row + d, kn);
  load8(row + 2 * d, vn);
#pragma unroll
  for (int i = 0; i < 8; ++i) q[i] *= 0.125f * 1.4426950408889634f;
  T* kbase = kc + (int64_t)r * Lmax * d + lane * 8;
  T* vbase = vc + (int64_t)r * Lmax * d + lane * 8;
  store8(kbase + (int64_t)p * d, kn);
  store8(vbase + (int64_t)p * d, vn);
  float m = -INFINITY, l = 0.f, o[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = 0.f;
  constexpr int U = 4;   // cache positions per memory round trip (2 U loads in flight per lane)
  typedef typename Vec16<T>::type raw_t;   // 16 bytes as loaded; widened to fp32 only when consumed (registers = occupancy here)
  constexpr int NR = 8 * (int)sizeof(T) / 16;   // 16-byte pieces per 8 elements: 1 (bf16) / 2 (f32)
  for (int j0 = 0; j0 <= p; j0 += U) {
    raw_t kraw[U][NR], vraw[U][NR];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = j0 + u;
      if (j < p) {        // wave-uniform
#pragma unroll
        for (int c = 0; c < NR; ++c) {
          kraw[u][c] = ld16<T>(kbase + (int64_t)j * d + c * (8 / NR));
          vraw[u][c] = ld16<T>(vbase + (int64_t)j * d + c * (8 / NR));
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = j0 + u;
      if (j <= p) {
        float kk[8], vv[8];
        if (j < p) {
#pragma unroll
          for (int c = 0; c < NR; ++c) { unpack16(kraw[u][c], kk + c * (8 / NR)); unpack16(vraw[u][c], vv + c * (8 / NR)); }
        } else {          // position p itself comes from registers
#pragma unroll
          for (int i = 0; i < 8; ++i) { kk[i] = kn[i]; vv[i] = vn[i]; }
        }
        float sdot = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) sdot = fmaf(q[i], kk[i], sdot);
        sdot += __shfl_xor(sdot, 1, 64);
        sdot += __shfl_xor(sdot, 2, 64);
        sdot += __shfl_xor(sdot, 4, 64);
        const float mn = fmaxf(m, sdot);
        const float a = __builtin_amdgcn_exp2f(m - mn);   // exp2(-inf) = 0 on the first key
        const float pj = __builtin_amdgcn_exp2f(sdot - mn);
        l = l * a + pj;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = o[i] * a + pj * vv[i];
        m = mn;
      }
    }
  }
  const float inv = 1.0f / l;
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] *= inv;
  if (out_split != nullptr) {
    // the parity engine's chains take the attention output as split pairs [R, 2 d] = [hi | lo] (hi = bf16(x), lo = bf16(x - hi): the planes
    // omp_split_bf16 writes): written here, the fp32 tensor and the conversion launch behind it (42 MB per 10 240-row launch) do not exist
    bf16x8 hi, lo;
#pragma unroll
    for (int i = 0; i < 8; ++i) { hi[i] = (bf16_t)o[i]; lo[i] = (bf16_t)(o[i] - (float)hi[i]); }
    *reinterpret_cast<bf16x8*>(out_split + (int64_t)r * 2 * d + lane * 8) = hi;
    *reinterpret_cast<bf16x8*>(out_split + (int64_t)r * 2 * d + d + lane * 8) = lo;
    return;
  }
  store8(out + (int64_t)r * d + lane * 8, o);
}


// ---------------------------------------------------------------------------------------------
// Few-row phases (the point decoder: one row per image): FUSED  LayerNorm1 -> q, k, v of ONE head -> cache append ->
// causal self-attention.  A captured decoder step of an 8-image call was 44 launches of ~3.5-19 us each
// (profiles/r03b_prof8_pt_step_timeline.txt): no kernel body there is bound by anything but its own dependent memory round
// trips plus the ~3.5 us a dependent launch costs in a graph.  Self-attention needs q, k, v of ONE head only, so the
// all-to-all boundary between the QKV projection and the attention disappears once a workgroup owns (row tile, head):
//   * grid = (ceil(R / 4), heads), 4 waves; the head's 192 weight rows (q | k | v: 192 KB bf16, L2 / MALL resident) are
//     requested FIRST, all 48 sixteen-byte fragments per lane in flight, before anything that depends on the rows;
//   * wave w normalises row 4 * tile + w (two-pass fp32 statistics; EMBED: LN(word[token] + position) first -- layer 0 --
//     written to the residual stream by the head-0 workgroups) into LDS as the bf16 B operand;
//   * D[feature][row] on the matrix cores (4 of 16 columns live), + per-position bias table, rounded to bf16 exactly where
//     the unfused path stores qkv, into LDS;
//   * wave w then attends row w: lane = (key slot js = l >> 3, 8-dim chunk dc = l & 7), 64 keys per chunk with all of a
//     chunk's 16 loads per lane issued at once and three chunks in flight (192 cache positions per memory round trip), ONE
//     running-max update per chunk (two-phase: all scores of the chunk, then the exponentials).
// Replaces omp_dec_embed_ln (layer 0) + gemm_small<LN> + dec_self_attn_kernel: 3 launches of 8.5 + 19.3 us -> one of 10.7 us
// per layer (8 rows, 135 positions; profiles/r03d_prof8_pt_step_timeline.txt).
// ---------------------------------------------------------------------------------------------
struct FusedSaP {
  const float* x;                 // [R, 512] fp32 residual stream (EMBED: produced here)
  const float* ln_g; const float* ln_b; float eps;
  const bf16_t* W;                // self_attn.in_proj_weight [1536, 512]
  const float* bias_tab;          // [Pmax, 1536] in_proj_bias + position term of q and k
  bf16_t* kc; bf16_t* vc;         // [R, Lmax, 512]
  bf16_t* out;                    // [R, 512]
  const int32_t* d_pos;
  int R, Lmax;
  const int32_t* seq; int seq_ld; const float* word; const float* pos_tab; const float* emb_g; const float* emb_b;
  float* x_out;                   // EMBED: the embedded rows (fp32 residual stream)
};

// two-pass LayerNorm of one 512-wide row held 8 values per lane
__device__ __forceinline__ void ln512(float* v, const float* g, const float* b, int lane, float eps) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i];
  s = wave_sum(s);
  const float mean = s * (1.0f / 512.0f);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) { const float d = v[i] - mean; q += d * d; }
  q = wave_sum(q);
  const float rstd = 1.0f / sqrtf(q * (1.0f / 512.0f) + eps);
  const f32x4 g0 = *reinterpret_cast<const f32x4*>(g + lane * 8), g1 = *reinterpret_cast<const f32x4*>(g + lane * 8 + 4);
  const f32x4 b0 = *reinterpret_cast<const f32x4*>(b + lane * 8), b1 = *reinterpret_cast<const f32x4*>(b + lane * 8 + 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[i] = (v[i] - mean) * rstd * g0[i] + b0[i];
    v[i + 4] = (v[i + 4] - mean) * rstd * g1[i] + b1[i];
  }
}

// RT = rows per workgroup (a multiple of 4: RT / 4 rows per wave, one after the other).  Only RT = 4 is built: 16 rows per
// workgroup for the 512-row polygon / recognition phases measured 20-22 us against 25 us for the three launches it replaces
// and slowed the two concurrent decoder streams down (19.4 -> 29.6 ms per 8-image call, profiles/r03d_*).
template <bool EMBED, int RT>
__global__ __launch_bounds__(256) void dec_fused_self_attn_kernel(FusedSaP p) {
  constexpr int D = 512, XP = D + 8;   // row pitch of the LDS image (+16 B: spreads the fragment reads over the banks)
  constexpr int RPW = RT / 4;          // rows per wave
  __shared__ __attribute__((aligned(16))) bf16_t xs[RT][XP];
  __shared__ __attribute__((aligned(16))) float qkv_s[RT][192];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, g = lane >> 4;
  const int h = blockIdx.y, r0 = blockIdx.x * RT;
  const int pos = *p.d_pos;

  // ---- the head's weight rows: independent of the rows, so all of it is requested before anything else ----------------
  bf16x8 wf[3][16];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int f0 = wave * 48 + t * 16;                       // first of 16 local features (q 0..63 | k 64..127 | v 128..191)
    const bf16_t* wp = p.W + (int64_t)((f0 >> 6) * D + h * DH + (f0 & 63) + li) * D + g * 8;
#pragma unroll
    for (int s_ = 0; s_ < 16; ++s_) wf[t][s_] = ld16<bf16_t>(wp + s_ * 32);
  }

  // ---- this wave's rows: (embedding +) LayerNorm -> bf16 B operand in LDS ---------------------------------------------------
#pragma unroll
  for (int u = 0; u < RPW; ++u) {
    const int lr = wave * RPW + u, r = r0 + lr;
    const bool live = r < p.R;
    const int rr = live ? r : p.R - 1;
    float v[8];
    if constexpr (EMBED) {
      const int tok = p.seq[(int64_t)rr * p.seq_ld + pos];
      const float* we = p.word + (int64_t)tok * D + lane * 8;
      const float* pe = p.pos_tab + (int64_t)pos * D + lane * 8;
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(we), a1 = *reinterpret_cast<const f32x4*>(we + 4);
      const f32x4 c0 = *reinterpret_cast<const f32x4*>(pe), c1 = *reinterpret_cast<const f32x4*>(pe + 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) { v[i] = a0[i] + c0[i]; v[i + 4] = a1[i] + c1[i]; }
      ln512(v, p.emb_g, p.emb_b, lane, p.eps);
      if (h == 0 && live) store8(p.x_out + (int64_t)r * D + lane * 8, v);
    } else {
      load8(p.x + (int64_t)rr * D + lane * 8, v);
    }
    ln512(v, p.ln_g, p.ln_b, lane, p.eps);
    store8(&xs[lr][lane * 8], v);
  }
  __syncthreads();

  // ---- q | k | v of head h: D[feature][row], 3 feature tiles per wave, K = 512 ------------------------------------------
  f32x4 acc[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bf16x8 zero8 = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
#pragma unroll
  for (int s_ = 0; s_ < 16; ++s_) {
    const bf16x8 xb = li < RT ? *reinterpret_cast<const bf16x8*>(&xs[li < RT ? li : 0][s_ * 32 + g * 8]) : zero8;
#pragma unroll
    for (int t = 0; t < 3; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t][s_], xb, acc[t], 0, 0, 0);
  }
  if (li < RT) {
    const float* bt = p.bias_tab + (int64_t)pos * (3 * D);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int f = wave * 48 + t * 16 + g * 4;              // acc[t][e] <-> local feature f + e of row li
      const f32x4 bb = *reinterpret_cast<const f32x4*>(bt + (f >> 6) * D + h * DH + (f & 63));
#pragma unroll
      for (int e = 0; e < 4; ++e) qkv_s[li][f + e] = (float)(bf16_t)(acc[t][e] + bb[e]);   // rounded where the unfused path stores qkv
    }
  }
  __syncthreads();

  // ---- causal self-attention of this wave's rows, head h, over cache positions 0 .. pos (no barrier below) ---------------------
  const int js = lane >> 3, dc = lane & 7;
#pragma unroll 1
  for (int u_ = 0; u_ < RPW; ++u_) {
    const int lr = wave * RPW + u_, r = r0 + lr;
    if (r >= p.R) break;   // wave-uniform
    float q[8], kn[8], vn[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      q[i] = qkv_s[lr][dc * 8 + i] * (0.125f * 1.4426950408889634f);   // 1/sqrt(64) on q like nn.MultiheadAttention; base-2 softmax
      kn[i] = qkv_s[lr][64 + dc * 8 + i];
      vn[i] = qkv_s[lr][128 + dc * 8 + i];
    }
    bf16_t* kbase = p.kc + (int64_t)r * p.Lmax * D + h * DH + dc * 8;
    bf16_t* vbase = p.vc + (int64_t)r * p.Lmax * D + h * DH + dc * 8;
    if (js == 0) {
      store8(kbase + (int64_t)pos * D, kn);
      store8(vbase + (int64_t)pos * D, vn);
    }
    constexpr int PF = 3;                       // 64-key chunks in flight
    bf16x8 kr[PF][8], vr[PF][8];
    const int nch = (pos >> 6) + 1;             // keys 0 .. pos
    auto load_chunk = [&](int c, bf16x8* kq, bf16x8* vq) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int j = c * 64 + u * 8 + js;
        if (j < pos) {
          kq[u] = ld16<bf16_t>(kbase + (int64_t)j * D);
          vq[u] = ld16<bf16_t>(vbase + (int64_t)j * D);
        }
      }
    };
#pragma unroll
    for (int u = 0; u < PF; ++u)
      if (u < nch) load_chunk(u, kr[u], vr[u]);
    float m = -INFINITY, l = 0.f, o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = 0.f;
    auto compute = [&](int c, const bf16x8* kq, const bf16x8* vq) {
      float sc[8];
      float cm = -INFINITY;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int j = c * 64 + u * 8 + js;
        float kk[8];
        if (j < pos) unpack16(kq[u], kk);
        else {
#pragma unroll
          for (int i = 0; i < 8; ++i) kk[i] = kn[i];
        }
        float sd = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) sd = fmaf(q[i], kk[i], sd);
        sd += __shfl_xor(sd, 1, 64);
        sd += __shfl_xor(sd, 2, 64);
        sd += __shfl_xor(sd, 4, 64);
        sc[u] = j <= pos ? sd : -INFINITY;
        cm = fmaxf(cm, sc[u]);
      }
      cm = fmaxf(cm, __shfl_xor(cm, 8, 64));
      cm = fmaxf(cm, __shfl_xor(cm, 16, 64));
      cm = fmaxf(cm, __shfl_xor(cm, 32, 64));
      const float mn = fmaxf(m, cm);            // finite: key c * 64 (js = 0, u = 0) is always <= pos
      const float a = __builtin_amdgcn_exp2f(m - mn);
      l *= a;
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] *= a;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int j = c * 64 + u * 8 + js;
        const float pj = __builtin_amdgcn_exp2f(sc[u] - mn);   // exp2(-inf) = 0 beyond the current position
        float vv[8];
        if (j < pos) unpack16(vq[u], vv);
        else {
#pragma unroll
          for (int i = 0; i < 8; ++i) vv[i] = vn[i];
        }
        l += pj;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = fmaf(pj, vv[i], o[i]);
      }
      m = mn;
    };
    for (int c0 = 0; c0 < nch; c0 += PF) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int c = c0 + u;
        if (c < nch) {
          compute(c, kr[u], vr[u]);
          if (c + PF < nch) load_chunk(c + PF, kr[u], vr[u]);
        }
      }
    }
    // the 8 key slots (lanes differing in bits 3..5) share m; their partial sums and outputs add
    l += __shfl_xor(l, 8, 64); l += __shfl_xor(l, 16, 64); l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float t = o[i];
      t += __shfl_xor(t, 8, 64); t += __shfl_xor(t, 16, 64); t += __shfl_xor(t, 32, 64);
      o[i] = t * inv;
    }
    if (js == 0) store8(p.out + (int64_t)r * D + h * DH + dc * 8, o);
  }
}

// ---------------------------------------------------------------------------------------------
// cross attention over the image memory, flash-style on the matrix cores.
//
// Memory layout (written by the K/V projection GEMM epilogues, OMP_STORE_KBLK / OMP_STORE_VBLK):
//   K   [image][head][Mpad][64]            one contiguous M x 128 B (bf16) stream per (image, head)
//   V^T [image][head][Mpad/KB][64][KB]     the same bytes again, blocked by KB keys so that the 16-byte
//                                          A-operand fragment of the PV product is one aligned load
// One workgroup = (query group, head, key split z); its NW waves cut the split's keys again; every
// wave streams its key range ONCE (16-byte loads, PD key blocks in flight in a register ring) and
// applies it to up to QT tiles of 16 queries of that image:
//   S^T[key][query] = K[key,:] . Q[query,:]        (A operand = K rows, B operand = Q rows)
//   O^T[d][query]  += V^T[d][key] * P[key][query]   (A operand = V^T rows, B operand = P)
// The S^T accumulator layout (lane: query = l&15, keys 4*(l>>4)+r) IS the B-operand layout of the
// second product once the k-slots are mapped to keys {k0+4g+r} (and {k0+16+4g+r} for bf16), which is the
// order V^T blocks are stored in, so the probabilities never leave registers.
// The NW key slices merge in LDS; S > 1 workgroup splits go through a small fp32 partial buffer and
// dec_cross_merge_kernel.  All queries of an image share its K/V: the memory is never replicated
// (reference: memory.repeat(1, N, 1), transformer.py:88-96).
// ---------------------------------------------------------------------------------------------
struct CrossP {
  const void* q; int64_t ldq;
  const void* K; const void* V;   // this layer's slabs
  int64_t img_stride;             // elements between images: nH * Mpad * 64 (K and V^T alike)
  const uint8_t* kmask;
  const int32_t* groups;          // {row0, nrows (<= 16*QT), image}
  void* out; int64_t ldo;
  float* partial;
  int M, Mpad, nH, R, kpw;        // kpw: keys per wave, multiple of the key block
  int lo_off;                     // split-plane slabs only: > 0 = the output rows are split-bf16 pairs, lo plane lo_off columns after hi (ldo in bf16 elements)
};

// Slab element types: bf16, float, and bf16s_t = SPLIT-bf16 planes (round 4, the parity engine): every 32-key block of a K / V^T slab is
// [hi plane | lo plane] of bf16 (the bytes of the fp32 block), value = hi + lo to 16 mantissa bits.  q arrives and the output leaves
// in fp32; S and PV run as three bf16 matrix-core products each (K_hi q_hi + K_lo q_hi + K_hi q_lo; V_hi p_hi + V_lo p_hi + V_hi p_lo),
// so the fp32-grade cross-attention is paced by HBM like the bf16 one instead of by the 16x16x4 fp32 matrix-core rate.  A CPU
// simulation (tools/sim_cross_planes.py, profiles/r04_sim_cross_planes.txt) shows that NONE of K, V, q, P may stay a single bf16
// plane under north_star's 1e-3 logit gate.
struct bf16s_t {};
template <typename T>
struct CrossTraits;
template <>
struct CrossTraits<bf16_t> {
  typedef bf16_t elem;
  static constexpr int PL = 1;    // planes per block
  static constexpr int KB = 32;   // keys per block
  static constexpr int NSB = 2;   // 16-key score blocks per key block
  static constexpr int DSTEPS = 2;  // 64 dims / 32
  __device__ static __forceinline__ bf16x8 pfrag(const float* p) {
    bf16x8 f;
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (bf16_t)p[i];
    return f;
  }
};
template <>
struct CrossTraits<bf16s_t> {
  typedef bf16_t elem;
  static constexpr int PL = 2;
  static constexpr int KB = 32;
  static constexpr int NSB = 2;
  static constexpr int DSTEPS = 2;
  __device__ static __forceinline__ bf16x8 pfrag(const float* p) {
    bf16x8 f;
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (bf16_t)p[i];
    return f;
  }
};
template <>
struct CrossTraits<float> {
  typedef float elem;
  static constexpr int PL = 1;
  static constexpr int KB = 16;
  static constexpr int NSB = 1;
  static constexpr int DSTEPS = 4;  // 64 dims / 16
  __device__ static __forceinline__ f32x4 pfrag(const float* p) { return f32x4{p[0], p[1], p[2], p[3]}; }
};

constexpr int CROSS_PSTR = 68;   // m, l, pad, pad, o[64]

// 8 fp32 values -> the hi / lo bf16 fragments of their split representation (lo = bf16(x - hi): x = hi + lo to 2^-17 relative)
__device__ __forceinline__ void split8(const float* v, bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int i = 0; i < 8; ++i) { hi[i] = (bf16_t)v[i]; lo[i] = (bf16_t)(v[i] - (float)hi[i]); }
}

// wait until at most `blocks` of this wave's most recently issued key blocks (IPB DMA instructions each) are
// still in flight (the count must be an immediate; the switch is wave-uniform)
template <int IPB>
__device__ __forceinline__ void wait_dma_blocks(int blocks) {
  switch (blocks) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPB * 1) : "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPB * 2) : "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPB * 3) : "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPB * 4) : "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPB * 5) : "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPB * 6) : "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPB * 7) : "memory"); break;
  }
}

template <typename T, int NW, int QT, int PD, bool NT = false>
__global__ __launch_bounds__(NW * 64) void dec_cross_attn_kernel(CrossP p) {
  typedef CrossTraits<T> CT;
  typedef typename CT::elem E;      // slab element
  typedef Mma<E> MM;
  typedef typename MM::frag frag;
  constexpr int KB = CT::KB, PL = CT::PL;
  constexpr int NKF = CT::NSB * CT::DSTEPS;
  constexpr int PSTR = CROSS_PSTR;
  extern __shared__ __attribute__((aligned(16))) float part[];   // [NW][QT*16 queries][PSTR]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, g = lane >> 4;
  const int grp = blockIdx.x, h = blockIdx.y;
  const int row0 = p.groups[grp * 3], nrows = p.groups[grp * 3 + 1], img = p.groups[grp * 3 + 2];
  const int nqt = (nrows + 15) >> 4;
  const int S = gridDim.z, sp = blockIdx.z;
  const int kbeg = (sp * NW + wave) * p.kpw;
  int kend = kbeg + p.kpw;
  if (kend > p.M) kend = p.M;
  const int nblk = kbeg < kend ? (kend - kbeg + KB - 1) / KB : 0;

  // K / V^T fragment addresses of this lane inside the (image, head) slabs
  // (image, head) slab: Mpad keys x 64 dims x PL planes; block k0 / KB starts at element k0 * 64 * PL, its plane pl KB * 64 further
  const int64_t slab = (int64_t)img * p.img_stride + (int64_t)h * p.Mpad * 64 * PL;
  const E* Kb = reinterpret_cast<const E*>(p.K) + slab + li * 64 + g * MM::KPL;
  const E* Vb = reinterpret_cast<const E*>(p.V) + slab + li * KB + g * MM::KPL;
  const uint8_t* km = p.kmask ? p.kmask + (int64_t)img * p.M : nullptr;

  frag kr[PD][PL * NKF], vr[PD][PL * 4];
  // the K / V^T slabs are streamed exactly once per launch: NT = non-temporal loads (no L2 / MALL allocation)
  auto ldkv = [](const E* ptr) -> frag {
    if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const frag*>(ptr));
    else return ld16<E>(ptr);
  };
  auto load_block = [&](int k0, frag* kf, frag* vf) {
    const E* kp = Kb + (int64_t)k0 * 64 * PL;
#pragma unroll
    for (int pl = 0; pl < PL; ++pl)
#pragma unroll
      for (int sb = 0; sb < CT::NSB; ++sb)
#pragma unroll
        for (int s = 0; s < CT::DSTEPS; ++s) kf[pl * NKF + sb * CT::DSTEPS + s] = ldkv(kp + pl * KB * 64 + sb * 16 * 64 + s * MM::KSTEP);
    const E* vp = Vb + (int64_t)k0 * 64 * PL;
#pragma unroll
    for (int pl = 0; pl < PL; ++pl)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) vf[pl * 4 + dt] = ldkv(vp + pl * KB * 64 + dt * 16 * KB);
  };
  // the K/V stream depends on nothing computed here: put PD key blocks in flight before touching Q
#pragma unroll
  for (int u = 0; u < PD; ++u)
    if (u < nblk) load_block(kbeg + u * KB, kr[u], vr[u]);

  // Q fragments (B operand) of the QT query tiles: query li of the tile (clamped), dims s*KSTEP + g*KPL ..
  frag qf[QT][PL * CT::DSTEPS];   // split planes: [s] = hi, [DSTEPS + s] = lo of the fp32 query row
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    int qi = t * 16 + li;
    if (qi > nrows - 1) qi = nrows - 1;
    if constexpr (PL == 2) {
      const float* qp = reinterpret_cast<const float*>(p.q) + (int64_t)(row0 + qi) * p.ldq + h * DH + g * 8;
#pragma unroll
      for (int s = 0; s < CT::DSTEPS; ++s) {
        float tmp[8];
        unpack16(ld16<float>(qp + s * 32), tmp);
        unpack16(ld16<float>(qp + s * 32 + 4), tmp + 4);
#pragma unroll
        for (int i = 0; i < 8; ++i) tmp[i] *= 0.125f;   // a power of two: commutes with the split
        split8(tmp, qf[t][s], qf[t][CT::DSTEPS + s]);
      }
    } else {
      const E* qp = reinterpret_cast<const E*>(p.q) + (int64_t)(row0 + qi) * p.ldq + h * DH + g * MM::KPL;
#pragma unroll
      for (int s = 0; s < CT::DSTEPS; ++s) {
        float tmp[MM::KPL];
        unpack16(ld16<E>(qp + s * MM::KSTEP), tmp);
#pragma unroll
        for (int i = 0; i < MM::KPL; ++i) tmp[i] *= 0.125f;   // 1/sqrt(64) on q, like nn.MultiheadAttention
        pack16(tmp, qf[t][s]);
      }
    }
  }

  float m[QT], lpart[QT];
  f32x4 ot[QT][4];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    m[t] = -INFINITY; lpart[t] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) ot[t][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  auto compute = [&](int k0, const frag* kc, const frag* vc) {
    // which of this lane's 4 (8) keys are masked: key padding mask + the end of the slice
    bool dead[CT::NSB * 4];
#pragma unroll
    for (int sb = 0; sb < CT::NSB; ++sb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kk = k0 + sb * 16 + g * 4 + r;
        dead[sb * 4 + r] = kk >= kend || (km != nullptr && km[kk < p.M ? kk : p.M - 1]);
      }
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      if (t < nqt) {   // wave-uniform: the matrix-core ops below always run with a full EXEC mask
        float sc[CT::NSB * 4];
        float bmax = -INFINITY;
#pragma unroll
        for (int sb = 0; sb < CT::NSB; ++sb) {
          f32x4 st = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < CT::DSTEPS; ++s) MM::mma(st, kc[sb * CT::DSTEPS + s], qf[t][s]);
          if constexpr (PL == 2) {   // + K_lo q_hi + K_hi q_lo (K_lo q_lo is 2^-16 relative: dropped)
#pragma unroll
            for (int s = 0; s < CT::DSTEPS; ++s) {
              MM::mma(st, kc[NKF + sb * CT::DSTEPS + s], qf[t][s]);
              MM::mma(st, kc[sb * CT::DSTEPS + s], qf[t][CT::DSTEPS + s]);
            }
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = dead[sb * 4 + r] ? -INFINITY : st[r];
            sc[sb * 4 + r] = v;
            bmax = fmaxf(bmax, v);
          }
        }
        bmax = quad_group_max(bmax);   // register swaps, no LDS round trip (common.h)
        const float mn = fmaxf(m[t], bmax);
        // If every key seen so far is masked (mn = -inf) use 0 as the reference so exp(-inf - 0) = 0, not NaN.
        const float mref = (mn == -INFINITY) ? 0.f : mn;
        const float alpha = expf(m[t] - mref);
        float ps = 0.f;
#pragma unroll
        for (int i = 0; i < CT::NSB * 4; ++i) { sc[i] = expf(sc[i] - mref); ps += sc[i]; }
        lpart[t] = lpart[t] * alpha + ps;
        m[t] = mn;
        if constexpr (PL == 2) {
          frag ph, plo;
          split8(sc, ph, plo);
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) ot[t][dt][r] *= alpha;
            MM::mma(ot[t][dt], vc[dt], ph);
            MM::mma(ot[t][dt], vc[4 + dt], ph);
            MM::mma(ot[t][dt], vc[dt], plo);
          }
        } else {
          const frag pf = CT::pfrag(sc);
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) ot[t][dt][r] *= alpha;
            MM::mma(ot[t][dt], vc[dt], pf);
          }
        }
      }
    }
  };

  for (int b0 = 0; b0 < nblk; b0 += PD) {
#pragma unroll
    for (int u = 0; u < PD; ++u) {
      const int blk = b0 + u;
      if (blk < nblk) {
        compute(kbeg + blk * KB, kr[u], vr[u]);
        if (blk + PD < nblk) load_block(kbeg + (blk + PD) * KB, kr[u], vr[u]);
      }
    }
  }

  // merge the NW key slices of this (group, head) in LDS
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    if (t < nqt) {
      const float l = quad_group_sum(lpart[t]);
      float* mine = part + ((wave * QT + t) * 16 + li) * PSTR;
      if (g == 0) { mine[0] = m[t]; mine[1] = l; }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) *reinterpret_cast<f32x4*>(mine + 4 + dt * 16 + g * 4) = ot[t][dt];
    }
  }
  __syncthreads();
  for (int qi = wave; qi < nrows; qi += NW) {
    float mall = -INFINITY;
#pragma unroll
    for (int w = 0; w < NW; ++w) mall = fmaxf(mall, part[(w * QT * 16 + qi) * PSTR]);
    float L = 0.f, o = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float* pw = part + (w * QT * 16 + qi) * PSTR;
      const float wt = (pw[0] == -INFINITY) ? 0.f : expf(pw[0] - mall);
      L += pw[1] * wt;
      o += pw[4 + lane] * wt;
    }
    if (S == 1) {
      const float val = o / L;
      const int64_t at = (int64_t)(row0 + qi) * p.ldo + h * DH + lane;
      if constexpr (PL == 2) {
        if (p.lo_off > 0) {   // split-pair rows for the out-projection's bf16x3 product
          bf16_t* ob = reinterpret_cast<bf16_t*>(p.out) + at;
          const bf16_t hi = (bf16_t)val;
          ob[0] = hi;
          ob[p.lo_off] = (bf16_t)(val - (float)hi);
        } else {
          reinterpret_cast<float*>(p.out)[at] = val;
        }
      } else {
        reinterpret_cast<E*>(p.out)[at] = from_f32<E>(val);
      }
    } else {
      float* dst = p.partial + (((int64_t)(row0 + qi) * p.nH + h) * S + sp) * PSTR;
      if (lane == 0) { dst[0] = mall; dst[1] = L; }
      dst[4 + lane] = o;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// cross attention, many query rows per image (33..64: the polygon / recognition phases), bf16.
//
// dec_cross_attn_kernel gives every wave ALL query tiles and a private key slice: with 4 tiles a wave has
// room for only two key blocks in flight and serialises 4x(QK^T, softmax, PV) behind every block -- it is
// latency bound (1.8 TB/s).  Here the decomposition is turned around: the 4 waves of a workgroup each own ONE
// tile of 16 queries and all consume the SAME key stream, which is staged once per workgroup in an LDS ring by
// DMA (global_load_lds_dwordx4): NS blocks of 32 keys (4 KB of K + 4 KB of V^T each) are in flight per
// workgroup without costing a register, every byte of K/V still crosses HBM->CU once per (image, head), and
// no cross-wave merge is needed (a wave owns its queries; only the S workgroup-level key splits meet, in
// dec_cross_merge_kernel).  LDS image of a block = 32 rows x 128 B for K (row = key) and for V^T (row = two
// consecutive dims x 32 key slots); a DMA instruction fills 8 rows lane-linearly, so the XOR swizzle that
// makes the 16-byte fragment reads conflict-free (slot = chunk ^ (row & 7)) is applied to the per-lane
// SOURCE address.  The ring is consumed in CHUNKS of CH blocks (64 keys at CH = 2): wait own DMA of chunk c
// (counted vmcnt, later chunks stay in flight) -> one s_barrier (everybody's part of c has landed, everybody is
// done with c-1) -> refill the stages of c-1 with chunk c+D -> compute c with ONE running-max update, one
// accumulator rescale and one cross-lane maximum (v_permlane swaps, no LDS round trip) per chunk.
// ---------------------------------------------------------------------------------------------
// T = bf16 (32-key blocks) or float (16-key blocks: the fp32 slabs of the fp32 and bf16x3 engines; the register-streaming
// kernel reached 1.6 TB/s there -- 1.66 ms per 160-image launch, 31 % of the parity engine's time, profiles/r03e_*).  A ring
// stage is 8 KB either way (K then V^T); what differs is the LDS image of a block and its conflict-free swizzle:
//   bf16: K = 32 rows (keys) x 128 B, V^T = 32 rows (two dims x 32 key slots) x 128 B, slot = chunk ^ (row & 7);
//   f32 : K = 16 rows (keys) x 256 B, slot = chunk ^ row (16 chunks of 16 B); V^T = 64 rows (dims) x 64 B, slot = chunk ^ t(row >> 2
//         & 3) with t = {0, 2, 3, 1}: the four rows a ds_read_b128 lane group touches in one 256-byte bank row get four
//         different slots.
//   split planes (bf16s_t): a stage is 16 KB = K_hi | K_lo | V^T_hi | V^T_lo, each plane the bf16 image above (same swizzle); 4 DMA
//         instructions per wave and block; the K_hi / V^T_hi fragments are read from LDS once and used in two products.
template <typename T, int NS, int CH, bool NT>
__global__ __launch_bounds__(256) void dec_cross_attn_q4_kernel(CrossP p) {
  typedef CrossTraits<T> CT;
  typedef typename CT::elem E;
  typedef Mma<E> MM;
  typedef typename MM::frag frag;
  constexpr bool F32 = sizeof(E) == 4;
  constexpr int PL = CT::PL;
  constexpr int KB = CT::KB, BLKB = 8192 * PL;   // keys per block; bytes per ring stage (K planes then V^T planes)
  constexpr int VOFF = 4096 * PL;                // V^T planes start here inside a stage
  constexpr int SPB = KB / 4;               // scores per lane per block
  constexpr int QS = CT::DSTEPS;            // k-steps over the 64 head dims
  constexpr int SLOTS = NS / CH, D = SLOTS - 1;   // a chunk = CH blocks; D chunks in flight ahead of the one in use
  static_assert(NS % CH == 0 && D >= 1 && 2 * PL * CH * (D - 1) <= 56, "ring geometry");
  extern __shared__ __attribute__((aligned(16))) char ring[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, g = lane >> 4;
  const int grp = blockIdx.x, h = blockIdx.y, sp = blockIdx.z, S = gridDim.z;
  const int row0 = p.groups[grp * 3], nrows = p.groups[grp * 3 + 1], img = p.groups[grp * 3 + 2];
  const bool active = wave * 16 < nrows;          // wave-uniform: this wave's query tile exists
  const int kbeg = sp * p.kpw;                    // kpw: keys per workgroup split, multiple of KB
  int kend = kbeg + p.kpw;
  if (kend > p.M) kend = p.M;
  const int nblk = kbeg < kend ? (kend - kbeg + KB - 1) / KB : 0;
  const int nchunk = (nblk + CH - 1) / CH;
  auto tsw = [](int q) -> int { return (0x1320 >> (q * 4)) & 3; };   // t = {0, 2, 3, 1}

  const int64_t slab = (int64_t)img * p.img_stride + (int64_t)h * p.Mpad * 64 * PL;
  // DMA: this wave's instruction fills bytes [1024 * wave, + 1024) of the K image and of the V^T image of a block (of each plane),
  // lane-linearly; the swizzle is applied to the per-lane SOURCE address
  const E* ksrc;
  const E* vsrc;
  if constexpr (F32) {
    const int kr = wave * 4 + (lane >> 4), kc = (lane & 15) ^ kr;                 // K row (key) and the chunk that lands in this lane's slot
    const int vr = wave * 16 + (lane >> 2), vc_ = (lane & 3) ^ tsw((lane >> 4) & 3);   // V^T row (dim): (vr >> 2) & 3 == (lane >> 4) & 3
    ksrc = reinterpret_cast<const E*>(p.K) + slab + kr * 64 + kc * 4;
    vsrc = reinterpret_cast<const E*>(p.V) + slab + vr * 16 + vc_ * 4;
  } else {
    const int dr = lane >> 3, dc = (lane & 7) ^ dr;
    ksrc = reinterpret_cast<const E*>(p.K) + slab + (int64_t)(wave * 8 + dr) * 64 + dc * 8;
    vsrc = reinterpret_cast<const E*>(p.V) + slab + (int64_t)(wave * 8 + dr) * 64 + dc * 8;
  }
  // chunk c -> ring stages slot * CH .. slot * CH + CH - 1.  A ragged last chunk re-requests the last valid block for
  // its missing ones: their keys are masked below, but their V^T image must hold finite numbers (0 x NaN = NaN)
  auto issue_chunk = [&](int c, int slot) {
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      int blk = c * CH + j;
      if (blk > nblk - 1) blk = nblk - 1;
      const int64_t off = (int64_t)(kbeg + blk * KB) * 64 * PL;   // a block is KB * 64 * PL elements in both slabs
      char* dst = ring + (slot * CH + j) * BLKB + wave * 1024;
      // NT: the slabs are read once per step and never fit a cache at 256 images per call -> non-temporal (aux = 2)
#pragma unroll
      for (int pl = 0; pl < PL; ++pl) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ksrc + off + pl * KB * 64),
                                         (__attribute__((address_space(3))) void*)(dst + pl * 4096), 16, 0, NT ? 2 : 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vsrc + off + pl * KB * 64),
                                         (__attribute__((address_space(3))) void*)(dst + VOFF + pl * 4096), 16, 0, NT ? 2 : 0);
      }
    }
  };
#pragma unroll
  for (int t = 0; t < D; ++t)
    if (t < nchunk) issue_chunk(t, t);

  // fragment byte offsets inside a stage
  int koff[CT::NSB][QS], voff[4];
  if constexpr (F32) {
#pragma unroll
    for (int st = 0; st < QS; ++st) koff[0][st] = li * 256 + (((st * 4 + g) ^ li) << 4);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) voff[dt] = VOFF + (dt * 16 + li) * 64 + ((g ^ tsw(li >> 2)) << 4);
  } else {
#pragma unroll
    for (int sb = 0; sb < CT::NSB; ++sb)
#pragma unroll
      for (int st = 0; st < QS; ++st) koff[sb][st] = (sb * 16 + li) * 128 + (((st * 4 + g) ^ (li & 7)) << 4);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const int rw = dt * 8 + (li >> 1), c = (li & 1) * 4 + g;
      voff[dt] = VOFF + rw * 128 + ((c ^ (rw & 7)) << 4);
    }
  }

  // Q fragments (B operand) of this wave's tile, 1/sqrt(64) folded in (split planes: [st] = hi, [QS + st] = lo of the fp32 row)
  frag qf[PL * QS];
  {
    int qi = wave * 16 + li;
    if (qi > nrows - 1) qi = nrows - 1;
    if constexpr (PL == 2) {
      const float* qp = reinterpret_cast<const float*>(p.q) + (int64_t)(row0 + qi) * p.ldq + h * DH + g * 8;
#pragma unroll
      for (int st = 0; st < QS; ++st) {
        float tmp[8];
        unpack16(ld16<float>(qp + st * 32), tmp);
        unpack16(ld16<float>(qp + st * 32 + 4), tmp + 4);
#pragma unroll
        for (int i = 0; i < 8; ++i) tmp[i] *= 0.125f;
        split8(tmp, qf[st], qf[QS + st]);
      }
    } else {
      const E* qp = reinterpret_cast<const E*>(p.q) + (int64_t)(row0 + qi) * p.ldq + h * DH + g * MM::KPL;
#pragma unroll
      for (int st = 0; st < QS; ++st) {
        float tmp[MM::KPL];
        unpack16(ld16<E>(qp + st * MM::KSTEP), tmp);
#pragma unroll
        for (int i = 0; i < MM::KPL; ++i) tmp[i] *= 0.125f;
        pack16(tmp, qf[st]);
      }
    }
  }
  const uint8_t* km = p.kmask ? p.kmask + (int64_t)img * p.M : nullptr;

  float m = -INFINITY, lpart = 0.f;
  f32x4 ot[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) ot[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // key of score i of a chunk: block i / SPB, inside it (bf16: second 16-key half, then) 4 g + r
  auto key_of = [&](int k0, int i) -> int { return k0 + (i / SPB) * KB + (((i % SPB) >> 2) & 1) * 16 + g * 4 + (i & 3); };

  int sl_c = 0, sl_i = D;   // slot in use, slot to refill (the one used by the previous chunk)
  for (int c = 0; c < nchunk; ++c) {
    const int after = nchunk - 1 - c;
    wait_dma_blocks<2 * PL * CH>(after < D - 1 ? after : D - 1);
    __builtin_amdgcn_s_barrier();
    if (c + D < nchunk) issue_chunk(c + D, sl_i);
    if (active) {
      float sc[CH * SPB];
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const char* base = ring + (sl_c * CH + j) * BLKB;
#pragma unroll
        for (int sb = 0; sb < CT::NSB; ++sb) {
          f32x4 sacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int st = 0; st < QS; ++st) {
            const frag kh = *reinterpret_cast<const frag*>(base + koff[sb][st]);
            MM::mma(sacc, kh, qf[st]);
            if constexpr (PL == 2) {   // + K_lo q_hi + K_hi q_lo
              MM::mma(sacc, *reinterpret_cast<const frag*>(base + 4096 + koff[sb][st]), qf[st]);
              MM::mma(sacc, kh, qf[QS + st]);
            }
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) sc[j * SPB + sb * 4 + r] = sacc[r];
        }
      }
      // masking, branch-free per element (both conditions are wave-uniform): key padding mask bytes are all
      // loaded before any is used; the ragged tail of the split needs only the index compare
      const int k0 = kbeg + c * CH * KB;
      if (km != nullptr) {
        uint8_t mk[CH * SPB];
#pragma unroll
        for (int i = 0; i < CH * SPB; ++i) {
          const int kk = key_of(k0, i);
          mk[i] = km[kk < p.M ? kk : p.M - 1];
        }
#pragma unroll
        for (int i = 0; i < CH * SPB; ++i) {
          const int kk = key_of(k0, i);
          const bool dead = (kk >= kend) | (mk[i] != 0);
          sc[i] = dead ? -INFINITY : sc[i];
        }
      } else if (k0 + CH * KB > kend) {
#pragma unroll
        for (int i = 0; i < CH * SPB; ++i) {
          const int kk = key_of(k0, i);
          sc[i] = (kk >= kend) ? -INFINITY : sc[i];
        }
      }
      float bmax = sc[0];
#pragma unroll
      for (int i = 1; i < CH * SPB; ++i) bmax = fmaxf(bmax, sc[i]);
      bmax = quad_group_max(bmax);        // the 4 lane groups hold different keys of the same query
      const float mn = fmaxf(m, bmax);
      const float mref = (mn == -INFINITY) ? 0.f : mn;   // all keys so far masked: exp(-inf - 0) = 0, not NaN
      float alpha, ps = 0.f;
      if constexpr (F32 || PL == 2) {     // the fp32-grade engines keep the accurate exponential (their parity gate is 1e-3 on logits)
        alpha = expf(m - mref);
#pragma unroll
        for (int i = 0; i < CH * SPB; ++i) { sc[i] = expf(sc[i] - mref); ps += sc[i]; }
      } else {
        alpha = __expf(m - mref);
#pragma unroll
        for (int i = 0; i < CH * SPB; ++i) { sc[i] = __expf(sc[i] - mref); ps += sc[i]; }
      }
      lpart = lpart * alpha + ps;
      m = mn;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) ot[dt][r] *= alpha;
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const char* base = ring + (sl_c * CH + j) * BLKB;
        if constexpr (PL == 2) {
          frag ph, plo;
          split8(sc + j * SPB, ph, plo);
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            const frag vh = *reinterpret_cast<const frag*>(base + voff[dt]);
            MM::mma(ot[dt], vh, ph);
            MM::mma(ot[dt], *reinterpret_cast<const frag*>(base + 4096 + voff[dt]), ph);
            MM::mma(ot[dt], vh, plo);
          }
        } else {
          const frag pf = CT::pfrag(sc + j * SPB);
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            const frag vc = *reinterpret_cast<const frag*>(base + voff[dt]);
            MM::mma(ot[dt], vc, pf);
          }
        }
      }
    }
    sl_c = (sl_c + 1 == SLOTS) ? 0 : sl_c + 1;
    sl_i = (sl_i + 1 == SLOTS) ? 0 : sl_i + 1;
  }

  if (!active) return;
  const float l = quad_group_sum(lpart);
  const int qi = wave * 16 + li;
  if (qi >= nrows) return;
  if (S == 1) {
    const int64_t at = (int64_t)(row0 + qi) * p.ldo + h * DH + g * 4;
    const float inv = 1.0f / l;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const f32x4 v = {ot[dt][0] * inv, ot[dt][1] * inv, ot[dt][2] * inv, ot[dt][3] * inv};
      if constexpr (PL == 2) {
        if (p.lo_off > 0) {   // split-pair rows for the out-projection's bf16x3 product
          bf16_t* ob = reinterpret_cast<bf16_t*>(p.out) + at + dt * 16;
          bf16x4 hi, lo;
#pragma unroll
          for (int r = 0; r < 4; ++r) { hi[r] = (bf16_t)v[r]; lo[r] = (bf16_t)(v[r] - (float)hi[r]); }
          *reinterpret_cast<bf16x4*>(ob) = hi;
          *reinterpret_cast<bf16x4*>(ob + p.lo_off) = lo;
        } else {
          *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + at + dt * 16) = v;
        }
      } else if constexpr (F32) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + at + dt * 16) = v;
      } else {
        *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(p.out) + at + dt * 16) = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
      }
    }
  } else {
    float* dst = p.partial + (((int64_t)(row0 + qi) * p.nH + h) * S + sp) * CROSS_PSTR;
    if (g == 0) { dst[0] = m; dst[1] = l; }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) *reinterpret_cast<f32x4*>(dst + 4 + dt * 16 + g * 4) = ot[dt];
  }
}

// merge the S workgroup-level partials of every (row, head): one wave each, lane = output dim.
// All loads are issued before any math (S <= 16) -- the merge is a latency-, not a bandwidth problem.
template <typename T, int S>
__global__ __launch_bounds__(256) void dec_cross_merge_kernel(const float* __restrict__ partial, T* __restrict__ out,
                                                              int64_t ldo, int nH, int total, int lo_off) {
  const int lane = threadIdx.x & 63, wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= total) return;
  const int r = wid / nH, h = wid % nH;
  const float* base = partial + (int64_t)wid * S * 68;
  float mv[S], lv[S], ov[S];
#pragma unroll
  for (int s = 0; s < S; ++s) { mv[s] = base[s * 68]; lv[s] = base[s * 68 + 1]; ov[s] = base[s * 68 + 4 + lane]; }
  float mall = -INFINITY;
#pragma unroll
  for (int s = 0; s < S; ++s) mall = fmaxf(mall, mv[s]);
  float L = 0.f, o = 0.f;
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const float w = (mv[s] == -INFINITY) ? 0.f : expf(mv[s] - mall);
    L += lv[s] * w;
    o += ov[s] * w;
  }
  if constexpr (sizeof(T) == 4) {
    if (lo_off > 0) {   // split-bf16 pair rows (ldo in bf16 elements)
      bf16_t* ob = reinterpret_cast<bf16_t*>(out) + (int64_t)r * ldo + h * DH + lane;
      const float val = o / L;
      const bf16_t hi = (bf16_t)val;
      ob[0] = hi;
      ob[lo_off] = (bf16_t)(val - (float)hi);
      return;
    }
  }
  out[(int64_t)r * ldo + h * DH + lane] = from_f32<T>(o / L);
}

// ---------------------------------------------------------------------------------------------
// greedy sampling: softmax over the support, candidate filter, argmax, probability. wave per row.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool is_candidate(const omp_sample_cfg& c, int tok, int i) {
  if (c.kind == OMP_DEC_PT) {
    const int period = c.infer_vie ? 3 : 2;
    const int r = i % period;
    if (r == 0) return tok < c.num_bins || (tok == c.pt_eos && !c.suppress_eos);
    if (r == 1) return tok < c.num_bins;
    return tok >= c.vocab - c.vie_categories;
  }
  if (c.kind == OMP_DEC_POLY) return tok < c.num_bins;
  return tok >= c.num_bins && tok <= c.rec_eos && tok != c.pt_eos && tok != c.poly_eos;
}

__global__ __launch_bounds__(256) void dec_sample_kernel(const float* __restrict__ logits, int ld, int R,
                                                         omp_sample_cfg c, int32_t* __restrict__ seq,
                                                         float* __restrict__ probs, int seq_ld,
                                                         int32_t* __restrict__ finished,
                                                         int32_t* __restrict__ lengths,
                                                         const int32_t* __restrict__ d_pos) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = blockIdx.x * 4 + wave;
  if (r >= R) return;
  const int p = *d_pos;
  const int i = p + 1 - c.step0;
  if (i < 0) return;
  const bool slice = c.infer_vie && c.kind != OMP_DEC_PT;
  const int Vs = c.vocab - (slice ? c.vie_categories : 0);
  const float* lg = logits + (int64_t)r * ld;
  float mx = -INFINITY, best = -INFINITY;
  int bi = 0x7fffffff;
  for (int t = lane; t < Vs; t += 64) {
    const float v = lg[t];
    mx = fmaxf(mx, v);
    if (is_candidate(c, t, i) && v > best) { best = v; bi = t; }
  }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int t = lane; t < Vs; t += 64) sum += expf(lg[t] - mx);
  sum = wave_sum(sum);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  if (lane == 0) {
    seq[(int64_t)r * seq_ld + p + 1] = bi;
    probs[(int64_t)r * seq_ld + p + 1] = expf(best - mx) / sum;
    if (c.kind == OMP_DEC_PT && finished != nullptr) {
      // reference stops at the first EOS (transformer.py:126); later tokens of a finished row are ignored
      if (!finished[r] && bi == c.pt_eos) { finished[r] = 1; lengths[r] = p + 1; }
    }
  }
}

__global__ void advance_pos_kernel(int32_t* d_pos) { *d_pos += 1; }

// The same sampling with a WORKGROUP per row (4 waves x <= 5 logits per lane instead of one wave x 18: the wave-per-row kernel
// took 10.5 us of a point-decoder step) and the position advance folded in: every workgroup has read *d_pos before it
// takes a ticket, so the last ticket holder may publish pos + 1 for the next launch (no advance_pos launch): 5.3 us for both.
__global__ __launch_bounds__(256) void dec_sample_block_kernel(const float* __restrict__ logits, int ld, int R, omp_sample_cfg c,
                                                               int32_t* __restrict__ seq, float* __restrict__ probs, int seq_ld,
                                                               int32_t* __restrict__ finished, int32_t* __restrict__ lengths,
                                                               int32_t* d_pos, int32_t* ticket) {
  __shared__ float s_f[8];
  __shared__ int s_i[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = blockIdx.x;
  const int p = *d_pos;
  const int i = p + 1 - c.step0;
  if (i >= 0) {
    const bool slice = c.infer_vie && c.kind != OMP_DEC_PT;
    const int Vs = c.vocab - (slice ? c.vie_categories : 0);
    const float* lg = logits + (int64_t)r * ld;
    float mx = -INFINITY, best = -INFINITY;
    int bi = 0x7fffffff;
    for (int t = tid; t < Vs; t += 256) {
      const float v = lg[t];
      mx = fmaxf(mx, v);
      if (is_candidate(c, t, i) && v > best) { best = v; bi = t; }
    }
    mx = wave_max(mx);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) { s_f[wave] = mx; s_f[4 + wave] = best; s_i[wave] = bi; }
    __syncthreads();
    mx = fmaxf(fmaxf(s_f[0], s_f[1]), fmaxf(s_f[2], s_f[3]));
    best = s_f[4]; bi = s_i[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
      if (s_f[4 + w] > best || (s_f[4 + w] == best && s_i[w] < bi)) { best = s_f[4 + w]; bi = s_i[w]; }
    float sum = 0.f;
    for (int t = tid; t < Vs; t += 256) sum += expf(lg[t] - mx);
    sum = wave_sum(sum);
    __syncthreads();            // everybody has read s_f / s_i
    if (lane == 0) s_f[wave] = sum;
    __syncthreads();
    if (tid == 0) {
      sum = (s_f[0] + s_f[1]) + (s_f[2] + s_f[3]);
      seq[(int64_t)r * seq_ld + p + 1] = bi;
      probs[(int64_t)r * seq_ld + p + 1] = expf(best - mx) / sum;
      if (c.kind == OMP_DEC_PT && finished != nullptr) {
        if (!finished[r] && bi == c.pt_eos) { finished[r] = 1; lengths[r] = p + 1; }
      }
    }
  }
  if (ticket != nullptr && tid == 0) {
    const int t = atomicAdd(ticket, 1);
    if (t == (int)gridDim.x - 1) { *ticket = 0; *d_pos = p + 1; }
  }
}

// Many rows (the polygon / recognition phases: thousands of rows): a WAVE per row with the whole row in registers -- every lane requests its
// (up to) eight 16-byte pieces of the row before it looks at any of them, the candidate filter, the maximum and the exponentials run on those
// registers (the row is read once), and the position advance is folded in as in the block kernel.  The wave-per-row kernel above walks a row 64
// logits at a time, one dependent round trip each (27 us alone at 10 240 rows); the workgroup-per-row kernel pays 10 240 workgroups of three
// barriers (122 us alone; profiles/r05zh_kernel_shapes_bf16.txt).  ld % 4 == 0, vocab <= 2048.
__global__ __launch_bounds__(256) void dec_sample_rows_kernel(const float* __restrict__ logits, int ld, int R, omp_sample_cfg c,
                                                              int32_t* __restrict__ seq, float* __restrict__ probs, int seq_ld,
                                                              int32_t* __restrict__ finished, int32_t* __restrict__ lengths,
                                                              int32_t* d_pos, int32_t* ticket) {
  constexpr int NQ = 8;   // 16-byte pieces per lane: 8 x 256 = 2048 logits
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = blockIdx.x * 4 + wave;
  const int p = *d_pos;
  const int i = p + 1 - c.step0;
  if (i >= 0 && r < R) {
    const bool slice = c.infer_vie && c.kind != OMP_DEC_PT;
    const int Vs = c.vocab - (slice ? c.vie_categories : 0);
    const float* lg = logits + (int64_t)r * ld;
    f32x4 v[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int t0 = q * 256 + lane * 4;
      v[q] = t0 < ld ? *reinterpret_cast<const f32x4*>(lg + t0) : f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};   // ld % 4 == 0: a piece is inside the row or not at all
    }
    float mx = -INFINITY, best = -INFINITY;
    int bi = 0x7fffffff;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int t = q * 256 + lane * 4 + e;
        if (t < Vs) {
          mx = fmaxf(mx, v[q][e]);
          if (is_candidate(c, t, i) && v[q][e] > best) { best = v[q][e]; bi = t; }   // ascending t per lane: the first maximum wins, as in the kernels above
        }
      }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (q * 256 + lane * 4 + e < Vs) sum += expf(v[q][e] - mx);
    sum = wave_sum(sum);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) {
      seq[(int64_t)r * seq_ld + p + 1] = bi;
      probs[(int64_t)r * seq_ld + p + 1] = expf(best - mx) / sum;
      if (c.kind == OMP_DEC_PT && finished != nullptr) {
        if (!finished[r] && bi == c.pt_eos) { finished[r] = 1; lengths[r] = p + 1; }
      }
    }
  }
  if (ticket != nullptr) {
    __syncthreads();   // the four rows of this workgroup are done (every wave has read *d_pos long before)
    if (tid == 0) {
      const int t = atomicAdd(ticket, 1);
      if (t == (int)gridDim.x - 1) { *ticket = 0; *d_pos = p + 1; }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Text-spotting results of an engine call -> fixed-size padded tensors (the payload of the per-call all-gather of an image-
// sharded deployment): one launch instead of a Python loop of ~12 tiny device copies per image.
// ids[b, n, :] = point (2) | polygon (32) | recognition (rec_len) tokens of instance n of image b (rows row0[b] .. + cnt[b],
// capped at N), zero padded; probs[b, n, :] = its recognition probabilities.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void pack_spotting_kernel(const int32_t* __restrict__ points, const int32_t* __restrict__ poly, int poly_ld,
                                                          const int32_t* __restrict__ rec, int rec_ld, const float* __restrict__ rprob,
                                                          int prob_ld, const int32_t* __restrict__ row0, const int32_t* __restrict__ cnt,
                                                          int N, int rec_len, int32_t* __restrict__ ids, float* __restrict__ probs) {
  const int n = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
  const int W = 34 + rec_len;
  const bool live = n < cnt[b];
  const int64_t r = (int64_t)row0[b] + n;
  int32_t* io = ids + ((int64_t)b * N + n) * W;
  float* po = probs + ((int64_t)b * N + n) * rec_len;
  for (int c = t; c < W; c += 64) {
    int32_t v = 0;
    if (live) v = c < 2 ? points[r * 2 + c] : (c < 34 ? poly[r * poly_ld + (c - 2)] : rec[r * rec_ld + (c - 34)]);
    io[c] = v;
  }
  for (int c = t; c < rec_len; c += 64) po[c] = live ? rprob[r * prob_ld + c] : 0.f;
}

// ---------------------------------------------------------------------------------------------
// host-side launch helpers
// ---------------------------------------------------------------------------------------------
// optional hipEvent bracketing of the cross-attention kernel (bench.py's roofline measurement)
thread_local bool g_capturing = false;   // one host thread per pipeline lane may be capturing
// The capture gate (round 6).  omp_decoder_run captures its graphs on the CALLER's stream (a graph captured on another stream and launched on
// the caller's was measured: the polygon || recognition graphs then serialise, 100 -> 120 ms per 160 images, profiles/r06ze_*).  While a stream
// captures, HIP refuses -- and INVALIDATES the capture on -- hipEventSynchronize / hipEventQuery / hipStreamWaitEvent of any event last recorded
// in that stream, completion events of EARLIER, uncaptured work included; another host thread waiting for a pipeline lane's previous call does
// exactly that (tests/test_gpu_e2e.py::test_pipelined_lanes_match_direct: hipErrorCapturedEvent once in nine full suites).  A capture therefore
// holds this process-wide mutex from begin to end, and host threads take it (omp_capture_gate_enter / _leave, include/omp355.h) around
// NON-BLOCKING event operations on another thread's streams.  Graphs are captured once per slot: steady-state calls never touch the mutex.
static std::mutex g_capture_gate;


template <typename T, int QT, int PD, bool NT = false>
int launch_cross_t(const CrossP& cp, int n_groups, int S, hipStream_t st) {
  constexpr int NW = 4;
  const size_t smem = (size_t)NW * QT * 16 * CROSS_PSTR * sizeof(float);
  auto kern = dec_cross_attn_kernel<T, NW, QT, PD, NT>;
  if (smem > 48 * 1024) {
    static bool done = false;   // per template instantiation
    if (!done) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
        omp_set_error("omp_dec_cross_attn_step: cannot raise dynamic LDS limit");
        return OMP_ERR_LAUNCH;
      }
      done = true;
    }
  }
  hipLaunchKernelGGL(kern, dim3(n_groups, cp.nH, S), dim3(NW * 64), smem, st, cp);
  return OMP_OK;
}

template <typename T>
int launch_merge(const CrossP& cp, int S, hipStream_t st) {
  const int total = cp.R * cp.nH;
  dim3 grid((total + 3) / 4), block(256);
  T* out = reinterpret_cast<T*>(cp.out);
  switch (S) {
    case 2: hipLaunchKernelGGL((dec_cross_merge_kernel<T, 2>), grid, block, 0, st, cp.partial, out, cp.ldo, cp.nH, total, cp.lo_off); break;
    case 4: hipLaunchKernelGGL((dec_cross_merge_kernel<T, 4>), grid, block, 0, st, cp.partial, out, cp.ldo, cp.nH, total, cp.lo_off); break;
    case 8: hipLaunchKernelGGL((dec_cross_merge_kernel<T, 8>), grid, block, 0, st, cp.partial, out, cp.ldo, cp.nH, total, cp.lo_off); break;
    case 16: hipLaunchKernelGGL((dec_cross_merge_kernel<T, 16>), grid, block, 0, st, cp.partial, out, cp.ldo, cp.nH, total, cp.lo_off); break;
    default: omp_set_error("omp_dec_cross_attn_step: unsupported workgroup split %d", S); return OMP_ERR_INVALID;
  }
  return OMP_OK;
}


template <typename T, int NS, int CH, bool NT>
int launch_cross_q4(const CrossP& cp, int n_groups, int S, hipStream_t st) {
  const size_t smem = (size_t)NS * 8192 * CrossTraits<T>::PL;
  auto kern = dec_cross_attn_q4_kernel<T, NS, CH, NT>;
  static bool done = false;   // per template instantiation
  if (!done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
      omp_set_error("omp_dec_cross_attn_step: cannot raise dynamic LDS limit");
      return OMP_ERR_LAUNCH;
    }
    done = true;
  }
  hipLaunchKernelGGL(kern, dim3(n_groups, cp.nH, S), dim3(256), smem, st, cp);
  return OMP_OK;
}

// S = workgroup-level key splits (power of two <= 16); each workgroup's 4 waves split its keys again.
// qt = query tiles (of 16 rows) per group: 1, 2 or 4.
int launch_cross(CrossP cp, int n_groups, int dtype, int S, int qt, hipStream_t st) {
  if (S < 1 || S > 16 || (S & (S - 1)) != 0) { omp_set_error("omp_dec_cross_attn_step: n_split must be a power of two in [1,16] (got %d)", S); return OMP_ERR_INVALID; }
  if (qt != 1 && qt != 2 && qt != 4) { omp_set_error("omp_dec_cross_attn_step: q_tiles must be 1, 2 or 4 (got %d)", qt); return OMP_ERR_INVALID; }
  if (S > 1 && cp.partial == nullptr) { omp_set_error("omp_dec_cross_attn_step: n_split %d needs a partial buffer", S); return OMP_ERR_INVALID; }
  const int KB = dtype == OMP_F32 ? 16 : 32;
  if (cp.Mpad % KB != 0 || cp.Mpad < cp.M) { omp_set_error("omp_dec_cross_attn_step: Mpad %d must be a multiple of %d and >= M", cp.Mpad, KB); return OMP_ERR_INVALID; }
  const omp_ctx& cx = omp_cur();
  const bool q4 = cx.cross_q4 != 0 && qt == 4;   // waves own query tiles, not key slices (bf16 and fp32 slabs alike)
  const int slices = q4 ? S : S * 4;
  cp.kpw = (((cp.M + slices - 1) / slices + KB - 1) / KB) * KB;
  const bool prof = omp_prof_active(OMP_PROF_CROSS) && !g_capturing;
  int prof_slot = -1;
  if (prof) {
    prof_slot = omp_prof_begin(OMP_PROF_CROSS, st, 0.0);   // the caller knows how many images share the launch: bytes are computed there
  }
  int rc;
  const bool f = dtype == OMP_F32;
  const bool nt_on = cx.cross_nt == 1 || (cx.cross_nt == 2 && n_groups >= 32);
  if (dtype == OMP_BF16X2) {
    // split-plane slabs (the parity engine): fp32 q / out, three bf16 products per score / value block.  A ring stage is 16 KB;
    // the register-streaming kernel keeps 2 blocks (32 KB) in flight per wave.
    if (q4) {
      // ring geometry, measured at 160 images x 64 rows (profiles/r04h_kbench_cross_split_rings.txt, r04i_*): the more workgroups per CU the
      // better -- TWO one-block stages (32 KB, five workgroups per CU) 474 us = 0.72 of HBM, three stages 481-489, four 528, eight stages
      // in 64-key chunks (one workgroup) 577-587
      if (cx.cross_q4 == 4) rc = nt_on ? launch_cross_q4<bf16s_t, 8, 2, true>(cp, n_groups, S, st) : launch_cross_q4<bf16s_t, 8, 2, false>(cp, n_groups, S, st);
      else if (cx.cross_q4 == 5) rc = launch_cross_q4<bf16s_t, 4, 1, true>(cp, n_groups, S, st);   // A/B: four stages (64 KB, two workgroups per CU)
      else if (cx.cross_q4 == 6) rc = launch_cross_q4<bf16s_t, 3, 1, true>(cp, n_groups, S, st);   // A/B: three stages (48 KB, three workgroups per CU)
      else rc = nt_on ? launch_cross_q4<bf16s_t, 2, 1, true>(cp, n_groups, S, st) : launch_cross_q4<bf16s_t, 2, 1, false>(cp, n_groups, S, st);   // two stages: 32 KB, five workgroups per CU
    }
    else if (qt == 1) rc = nt_on ? launch_cross_t<bf16s_t, 1, 2, true>(cp, n_groups, S, st) : launch_cross_t<bf16s_t, 1, 2>(cp, n_groups, S, st);
    else if (qt == 2) rc = launch_cross_t<bf16s_t, 2, 2>(cp, n_groups, S, st);
    else rc = launch_cross_t<bf16s_t, 4, 1>(cp, n_groups, S, st);
    if (rc != OMP_OK) return rc;
    if (prof) omp_prof_end(OMP_PROF_CROSS, prof_slot, st);
    OMP_CHECK_LAUNCH("omp_dec_cross_attn_step(split planes)");
    if (S > 1) {
      rc = launch_merge<float>(cp, S, st);
      if (rc != OMP_OK) return rc;
      OMP_CHECK_LAUNCH("omp_dec_cross_attn_step(merge)");
    }
    return OMP_OK;
  }
  // PD key blocks in flight per wave: with one query tile a wave's whole slice is usually 4 blocks -> all of it
  if (q4) {
    const bool nt = nt_on;
    switch (cx.cross_q4) {
      case 2: rc = f ? launch_cross_q4<float, 8, 1, false>(cp, n_groups, S, st) : launch_cross_q4<bf16_t, 8, 1, false>(cp, n_groups, S, st); break;    // one block per step (round-2 start)
      case 4: rc = f ? launch_cross_q4<float, 8, 2, false>(cp, n_groups, S, st) : launch_cross_q4<bf16_t, 8, 2, false>(cp, n_groups, S, st); break;    // A/B: chunks with temporal loads (an 80 KB ring, 4 chunks ahead, measured equal: r02y)
      case 5: rc = f ? launch_cross_q4<float, 6, 2, true>(cp, n_groups, S, st) : launch_cross_q4<bf16_t, 8, 2, true>(cp, n_groups, S, st); break;      // A/B (round 4): fp32 48 KB ring; bf16: the 64 KB ring of rounds 2-3
      case 6: rc = f ? launch_cross_q4<float, 4, 1, true>(cp, n_groups, S, st) : launch_cross_q4<bf16_t, 4, 1, true>(cp, n_groups, S, st); break;      // A/B (round 4): 32 KB ring of one-block stages, five workgroups per CU
      default:
        // round 4: the bf16 ring at 48 KB (three workgroups per CU) 223.8 vs 230.8 us at 64 KB per 160 images (profiles/r04i_*)
        if (f) rc = nt ? launch_cross_q4<float, 8, 2, true>(cp, n_groups, S, st) : launch_cross_q4<float, 8, 2, false>(cp, n_groups, S, st);
        else rc = nt ? launch_cross_q4<bf16_t, 6, 2, true>(cp, n_groups, S, st) : launch_cross_q4<bf16_t, 8, 2, false>(cp, n_groups, S, st);
    }
  }
  else if (qt == 1) rc = f ? launch_cross_t<float, 1, 4>(cp, n_groups, S, st) : ((cx.cross_nt == 1 || (cx.cross_nt == 2 && n_groups >= 32)) ? launch_cross_t<bf16_t, 1, 4, true>(cp, n_groups, S, st) : launch_cross_t<bf16_t, 1, 4>(cp, n_groups, S, st));
  else if (qt == 2) rc = f ? launch_cross_t<float, 2, 2>(cp, n_groups, S, st) : launch_cross_t<bf16_t, 2, 2>(cp, n_groups, S, st);
  else rc = f ? launch_cross_t<float, 4, 2>(cp, n_groups, S, st) : launch_cross_t<bf16_t, 4, 2>(cp, n_groups, S, st);
  if (rc != OMP_OK) return rc;
  if (prof) omp_prof_end(OMP_PROF_CROSS, prof_slot, st);
  OMP_CHECK_LAUNCH("omp_dec_cross_attn_step");
  if (S > 1) {
    rc = f ? launch_merge<float>(cp, S, st) : launch_merge<bf16_t>(cp, S, st);
    if (rc != OMP_OK) return rc;
    OMP_CHECK_LAUNCH("omp_dec_cross_attn_step(merge)");
  }
  return OMP_OK;
}

}  // namespace

extern "C" int omp_pack_spotting(const int32_t* points, const int32_t* poly, int poly_ld, const int32_t* rec, int rec_ld,
                                 const float* rec_probs, int prob_ld, const int32_t* row0, const int32_t* counts, int B, int N,
                                 int rec_len, int32_t* ids, float* probs, omp_stream_t s) {
  OMP_CHECK_ARG(points && poly && rec && rec_probs && row0 && counts && ids && probs, "omp_pack_spotting: null pointer");
  OMP_CHECK_ARG(B > 0 && N > 0 && rec_len > 0 && poly_ld >= 32 && rec_ld >= rec_len && prob_ld >= rec_len, "omp_pack_spotting: bad shape");
  hipLaunchKernelGGL(pack_spotting_kernel, dim3(N, B), dim3(64), 0, (hipStream_t)s, points, poly, poly_ld, rec, rec_ld, rec_probs, prob_ld,
                     row0, counts, N, rec_len, ids, probs);
  OMP_CHECK_LAUNCH("omp_pack_spotting");
  return OMP_OK;
}

extern "C" int omp_dec_embed_ln(const int32_t* seq, int seq_ld, const int32_t* d_pos, const float* word_emb,
                                const float* pos_tab, const float* gamma, const float* beta, float* x,
                                void* y, int y_dtype, int R, int d, float eps, omp_stream_t s) {
  OMP_CHECK_ARG(seq && d_pos && word_emb && pos_tab && gamma && beta && (x || y), "omp_dec_embed_ln: null pointer");
  OMP_CHECK_ARG(R > 0 && d > 0 && d % 4 == 0 && d <= 1024, "omp_dec_embed_ln: bad shape R=%d d=%d", R, d);
  dim3 grid((R + 3) / 4);
  if (y_dtype == OMP_BF16 && y != nullptr)
    hipLaunchKernelGGL((dec_embed_ln_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)s, seq, seq_ld, d_pos,
                       word_emb, pos_tab, gamma, beta, x, (bf16_t*)y, R, d, eps);
  else
    hipLaunchKernelGGL((dec_embed_ln_kernel<float>), grid, dim3(256), 0, (hipStream_t)s, seq, seq_ld, d_pos,
                       word_emb, pos_tab, gamma, beta, x, (float*)y, R, d, eps);
  OMP_CHECK_LAUNCH("omp_dec_embed_ln");
  return OMP_OK;
}

extern "C" int omp_dec_self_attn_step(const void* qkv, void* kcache, void* vcache, void* out,
                                      const int32_t* d_pos, int dtype, int R, int nH, int d, int Lmax,
                                      omp_stream_t s) {
  OMP_CHECK_ARG(qkv && kcache && vcache && out && d_pos, "omp_dec_self_attn_step: null pointer");
  OMP_CHECK_ARG(d == nH * DH, "omp_dec_self_attn_step: head_dim must be 64 (d=%d nH=%d)", d, nH);
  OMP_CHECK_ARG(dtype == OMP_F32 || dtype == OMP_BF16, "omp_dec_self_attn_step: bad dtype");
  // many rows, short caches (polygon / recognition): one wave per row; few rows, long caches (points): one wave per (row, head)
  const int impl = omp_cur().self_attn_impl;
  const bool rows = impl == 2 || (impl == 0 && nH == 8 && R >= 1024);
  if (rows) {
    OMP_CHECK_ARG(nH == 8, "omp_dec_self_attn_step: the row kernel covers d = 512 (8 heads)");
    dim3 g4((R + 3) / 4);
    if (dtype == OMP_F32)
      hipLaunchKernelGGL((dec_self_attn_row_kernel<float>), g4, dim3(256), 0, (hipStream_t)s, (const float*)qkv, (float*)kcache,
                         (float*)vcache, (float*)out, d_pos, R, d, Lmax);
    else
      hipLaunchKernelGGL((dec_self_attn_row_kernel<bf16_t>), g4, dim3(256), 0, (hipStream_t)s, (const bf16_t*)qkv, (bf16_t*)kcache,
                         (bf16_t*)vcache, (bf16_t*)out, d_pos, R, d, Lmax);
    OMP_CHECK_LAUNCH("omp_dec_self_attn_step(rows)");
    return OMP_OK;
  }
  dim3 grid(R, (nH + 3) / 4);
  if (dtype == OMP_F32)
    hipLaunchKernelGGL((dec_self_attn_kernel<float>), grid, dim3(256), 0, (hipStream_t)s, (const float*)qkv,
                       (float*)kcache, (float*)vcache, (float*)out, d_pos, R, nH, d, Lmax);
  else if (dtype == OMP_BF16)
    hipLaunchKernelGGL((dec_self_attn_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)s, (const bf16_t*)qkv,
                       (bf16_t*)kcache, (bf16_t*)vcache, (bf16_t*)out, d_pos, R, nH, d, Lmax);
  else { omp_set_error("omp_dec_self_attn_step: bad dtype"); return OMP_ERR_INVALID; }
  OMP_CHECK_LAUNCH("omp_dec_self_attn_step");
  return OMP_OK;
}

extern "C" int omp_dec_cross_attn_step(const void* q, int64_t ldq, const void* K, const void* Vt, int64_t img_stride,
                                       int Mpad, const uint8_t* key_mask, const int32_t* groups, int n_groups,
                                       int q_tiles, int R, float* partial, void* out, int64_t ldo, int dtype, int M,
                                       int nH, int n_split, omp_stream_t s) {
  OMP_CHECK_ARG(q && K && Vt && groups && out, "omp_dec_cross_attn_step: null pointer");
  OMP_CHECK_ARG(R > 0, "omp_dec_cross_attn_step: bad R");
  OMP_CHECK_ARG(dtype == OMP_F32 || dtype == OMP_BF16 || dtype == OMP_BF16X2, "omp_dec_cross_attn_step: bad dtype");
  OMP_CHECK_ARG(n_groups > 0 && n_split > 0 && M > 0, "omp_dec_cross_attn_step: bad sizes");
  OMP_CHECK_ARG(dtype != OMP_BF16X2 || (ldq % 4 == 0 && ldo % 4 == 0 && ((uintptr_t)q % 16) == 0 && ((uintptr_t)out % 16) == 0),
                "omp_dec_cross_attn_step: split-plane slabs take 16-byte aligned fp32 q / out rows");
  CrossP cp;
  cp.q = q; cp.ldq = ldq; cp.K = K; cp.V = Vt; cp.img_stride = img_stride; cp.kmask = key_mask; cp.groups = groups;
  cp.out = out; cp.ldo = ldo; cp.partial = partial; cp.M = M; cp.Mpad = Mpad; cp.nH = nH; cp.R = R; cp.kpw = 0; cp.lo_off = 0;
  return launch_cross(cp, n_groups, dtype, n_split, q_tiles, (hipStream_t)s);
}

namespace {
// OMP355_SAMPLE_BLOCK_MAX_ROWS (A/B knob): parsed strictly -- a value that is not a non-negative integer is an ERROR of every decoder run
// (check_plan), not a silently different sampling kernel (strtol turned garbage into 0)
struct SampleEnv { int rows; bool ok; };
SampleEnv strict_env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  if (e == nullptr || *e == 0) return SampleEnv{dflt, true};
  char* end = nullptr;
  const long x = strtol(e, &end, 10);
  if (end == e || *end != 0 || x < 0) return SampleEnv{dflt, false};
  return SampleEnv{(int)(x > (1 << 30) ? (1 << 30) : x), true};
}
const SampleEnv& sample_env() {
  static const SampleEnv v = strict_env_int("OMP355_SAMPLE_BLOCK_MAX_ROWS", 1024);
  return v;
}
const SampleEnv& fused_sa_env() {
  static const SampleEnv v = strict_env_int("OMP355_FUSED_SA_MAX_ROWS", 63);
  return v;
}
int sample_block_max_rows() {   // up to this many rows a WORKGROUP per row samples (few-row phases: latency), beyond a wave per row with the row in registers
  return sample_env().rows;
}
}  // namespace

extern "C" int omp_head_softmax_mask_argmax(const float* logits, int ld, int R, const omp_sample_cfg* cfg,
                                            int32_t* seq, float* probs, int seq_ld, int32_t* finished,
                                            int32_t* lengths, int32_t* d_pos, int advance, omp_stream_t s) {
  OMP_CHECK_ARG(logits && cfg && seq && probs && d_pos, "omp_head_softmax_mask_argmax: null pointer");
  OMP_CHECK_ARG(R > 0 && cfg->vocab > 0 && cfg->vocab <= ld, "omp_head_softmax_mask_argmax: bad shape");
  if (R > sample_block_max_rows() && ld % 4 == 0 && ((uintptr_t)logits % 16) == 0 && cfg->vocab <= 2048)   // many rows: the row in registers (dec_sample_rows_kernel)
    hipLaunchKernelGGL(dec_sample_rows_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)s, logits, ld, R, *cfg, seq, probs, seq_ld, finished, lengths,
                       d_pos, (int32_t*)nullptr);
  else
    hipLaunchKernelGGL(dec_sample_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)s, logits, ld, R, *cfg,
                       seq, probs, seq_ld, finished, lengths, d_pos);
  OMP_CHECK_LAUNCH("omp_head_softmax_mask_argmax");
  if (advance) {
    hipLaunchKernelGGL(advance_pos_kernel, dim3(1), dim3(1), 0, (hipStream_t)s, d_pos);
    OMP_CHECK_LAUNCH("omp_head_softmax_mask_argmax(advance)");
  }
  return OMP_OK;
}

// ---------------------------------------------------------------------------------------------
// one full decoder step, launched from the host (eager or under stream capture)
// ---------------------------------------------------------------------------------------------
namespace {

#define RUN(expr)                 \
  do {                            \
    int rc__ = (expr);            \
    if (rc__ != OMP_OK) return rc__; \
  } while (0)

int gemm(const omp_decoder_plan* P, const void* A, int64_t lda, const void* W, int K, int N, const float* bias,
         const int32_t* bias_row, int64_t bias_stride, const void* res, void* C, int out_dtype, int act,
         hipStream_t st, const float* ln_g = nullptr, const float* ln_b = nullptr) {
  omp_gemm_args a{};
  a.A = A; a.lda = lda; a.W = W; a.ldw = K; a.bias = bias; a.bias_row = bias_row; a.bias_row_stride = bias_stride;
  a.residual = res; a.ldr = N; a.C = C; a.ldc = N; a.M = P->R; a.N = N; a.K = K;
  a.dtype = P->dtype; a.out_dtype = out_dtype; a.act = act; a.trans_out = 0; a.trans_rows = 0; a.trans_ld = 0;
  a.ln_gamma = ln_g; a.ln_beta = ln_b; a.ln_eps = P->eps; a.small_m_splitk = 1;
  return omp_gemm_bias_act(&a, st);
}

// bf16x3 product of a decoder step (plan->gemm_x3): A = split-bf16 pair rows [R, 2 K0], W = [N, 3 K0] image of the fp32 weight,
// fp32 or split-pair destination (out_dtype OMP_F32 / OMP_BF16X2), fp32 residual
int gemm_x3(const omp_decoder_plan* P, const void* A, const void* W, int K0, int N, const float* bias, const int32_t* bias_row,
            int64_t bias_stride, const void* res, void* C, int out_dtype, int act, hipStream_t st) {
  omp_gemm_args a{};
  a.A = A; a.lda = 2 * (int64_t)K0; a.W = W; a.ldw = 3 * (int64_t)K0; a.bias = bias; a.bias_row = bias_row; a.bias_row_stride = bias_stride;
  a.residual = res; a.ldr = N; a.C = C; a.ldc = out_dtype == OMP_BF16X2 ? 2 * (int64_t)N : N; a.M = P->R; a.N = N; a.K = 3 * K0;
  a.dtype = OMP_BF16; a.out_dtype = out_dtype; a.act = act; a.a_wrap = 2 * K0;
  return omp_gemm_bias_act(&a, st);
}

// one pre-norm layer stack + head of the bf16x3 engine's many-row phases: fp32 residual stream / caches / slabs / attention
// kernels, every product on the bf16 matrix cores as three bf16 products of split operands
int step_launch_x3(const omp_decoder_plan* P, bool do_head, hipStream_t st) {
  const int d = P->d_model, R = P->R;
  const int S2 = OMP_BF16X2, F = OMP_F32;
  CrossP cp;
  cp.q = P->q; cp.ldq = d; cp.img_stride = P->kv_img_stride; cp.Mpad = P->Mpad;
  cp.kmask = P->key_mask; cp.groups = P->tiles; cp.out = P->att; cp.ldo = d;
  cp.partial = P->partial; cp.M = P->M; cp.nH = P->n_heads; cp.R = R; cp.kpw = 0; cp.lo_off = 0;
  RUN(omp_dec_embed_ln(P->seq, P->seq_ld, P->d_pos, P->word_emb, P->pos_tab, P->emb_g, P->emb_b, P->x, nullptr, F, R, d, P->eps, st));
  void* ys = P->y;       // [R, 2d] bf16 split pairs (the bytes of the fp32 [R, d] buffer)
  void* as = P->ffh;     // attention outputs as split pairs (the FFN hidden buffer is free until ff1)
  if (P->kv_split) {     // split-plane slabs: the cross-attention kernels write the pair rows of the out-projection themselves
    cp.out = as; cp.ldo = 2 * (int64_t)d; cp.lo_off = d;
  }
  for (int li = 0; li < P->n_layers; ++li) {
    const omp_dec_layer& L = P->layers[li];
    cp.K = L.crossK; cp.V = L.crossVt;
    RUN(omp_layernorm(P->x, F, L.n1_g, L.n1_b, ys, S2, nullptr, R, d, P->eps, st));
    RUN(gemm_x3(P, ys, L.sa_in_w, d, 3 * d, L.sa_bias_tab, P->d_pos, 3 * d, nullptr, P->qkv, F, OMP_ACT_NONE, st));
    RUN(omp_dec_self_attn_step(P->qkv, L.kcache, L.vcache, P->att, P->d_pos, F, R, P->n_heads, d, P->Lmax, st));
    RUN(omp_split_bf16(reinterpret_cast<const float*>(P->att), d, as, 2 * d, R, d, 0, st));
    RUN(gemm_x3(P, as, L.sa_out_w, d, d, L.sa_out_b, nullptr, 0, P->x, P->x, F, OMP_ACT_NONE, st));
    RUN(omp_layernorm(P->x, F, L.n2_g, L.n2_b, ys, S2, nullptr, R, d, P->eps, st));
    RUN(gemm_x3(P, ys, L.ca_q_w, d, d, L.ca_qbias_tab, P->d_pos, d, nullptr, P->q, F, OMP_ACT_NONE, st));
    if (P->kv_split) {
      RUN(launch_cross(cp, P->n_tiles, OMP_BF16X2, P->n_split, P->q_tiles, st));
    } else {
      RUN(launch_cross(cp, P->n_tiles, F, P->n_split, P->q_tiles, st));
      RUN(omp_split_bf16(reinterpret_cast<const float*>(P->att), d, as, 2 * d, R, d, 0, st));
    }
    RUN(gemm_x3(P, as, L.ca_out_w, d, d, L.ca_out_b, nullptr, 0, P->x, P->x, F, OMP_ACT_NONE, st));
    RUN(omp_layernorm(P->x, F, L.n3_g, L.n3_b, ys, S2, nullptr, R, d, P->eps, st));
    RUN(gemm_x3(P, ys, L.ff1_w, d, P->d_ff, L.ff1_b, nullptr, 0, nullptr, P->ffh, S2, OMP_ACT_RELU, st));
    RUN(gemm_x3(P, P->ffh, L.ff2_w, P->d_ff, d, L.ff2_b, nullptr, 0, P->x, P->x, F, OMP_ACT_NONE, st));
  }
  if (do_head) {
    RUN(omp_layernorm(P->x, F, P->fn_g, P->fn_b, ys, S2, nullptr, R, d, P->eps, st));
    RUN(gemm_x3(P, ys, P->h0_w, d, d, P->h0_b, nullptr, 0, nullptr, P->hh0, S2, OMP_ACT_RELU, st));
    RUN(gemm_x3(P, P->hh0, P->h1_w, d, d, P->h1_b, nullptr, 0, nullptr, P->hh1, S2, OMP_ACT_RELU, st));
    RUN(gemm_x3(P, P->hh1, P->h2_w, d, P->vocab, P->h2_b, nullptr, 0, nullptr, P->logits, F, OMP_ACT_NONE, st));
  }
  return OMP_OK;
}

// y = LN(x) @ W^T (+bias...): fused LayerNorm prologue when the phase has <= 64 rows (split-K small-M
// kernel), otherwise a LayerNorm launch followed by the tiled GEMM.
int ln_gemm(const omp_decoder_plan* P, const float* g, const float* b, const void* W, int N, const float* bias,
            const int32_t* bias_row, int64_t bias_stride, void* C, int out_dtype, int act, hipStream_t st) {
  const int d = P->d_model;
  if (P->R <= 64)
    return gemm(P, P->x, d, W, d, N, bias, bias_row, bias_stride, nullptr, C, out_dtype, act, st, g, b);
  int rc = omp_layernorm(P->x, OMP_F32, g, b, P->y, P->dtype, nullptr, P->R, d, P->eps, st);
  if (rc != OMP_OK) return rc;
  return gemm(P, P->y, d, W, d, N, bias, bias_row, bias_stride, nullptr, C, out_dtype, act, st);
}

// Few-row phases (the point decoder of a small engine call) take the fused kernel for the self-attention half of a layer,
// with the embedding folded into layer 0 and the position advance into the sampling kernel: 44 -> 35 launches per step.
// (The query projection inside the cross-attention kernel was built and measured too: every (image, head, key split)
// workgroup repeating LayerNorm2 + 64 x 512 weights costs 26.4 us against 16.7 + 8.5 us for the two launches, also with the
// projection operands requested ahead of the K / V^T prefetch -- profiles/r03d_prof8_pt_step_timeline.txt.  Removed.)
int fused_sa_max_rows() {   // beyond: the workgroups' repeated weight streams (192 KB per 4 rows) cost more than the launches
  return fused_sa_env().rows;
}
bool fused_step_ok(const omp_decoder_plan* P) {
  return omp_cur().dec_fused != 1 && P->pre_norm && P->dtype == OMP_BF16 && P->d_model == 512 && P->n_heads == 8 && P->R <= fused_sa_max_rows();
}

int launch_fused_self_attn(const omp_decoder_plan* P, const omp_dec_layer& L, bool embed, hipStream_t st) {
  FusedSaP fp;
  fp.x = P->x; fp.ln_g = L.n1_g; fp.ln_b = L.n1_b; fp.eps = P->eps;
  fp.W = reinterpret_cast<const bf16_t*>(L.sa_in_w); fp.bias_tab = L.sa_bias_tab;
  fp.kc = reinterpret_cast<bf16_t*>(L.kcache); fp.vc = reinterpret_cast<bf16_t*>(L.vcache);
  fp.out = reinterpret_cast<bf16_t*>(P->att); fp.d_pos = P->d_pos; fp.R = P->R; fp.Lmax = P->Lmax;
  fp.seq = P->seq; fp.seq_ld = P->seq_ld; fp.word = P->word_emb; fp.pos_tab = P->pos_tab; fp.emb_g = P->emb_g; fp.emb_b = P->emb_b;
  fp.x_out = P->x;
  const dim3 grid((P->R + 3) / 4, P->n_heads);
  if (embed) hipLaunchKernelGGL((dec_fused_self_attn_kernel<true, 4>), grid, dim3(256), 0, st, fp);
  else hipLaunchKernelGGL((dec_fused_self_attn_kernel<false, 4>), grid, dim3(256), 0, st, fp);
  OMP_CHECK_LAUNCH("omp_decoder_run(fused self-attention)");
  return OMP_OK;
}

// fp32 self-attention step of the many-row kernel writing SPLIT PAIRS (the parity engine's chains): omp_dec_self_attn_step + omp_split_bf16 in one launch
int self_attn_rows_split(const omp_decoder_plan* P, const omp_dec_layer& L, void* out_split, hipStream_t st) {
  hipLaunchKernelGGL((dec_self_attn_row_kernel<float>), dim3((P->R + 3) / 4), dim3(256), 0, st, reinterpret_cast<const float*>(P->qkv), reinterpret_cast<float*>(L.kcache),
                     reinterpret_cast<float*>(L.vcache), reinterpret_cast<float*>(P->att), P->d_pos, P->R, P->d_model, P->Lmax, reinterpret_cast<bf16_t*>(out_split));
  OMP_CHECK_LAUNCH("omp_decoder_run(self-attention, split pairs)");
  return OMP_OK;
}

// Many-row phases (plan->rows_fused): the Linear layers as row-owner chains (bf16 engine: csrc/dec_rows.hip; parity engine, plan->gemm_x3:
// csrc/dec_rows_x3.hip -- three bf16 matrix-core products per Linear over split operands, fp32 self-attention, split-plane cross-attention) --
// embedding | q k v, then per layer self-attention, out-projection .. cross-attention query, cross-attention, out-projection .. FFN .. next
// layer's q k v (the last layer: .. prediction head): 2 + 4 per layer launches instead of 11 per layer + 5.  Prefill positions run the head too
// (its logits are not sampled): one kernel variant fewer.  The step is cut at its cross-attention kernels (RowsStep::embed / pre_cross / cross /
// post_cross) so that omp_decoder_run_pair can interleave TWO decoders' steps around them.
struct RowsStep {
  const omp_decoder_plan* P;
  omp_dec_rows_args a{};
  CrossP cp;
  bool x3;

  explicit RowsStep(const omp_decoder_plan* P_) : P(P_) {
    const int d = P->d_model, R = P->R;
    x3 = P->gemm_x3 != 0;
    a.x3 = x3 ? 1 : 0;
    a.R = R; a.eps = P->eps; a.d_pos = P->d_pos; a.x = P->x;
    a.att = x3 ? P->ffh : P->att;   // x3: attention outputs as split pairs [R, 2 d] bf16 (the FFN hidden buffer is unused on this path)
    a.seq = P->seq; a.seq_ld = P->seq_ld; a.word_emb = P->word_emb; a.pos_tab = P->pos_tab; a.emb_g = P->emb_g; a.emb_b = P->emb_b;
    a.qkv = P->qkv; a.q = P->q; a.logits = P->logits; a.vocab = P->vocab; a.h0_b = P->h0_b; a.h1_b = P->h1_b; a.h2_b = P->h2_b;
    a.xcd_mask = P->rows_xcd_mask;
    cp.q = P->q; cp.ldq = d; cp.img_stride = P->kv_img_stride; cp.Mpad = P->Mpad;
    cp.kmask = P->key_mask; cp.groups = P->tiles; cp.partial = P->partial; cp.M = P->M; cp.nH = P->n_heads; cp.R = R; cp.kpw = 0;
    if (x3 && P->kv_split) { cp.out = P->ffh; cp.ldo = 2 * (int64_t)d; cp.lo_off = d; }   // the split-plane kernels write the chains' pair rows themselves
    else { cp.out = P->att; cp.ldo = d; cp.lo_off = 0; }
  }
  // layer 0's q | k | v behind the embedding
  int embed(hipStream_t st) {
    a.prologue = 1; a.tail = 0; a.wstream = P->rows_embed; a.wave_stride = P->rows_embed_stride;   // the packer's own count: omp_dec_rows_* demand equality
    a.lnt_g = P->layers[0].n1_g; a.lnt_b = P->layers[0].n1_b; a.bias_tab = P->layers[0].sa_bias_tab;
    return omp_dec_rows_ffn(&a, st);
  }
  // self-attention, then out-projection + residual, norm2, cross-attention query (the mid chain)
  int pre_cross(int li, hipStream_t st) {
    const omp_dec_layer& L = P->layers[li];
    const int d = P->d_model, R = P->R;
    if (x3 && P->n_heads == 8 && omp_cur().self_attn_impl != 1) {
      RUN(self_attn_rows_split(P, L, P->ffh, st));   // one wave per row whatever the row count (the chains' tests run 1 .. 64 rows too)
    } else {
      RUN(omp_dec_self_attn_step(P->qkv, L.kcache, L.vcache, P->att, P->d_pos, x3 ? OMP_F32 : OMP_BF16, R, P->n_heads, d, P->Lmax, st));
      if (x3) RUN(omp_split_bf16(reinterpret_cast<const float*>(P->att), d, P->ffh, 2 * d, R, d, 0, st));
    }
    a.wstream = L.rows_mid; a.wave_stride = L.rows_mid_stride;
    a.out_b = L.sa_out_b; a.ln_g = L.n2_g; a.ln_b = L.n2_b; a.qbias_tab = L.ca_qbias_tab;
    return omp_dec_rows_mid(&a, st);
  }
  int cross(int li, hipStream_t st) {
    const omp_dec_layer& L = P->layers[li];
    const int d = P->d_model, R = P->R;
    cp.K = L.crossK; cp.V = L.crossVt;
    if (!x3) return launch_cross(cp, P->n_tiles, OMP_BF16, P->n_split, P->q_tiles, st);
    if (P->kv_split) return launch_cross(cp, P->n_tiles, OMP_BF16X2, P->n_split, P->q_tiles, st);
    RUN(launch_cross(cp, P->n_tiles, OMP_F32, P->n_split, P->q_tiles, st));
    return omp_split_bf16(reinterpret_cast<const float*>(P->att), d, P->ffh, 2 * d, R, d, 0, st);
  }
  // out-projection + residual, norm3, FFN, and the next layer's norm1 + q k v (last layer: final norm + prediction head)
  int post_cross(int li, hipStream_t st) {
    const omp_dec_layer& L = P->layers[li];
    a.prologue = 0; a.wstream = L.rows_ffn; a.wave_stride = L.rows_ffn_stride;
    a.out_b = L.ca_out_b; a.ln_g = L.n3_g; a.ln_b = L.n3_b; a.ff1_b = L.ff1_b; a.ff2_b = L.ff2_b;
    if (li + 1 < P->n_layers) {
      a.tail = 0;
      a.lnt_g = P->layers[li + 1].n1_g; a.lnt_b = P->layers[li + 1].n1_b; a.bias_tab = P->layers[li + 1].sa_bias_tab;
    } else {
      a.tail = 1;
      a.lnt_g = P->fn_g; a.lnt_b = P->fn_b;
    }
    return omp_dec_rows_ffn(&a, st);
  }
};

int step_launch_rows(const omp_decoder_plan* P, hipStream_t st) {
  RowsStep rs(P);
  RUN(rs.embed(st));
  for (int li = 0; li < P->n_layers; ++li) {
    RUN(rs.pre_cross(li, st));
    RUN(rs.cross(li, st));
    RUN(rs.post_cross(li, st));
  }
  return OMP_OK;
}

int check_rows_plan(const omp_decoder_plan* P) {
  const int d = P->d_model;
  if (P->gemm_x3)
    OMP_CHECK_ARG(P->dtype == OMP_F32 && P->pre_norm && d % 64 == 0 && P->d_ff % 64 == 0, "omp_decoder_run: gemm_x3 plans are fp32, pre-norm, widths multiples of 64");
  else
    OMP_CHECK_ARG(P->dtype == OMP_BF16 && P->pre_norm && !P->kv_split, "omp_decoder_run: rows_fused plans without gemm_x3 are bf16, pre-norm, plain slabs");
  OMP_CHECK_ARG(d == 512 && P->d_ff == 2048 && P->n_heads == 8 && P->vocab % 4 == 0 && P->rows_embed != nullptr,
                "omp_decoder_run: rows_fused plans are d_model 512 / d_ff 2048 / 8 heads, vocab %% 4 == 0, with packed streams bound");
  for (int li = 0; li < P->n_layers; ++li)
    OMP_CHECK_ARG(P->layers[li].rows_mid && P->layers[li].rows_ffn, "omp_decoder_run: rows_fused plan without the packed streams of layer %d", li);
  return OMP_OK;
}

int step_launch(const omp_decoder_plan* P, bool do_head, hipStream_t st) {
  const int d = P->d_model, R = P->R, T = P->dtype;
  if (P->rows_fused) {
    RUN(check_rows_plan(P));
    return step_launch_rows(P, st);
  }
  if (P->gemm_x3) {
    OMP_CHECK_ARG(T == OMP_F32 && P->pre_norm && R > 64 && d % 64 == 0 && P->d_ff % 64 == 0,
                  "omp_decoder_run: gemm_x3 plans are fp32, pre-norm, more than 64 rows (any number on the row-owner chains), widths multiples of 64");
    return step_launch_x3(P, do_head, st);
  }
  const bool fused = fused_step_ok(P);
  // embedding: pre-norm needs only the fp32 stream; post-norm also the T copy (fused: layer 0's first kernel embeds)
  if (!fused)
    RUN(omp_dec_embed_ln(P->seq, P->seq_ld, P->d_pos, P->word_emb, P->pos_tab, P->emb_g, P->emb_b, P->x,
                         P->pre_norm ? nullptr : P->y, T, R, d, P->eps, st));
  CrossP cp;
  cp.q = P->q; cp.ldq = d; cp.img_stride = P->kv_img_stride; cp.Mpad = P->Mpad;
  cp.kmask = P->key_mask; cp.groups = P->tiles; cp.out = P->att; cp.ldo = d;
  cp.partial = P->partial; cp.M = P->M; cp.nH = P->n_heads; cp.R = R; cp.kpw = 0; cp.lo_off = 0;
  const int TC = P->kv_split ? OMP_BF16X2 : T;   // slab format the cross-attention kernels read (split planes: fp32 q / out)
  // Phases between the fused few-row kernels (<= 63 rows) and the full chains (rows_fused): when the caller bound layers[].rows_mid, the three
  // launches between self- and cross-attention -- 1 MB of weights, 23 us at 160 rows, bound by two launch boundaries -- are the mid chain on
  // 16-row workgroups (the point decoder of a 160-image call: 10 workgroups stream the megabyte in 9 us).  The FFN half stays on launches: its
  // 6.3 MB per workgroup would stream longer than the launches take.
  bool mid_chain = !fused && omp_cur().dec_fused != 1 && T == OMP_BF16 && P->pre_norm && d == 512 && P->n_heads == 8 && !P->kv_split && R >= 16;
  for (int li = 0; li < P->n_layers && mid_chain; ++li) mid_chain = P->layers[li].rows_mid != nullptr;
  omp_dec_rows_args ma{};
  ma.R = R; ma.eps = P->eps; ma.d_pos = P->d_pos; ma.x = P->x; ma.att = P->att; ma.q = P->q;
  for (int li = 0; li < P->n_layers; ++li) {
    const omp_dec_layer& L = P->layers[li];
    cp.K = L.crossK; cp.V = L.crossVt;
    if (P->pre_norm) {
      if (fused) {
        RUN(launch_fused_self_attn(P, L, li == 0, st));
      } else {
        RUN(ln_gemm(P, L.n1_g, L.n1_b, L.sa_in_w, 3 * d, L.sa_bias_tab, P->d_pos, 3 * d, P->qkv, T, OMP_ACT_NONE, st));
        RUN(omp_dec_self_attn_step(P->qkv, L.kcache, L.vcache, P->att, P->d_pos, T, R, P->n_heads, d, P->Lmax, st));
      }
      if (mid_chain) {   // out-projection + residual, norm2, cross-attention query: ONE launch (csrc/dec_rows.hip) instead of three
        ma.wstream = L.rows_mid; ma.wave_stride = L.rows_mid_stride; ma.out_b = L.sa_out_b; ma.ln_g = L.n2_g; ma.ln_b = L.n2_b; ma.qbias_tab = L.ca_qbias_tab;
        RUN(omp_dec_rows_mid(&ma, st));
      } else {
        RUN(gemm(P, P->att, d, L.sa_out_w, d, d, L.sa_out_b, nullptr, 0, P->x, P->x, OMP_F32, OMP_ACT_NONE, st));
        RUN(ln_gemm(P, L.n2_g, L.n2_b, L.ca_q_w, d, L.ca_qbias_tab, P->d_pos, d, P->q, T, OMP_ACT_NONE, st));
      }
      RUN(launch_cross(cp, P->n_tiles, TC, P->n_split, P->q_tiles, st));
      RUN(gemm(P, P->att, d, L.ca_out_w, d, d, L.ca_out_b, nullptr, 0, P->x, P->x, OMP_F32, OMP_ACT_NONE, st));
      RUN(ln_gemm(P, L.n3_g, L.n3_b, L.ff1_w, P->d_ff, L.ff1_b, nullptr, 0, P->ffh, T, OMP_ACT_RELU, st));
      RUN(gemm(P, P->ffh, P->d_ff, L.ff2_w, P->d_ff, d, L.ff2_b, nullptr, 0, P->x, P->x, OMP_F32, OMP_ACT_NONE, st));
    } else {
      RUN(gemm(P, P->y, d, L.sa_in_w, d, 3 * d, L.sa_bias_tab, P->d_pos, 3 * d, nullptr, P->qkv, T, OMP_ACT_NONE, st));
      RUN(omp_dec_self_attn_step(P->qkv, L.kcache, L.vcache, P->att, P->d_pos, T, R, P->n_heads, d, P->Lmax, st));
      RUN(gemm(P, P->att, d, L.sa_out_w, d, d, L.sa_out_b, nullptr, 0, P->x, P->x2, OMP_F32, OMP_ACT_NONE, st));
      RUN(omp_layernorm(P->x2, OMP_F32, L.n1_g, L.n1_b, P->y, T, P->x, R, d, P->eps, st));
      RUN(gemm(P, P->y, d, L.ca_q_w, d, d, L.ca_qbias_tab, P->d_pos, d, nullptr, P->q, T, OMP_ACT_NONE, st));
      RUN(launch_cross(cp, P->n_tiles, TC, P->n_split, P->q_tiles, st));
      RUN(gemm(P, P->att, d, L.ca_out_w, d, d, L.ca_out_b, nullptr, 0, P->x, P->x2, OMP_F32, OMP_ACT_NONE, st));
      RUN(omp_layernorm(P->x2, OMP_F32, L.n2_g, L.n2_b, P->y, T, P->x, R, d, P->eps, st));
      RUN(gemm(P, P->y, d, L.ff1_w, d, P->d_ff, L.ff1_b, nullptr, 0, nullptr, P->ffh, T, OMP_ACT_RELU, st));
      RUN(gemm(P, P->ffh, P->d_ff, L.ff2_w, P->d_ff, d, L.ff2_b, nullptr, 0, P->x, P->x2, OMP_F32, OMP_ACT_NONE, st));
      RUN(omp_layernorm(P->x2, OMP_F32, L.n3_g, L.n3_b, P->y, T, P->x, R, d, P->eps, st));
    }
  }
  if (do_head) {
    RUN(ln_gemm(P, P->fn_g, P->fn_b, P->h0_w, d, P->h0_b, nullptr, 0, P->hh0, T, OMP_ACT_RELU, st));
    RUN(gemm(P, P->hh0, d, P->h1_w, d, d, P->h1_b, nullptr, 0, nullptr, P->hh1, T, OMP_ACT_RELU, st));
    RUN(gemm(P, P->hh1, d, P->h2_w, d, P->vocab, P->h2_b, nullptr, 0, nullptr, P->logits, OMP_F32, OMP_ACT_NONE, st));
  }
  return OMP_OK;
}

// sampling steps per multi-step graph; OMP355_GRAPH_RUN=1 switches the runs off (A/B): parsed strictly like the other knobs of this file
int graph_run_steps() {
  static const SampleEnv v = strict_env_int("OMP355_GRAPH_RUN", OMP_GRAPH_RUN);
  return v.ok && v.rows >= 1 && v.rows <= OMP_GRAPH_RUN ? (v.rows == OMP_GRAPH_RUN ? OMP_GRAPH_RUN : 1) : OMP_GRAPH_RUN;
}

int check_plan(const omp_decoder_plan* P) {
  OMP_CHECK_ARG(P != nullptr, "omp_decoder_run: null plan");
  OMP_CHECK_ARG(sample_env().ok && fused_sa_env().ok, "omp_decoder_run: OMP355_SAMPLE_BLOCK_MAX_ROWS / OMP355_FUSED_SA_MAX_ROWS must be non-negative integers");
  OMP_CHECK_ARG(P->dtype == OMP_F32 || P->dtype == OMP_BF16, "omp_decoder_run: bad dtype");
  OMP_CHECK_ARG(P->n_layers > 0 && P->n_layers <= OMP_MAX_DEC_LAYERS, "omp_decoder_run: bad n_layers %d", P->n_layers);
  OMP_CHECK_ARG(P->d_model == P->n_heads * DH, "omp_decoder_run: head_dim must be 64");
  OMP_CHECK_ARG(P->R > 0 && P->Lmax > 0 && P->M > 0 && P->n_tiles > 0 && P->n_split > 0 && P->q_tiles > 0 && P->Mpad >= P->M,
                "omp_decoder_run: bad sizes");
  OMP_CHECK_ARG(P->seq && P->d_pos && P->probs && P->x && P->y && P->qkv && P->att && P->q && P->ffh && P->hh0 &&
                    P->hh1 && P->logits && P->tiles,
                "omp_decoder_run: null buffer in plan");
  OMP_CHECK_ARG(P->pre_norm || P->x2, "omp_decoder_run: post-norm needs x2");
  OMP_CHECK_ARG(!P->kv_split || (P->dtype == OMP_F32 && P->Mpad % 32 == 0), "omp_decoder_run: split-plane K / V^T slabs go with fp32 plans and 32-key blocks");
  return OMP_OK;
}

int sample_and_advance(const omp_decoder_plan* P, hipStream_t st) {
  if (P->R > sample_block_max_rows() && P->vocab % 4 == 0 && P->vocab <= 2048 && omp_cur().dec_fused != 1) {
    // many rows: a wave per row, the row in registers (dec_sample_rows_kernel); the last workgroup publishes the next position
    hipLaunchKernelGGL(dec_sample_rows_kernel, dim3((P->R + 3) / 4), dim3(256), 0, st, P->logits, P->vocab, P->R, P->sample, P->seq, P->probs,
                       P->seq_ld, P->finished, P->lengths, P->d_pos, P->d_pos + 1);
    OMP_CHECK_LAUNCH("omp_decoder_run(sample)");
    return OMP_OK;
  }
  if (P->R <= 65536 && omp_cur().dec_fused != 1) {
    // a workgroup per row; the last one to finish publishes the next position (P->d_pos[1] is the ticket word)
    hipLaunchKernelGGL(dec_sample_block_kernel, dim3(P->R), dim3(256), 0, st, P->logits, P->vocab, P->R, P->sample, P->seq, P->probs,
                       P->seq_ld, P->finished, P->lengths, P->d_pos, P->d_pos + 1);
    OMP_CHECK_LAUNCH("omp_decoder_run(sample)");
    return OMP_OK;
  }
  return omp_head_softmax_mask_argmax(P->logits, P->vocab, P->R, &P->sample, P->seq, P->probs, P->seq_ld,
                                      P->finished, P->lengths, P->d_pos, 1, st);
}

}  // namespace

extern "C" int omp_debug_dec_fused(int mode) {
  omp_cur().dec_fused = mode == 1 ? 1 : 0;
  return OMP_OK;
}

extern "C" int omp_debug_self_attn_impl(int which) {
  omp_cur().self_attn_impl = (which == 1 || which == 2) ? which : 0;
  return OMP_OK;
}

extern "C" int omp_debug_cross_nt(int on) {
  omp_cur().cross_nt = on == 2 ? 2 : (on ? 1 : 0);   // 1 = non-temporal at every size (default), 2 = only from 32 groups per launch (round 2), 0 = never
  return OMP_OK;
}

extern "C" int omp_debug_cross_q4(int on) {
  omp_cur().cross_q4 = (on == 0 || on == 2 || on == 4 || on == 5 || on == 6) ? on : 1;
  return OMP_OK;
}

extern "C" int omp_capture_gate_enter(void) { g_capture_gate.lock(); return OMP_OK; }
extern "C" int omp_capture_gate_leave(void) { g_capture_gate.unlock(); return OMP_OK; }

extern "C" int omp_decoder_graph_reset(int slot) {
  if (slot >= 0 && slot < OMP_MAX_GRAPH_SLOTS) {
    OmpGraphSlot& gs = omp_cur().slots[slot];
    if (gs.exec) (void)hipGraphExecDestroy(gs.exec);
    if (gs.graph) (void)hipGraphDestroy(gs.graph);
    if (gs.exec_n) (void)hipGraphExecDestroy(gs.exec_n);
    if (gs.graph_n) (void)hipGraphDestroy(gs.graph_n);
    gs = OmpGraphSlot();
  }
  return OMP_OK;
}

extern "C" int omp_decoder_run(const omp_decoder_plan* P, int first_pos, int n_steps, int graph_slot,
                               omp_stream_t s) {
  RUN(check_plan(P));
  OMP_CHECK_ARG(first_pos >= 0 && first_pos + n_steps <= P->Lmax, "omp_decoder_run: positions %d..%d exceed Lmax %d",
                first_pos, first_pos + n_steps, P->Lmax);
  OMP_CHECK_ARG(first_pos + n_steps + 1 <= P->seq_ld, "omp_decoder_run: seq_ld %d too small", P->seq_ld);
  hipStream_t st = (hipStream_t)s;
  for (int i = 0; i < n_steps; ++i) {
    const int pos = first_pos + i;
    const bool do_head = pos >= P->n_prompt - 1;
    if (!do_head) {
      RUN(step_launch(P, false, st));
      hipLaunchKernelGGL(advance_pos_kernel, dim3(1), dim3(1), 0, st, P->d_pos);
      OMP_CHECK_LAUNCH("omp_decoder_run(advance)");
      continue;
    }
    if (graph_slot < 0) {
      RUN(step_launch(P, true, st));
      RUN(sample_and_advance(P, st));
      continue;
    }
    OMP_CHECK_ARG(graph_slot < OMP_MAX_GRAPH_SLOTS, "omp_decoder_run: graph slot %d out of range", graph_slot);
    OmpGraphSlot& gs = omp_cur().slots[graph_slot];
    // capture `steps` sampling steps of this plan into (graph, exec) once, then replay: a step reads its position from the device counter, so every
    // step of a phase is the same launch sequence.  Consecutive graph launches are ~8.5 us apart on the GPU (profiles/r06f_pt_step_timeline_*): runs of
    // OMP_GRAPH_RUN steps go out as ONE graph (round 6), the remainder as single-step graphs.
    auto replay = [&](hipGraph_t& graph, hipGraphExec_t& exec, int steps) -> int {
      if (exec == nullptr) {
        std::lock_guard<std::mutex> gate(g_capture_gate);   // held until the graph is instantiated: see g_capture_gate
        hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        if (e != hipSuccess) { omp_set_error("omp_decoder_run: begin capture: %s", hipGetErrorString(e)); return OMP_ERR_LAUNCH; }
        g_capturing = true;
        int rc = OMP_OK;
        for (int k = 0; k < steps && rc == OMP_OK; ++k) {
          rc = step_launch(P, true, st);
          if (rc == OMP_OK) rc = sample_and_advance(P, st);
        }
        g_capturing = false;
        e = hipStreamEndCapture(st, &graph);
        if (rc != OMP_OK) return rc;
        if (e != hipSuccess) { omp_set_error("omp_decoder_run: end capture: %s", hipGetErrorString(e)); return OMP_ERR_LAUNCH; }
        e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        if (e != hipSuccess) { omp_set_error("omp_decoder_run: instantiate: %s", hipGetErrorString(e)); return OMP_ERR_LAUNCH; }
      }
      const hipError_t e = hipGraphLaunch(exec, st);
      if (e != hipSuccess) { omp_set_error("omp_decoder_run: graph launch: %s", hipGetErrorString(e)); return OMP_ERR_LAUNCH; }
      return OMP_OK;
    };
    if (graph_run_steps() > 1 && n_steps - i >= graph_run_steps()) {   // (every later position of this call samples too: pos only grows)
      RUN(replay(gs.graph_n, gs.exec_n, graph_run_steps()));
      i += graph_run_steps() - 1;
      continue;
    }
    RUN(replay(gs.graph, gs.exec, 1));
  }
  return OMP_OK;
}

// Two decoders' many-row phases (polygon || recognition: transformer.py:252-284) as ONE interleaved schedule on three streams.
// A step of a many-row phase alternates between two kinds of launches: the row-owner chains -- matrix-core work on at most HALF the compute
// units (a 10 240-row launch is 128 workgroups, each holding its CU's LDS whole) -- and the cross-attention kernel, which streams K / V^T of
// every image at the HBM rate the CUs it gets can draw (2.3 / 4.2 / 5.4 / 6.3 TB/s on 64 / 128 / 192 / 256 CUs, profiles/r02u_*).  Free-running
// on two streams the two decoders fall into lock-step: both reach their cross-attention together (each at half rate: 438-464 us instead of
// 230, profiles/r05m_q4_overlap_bf16.txt), then both run their chains together.  Here ALL cross-attention launches go to ONE stream sx in the
// order A[0], B[0], A[1], B[1] ... (serialised by the stream itself), each behind an event of its decoder's chain stream, and each chain stream
// waits for its decoder's cross-attention by an event of sx: one decoder's chains (and its self-attention) then run beside the OTHER decoder's
// cross-attention.  sa / sb should be HIGHER-priority streams than sx: a chain workgroup needs a whole CU's LDS, so it only gets a CU at the
// moment the previous cross-attention launch has drained -- the same moment the next one becomes ready -- and the queue priority decides who
// is placed first (with equal priorities the 1280 cross-attention workgroups take every CU and the chains wait: measured, no gain).
// Eager launches (a step is 19 launches of 20-280 us: the host stays far ahead); positions beyond the shorter decoder's last run alone.
extern "C" int omp_decoder_run_pair(const omp_decoder_plan* PA, const omp_decoder_plan* PB, int first_pos, int n_steps_a, int n_steps_b,
                                    omp_stream_t sa, omp_stream_t sb, omp_stream_t sx_) {
  RUN(check_plan(PA));
  RUN(check_plan(PB));
  OMP_CHECK_ARG(PA->rows_fused && PB->rows_fused && PA->n_layers == PB->n_layers, "omp_decoder_run_pair: two rows_fused plans with the same number of layers");
  RUN(check_rows_plan(PA));
  RUN(check_rows_plan(PB));
  OMP_CHECK_ARG(first_pos >= 0 && n_steps_a >= 0 && n_steps_b >= 0 && first_pos + n_steps_a <= PA->Lmax && first_pos + n_steps_b <= PB->Lmax,
                "omp_decoder_run_pair: positions exceed Lmax");
  OMP_CHECK_ARG(first_pos + n_steps_a + 1 <= PA->seq_ld && first_pos + n_steps_b + 1 <= PB->seq_ld, "omp_decoder_run_pair: seq_ld too small");
  OMP_CHECK_ARG(sa != sb && sa != sx_ && sb != sx_ && sa != nullptr && sb != nullptr && sx_ != nullptr, "omp_decoder_run_pair: three distinct non-default streams");
  hipStream_t st[2] = {(hipStream_t)sa, (hipStream_t)sb};
  hipStream_t sx = (hipStream_t)sx_;
  const omp_decoder_plan* P[2] = {PA, PB};
  const int n[2] = {n_steps_a, n_steps_b};
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr};   // [k]: decoder k's chain stream has reached its cross-attention; [2]: sx has finished a cross-attention
  for (int k = 0; k < 3; ++k)
    if (hipEventCreateWithFlags(&ev[k], hipEventDisableTiming) != hipSuccess) {
      for (int j = 0; j < k; ++j) (void)hipEventDestroy(ev[j]);
      omp_set_error("omp_decoder_run_pair: hipEventCreate failed");
      return OMP_ERR_LAUNCH;
    }
  RowsStep rs[2] = {RowsStep(PA), RowsStep(PB)};
  auto finish = [&](int k, int pos) -> int {
    if (pos >= P[k]->n_prompt - 1) return sample_and_advance(P[k], st[k]);
    hipLaunchKernelGGL(advance_pos_kernel, dim3(1), dim3(1), 0, st[k], P[k]->d_pos);
    OMP_CHECK_LAUNCH("omp_decoder_run_pair(advance)");
    return OMP_OK;
  };
  // decoder k's cross-attention of layer li on sx, between its chain stream's "ready" event and the event that releases the chain stream again
  auto cross_on_sx = [&](int k, int li) -> int {
    if (hipEventRecord(ev[k], st[k]) != hipSuccess || hipStreamWaitEvent(sx, ev[k], 0) != hipSuccess) { omp_set_error("omp_decoder_run_pair: event record / wait failed"); return OMP_ERR_LAUNCH; }
    RUN(rs[k].cross(li, sx));
    if (hipEventRecord(ev[2], sx) != hipSuccess || hipStreamWaitEvent(st[k], ev[2], 0) != hipSuccess) { omp_set_error("omp_decoder_run_pair: event record / wait failed"); return OMP_ERR_LAUNCH; }
    return OMP_OK;
  };
  auto body = [&]() -> int {
    const int steps = n[0] > n[1] ? n[0] : n[1];
    for (int i = 0; i < steps; ++i) {
      const int pos = first_pos + i;
      const bool on[2] = {i < n[0], i < n[1]};
      if (on[0] != on[1]) {   // the longer decoder alone, on its own stream
        const int k = on[0] ? 0 : 1;
        RUN(step_launch_rows(P[k], st[k]));
        RUN(finish(k, pos));
        continue;
      }
      RUN(rs[0].embed(st[0]));
      RUN(rs[1].embed(st[1]));
      for (int li = 0; li < PA->n_layers; ++li) {
        RUN(rs[0].pre_cross(li, st[0]));
        RUN(rs[1].pre_cross(li, st[1]));
        RUN(cross_on_sx(0, li));
        RUN(cross_on_sx(1, li));
        RUN(rs[0].post_cross(li, st[0]));
        RUN(rs[1].post_cross(li, st[1]));
      }
      RUN(finish(0, pos));
      RUN(finish(1, pos));
    }
    // sx has run ahead of nothing: its last cross-attention is awaited by a chain stream; the caller joins sa and sb
    return OMP_OK;
  };
  const int rc = body();
  for (int k = 0; k < 3; ++k) (void)hipEventDestroy(ev[k]);   // released once the recorded work has completed
  return rc;
}

extern "C" int omp_decoder_step_logits(const omp_decoder_plan* P, int pos, omp_stream_t s) {
  RUN(check_plan(P));
  (void)pos;
  hipStream_t st = (hipStream_t)s;
  RUN(step_launch(P, true, st));
  hipLaunchKernelGGL(advance_pos_kernel, dim3(1), dim3(1), 0, st, P->d_pos);
  OMP_CHECK_LAUNCH("omp_decoder_step_logits(advance)");
  return OMP_OK;
}
