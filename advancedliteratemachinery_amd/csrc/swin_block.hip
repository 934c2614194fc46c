// Attention half of a Swin block at C = 128 (Swin-B stage 0: 4 heads of 32) in ONE kernel:
//     x <- x + proj(window_attention(qkv(LayerNorm(x))))        (reference swin_transformer.py:196-253, first residual)
// The unfused chain moves 4.6 KB per token through HBM (LN 0.75 + qkv 1.0 + attention 1.0 + proj 1.25 ... of reads and
// writes); here a token's 512 B are read once and written once, q / k / v / P / O never leave the CU.
//
// One persistent workgroup per CU, 8 waves = 2 windows in flight x 4 heads:
//   * the qkv weights (384 x 128 bf16 = 96 KB) sit in LDS for the life of the workgroup, chunk-swizzled
//     (16-byte chunk c of row r at slot c ^ (r & 15)) so that every MFMA fragment is one conflict-free ds_read_b128;
//     biases and the norm1 vectors sit next to them (3 KB); the proj weights of the wave's 32 output channels (8 fragments,
//     L2-resident) are re-fetched per window together with the residual rows -- held across windows they cost 32 of the
//     256 registers a wave has at two waves per SIMD, and the kernel spilled
//   * per window: LayerNorm of its 49 tokens (fp32 rows fetched one window AHEAD; 8 lanes per token, DPP sums) -> bf16 tile
//     [64][128] in LDS (rows of padding tokens and rows 49..63 are zero: the reference pads AFTER norm1, so their
//     q / k / v are the bias) -> each wave reads the tile as 16 operand fragments and produces its head's
//       k^T = Wk X^T,  q^T = Wq X^T   (accumulator layout [dim 4g+r][token li]  == the operand layout of S^T = K Q^T, up to a
//                                      permutation of the 32 dims that K and Q share)
//       v   = X Wv^T                  (accumulator layout [token 4g+r][dim li]  == the A operand of O^T = V^T P with the key
//                                      order the S^T accumulators give P)
//     so the whole attention (same arithmetic as swin_attn_mfma_kernel<bf16, EXPB>: expanded bias seeds, base-2 softmax,
//     SW-MSA regions only on edge windows) runs out of registers, no LDS round trip
//   * O (bf16) goes through the same LDS tile to become the proj operand; out^T = Wp O^T + b + x is stored as 16-byte
//     pieces straight into the residual stream (in place: a window touches only its own tokens)
//   * the four phase barriers per window order LDS traffic only (lds_barrier): the rows in flight for the next window and
//     the residual / proj-weight loads are never drained at a barrier
// Measured (profiles/r03o_kbench_swin_block.txt): 32 images of 256 x 256 tokens, 2.26 ms for the four launches -> 1.11 ms;
// instruction-issue-bound (about 2 900 instructions per wave and window), HBM floor 0.36 ms.
#include <atomic>

#include "common.h"

namespace {

constexpr int WS = 7, WT = 49, BC = 128;
constexpr int W_BYTES = 384 * 256, TILE_BYTES = 64 * 256;
constexpr int VEC_FLOATS = 384 + 128 + 128 + 128;   // qkv bias | proj bias | norm1 gamma | norm1 beta, fp32 in LDS
constexpr int LDS_BYTES = W_BYTES + 2 * TILE_BYTES + VEC_FLOATS * 4;

struct SwinBlockP {
  const float* x; float* out;
  const float* ln_g; const float* ln_b; float eps;
  const bf16_t* qkv_w; const float* qkv_b;
  const float* bias_exp;                 // [4 heads][64 queries][64 keys] (omp_swin_expand_bias)
  const bf16_t* proj_w; const float* proj_b;
  int B, H, W, shift, nWy, nWx, n_win;
  unsigned long long* trace;             // development: per-workgroup phase cycle sums (TRACE instantiation)
};

__device__ __forceinline__ int sw_off(int row, int chunk) { return row * 256 + ((chunk ^ (row & 15)) << 4); }

struct Win { int b, wy, wx; };
__device__ __forceinline__ Win win_decode(const SwinBlockP& p, int widx) {
  Win w;
  w.wx = widx % p.nWx;
  const int r = widx / p.nWx;
  w.wy = r % p.nWy; w.b = r / p.nWy;
  return w;
}
// window-local token t (< 64; callers mask t >= 49) -> row of x (or -1: padding token); sy / sx = its place in the shifted grid
__device__ __forceinline__ int64_t win_token(const SwinBlockP& p, const Win& w, int t, int& sy, int& sx) {
  const int Hp = p.nWy * WS, Wp = p.nWx * WS;
  const int ty = (t * 37) >> 8, tx = t - ty * WS;   // t / 7 for t < 64
  sy = w.wy * WS + ty; sx = w.wx * WS + tx;
  int py = sy + p.shift, px = sx + p.shift;
  if (py >= Hp) py -= Hp;
  if (px >= Wp) px -= Wp;
  return (py < p.H && px < p.W) ? ((int64_t)w.b * p.H + py) * p.W + px : (int64_t)-1;
}

// sum over groups of 8 consecutive lanes on the DPP path (two quad permutes, one half-row mirror): no LDS traffic
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float sum8(float v) {
  v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);   // row_half_mirror: the other quad of the 8
  return v;
}

// workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt, i.e. it would wait at every phase boundary for
// the next window's rows (HBM latency) and for the residual / proj-weight loads issued just before it
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

__device__ __forceinline__ bf16x8 pack8(const f32x4& a, const f32x4& b) {
  bf16x8 f = {(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3], (bf16_t)b[0], (bf16_t)b[1], (bf16_t)b[2], (bf16_t)b[3]};
  return f;
}

template <bool TRACE>
__global__ __launch_bounds__(512, 1) void swin_block_kernel(SwinBlockP p) {
  typedef Mma<bf16_t> MM;
  unsigned long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  auto now = [&]() -> unsigned long long { if constexpr (TRACE) return __builtin_amdgcn_s_memtime(); else return 0ull; };
  const unsigned long long t_start = now();
  extern __shared__ __attribute__((aligned(16))) char lds[];   // Wqkv image | tile of window group 0 | tile of group 1 | vectors
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);   // wave-uniform: address terms built on it live in SGPRs
  const int grp = wave_u >> 2, head = wave_u & 3, li = lane & 15, g = lane >> 4;
  char* tile = lds + W_BYTES + grp * TILE_BYTES;
  const float* vec = reinterpret_cast<const float*>(lds + W_BYTES + 2 * TILE_BYTES);
  const float* v_qb = vec; const float* v_pb = vec + 384; const float* v_g = vec + 512; const float* v_b = vec + 640;

  for (int idx = tid; idx < 384 * 16; idx += 512) {
    const int row = idx >> 4, ch = idx & 15;
    *reinterpret_cast<bf16x8*>(lds + sw_off(row, ch)) = *reinterpret_cast<const bf16x8*>(p.qkv_w + row * BC + ch * 8);
  }
  for (int idx = tid; idx < VEC_FLOATS; idx += 512) {
    float v;
    if (idx < 384) v = p.qkv_b[idx];
    else if (idx < 512) v = p.proj_b[idx - 384];
    else if (idx < 640) v = p.ln_g[idx - 512];
    else v = p.ln_b[idx - 640];
    const_cast<float*>(vec)[idx] = v;
  }
  // fragment of rows R + li (R a multiple of 16), k-step ks: byte offset R * 256 + cx[ks] -- the row term is an immediate
  int cx[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) cx[ks] = li * 256 + (((ks * 4 + g) ^ li) << 4);
  const char* wl = lds + head * 32 * 256;   // this head's rows of the q block; k rows + 128 * 256, v rows + 256 * 256
  // LayerNorm: 8 lanes per token (lane j of the 8 holds the 16-byte pieces j, j + 8, j + 16, j + 24 of the 512-byte row, so one
  // wave load covers 8 tokens x 128 contiguous bytes); wave `head` normalises tokens it*32 + head*8 + tk, it = 0, 1
  const int tk = lane >> 3, lj = lane & 7;
  const int ln_row = head * 8 + tk;                       // + 32 it
  // bf16 piece of fp32 piece c = k*8 + lj: 16-byte chunk (c >> 1) = k*4 + (lj >> 1), half (lj & 1); chunk ^ (row & 15), k enters as ^ (k << 6)
  const int ln_off = ln_row * 256 + ((((lj >> 1)) ^ (ln_row & 15)) << 4) + (lj & 1) * 8;
  // O^T accumulator piece (dims dt*16 + 4g .. +4 of query t4*16 + li) -> tile[query][head*32 + dim]
  const int o_off0 = li * 256 + (((head * 4 + (g >> 1)) ^ li) << 4) + (g & 1) * 8;
  const int o_off1 = li * 256 + (((head * 4 + 2 + (g >> 1)) ^ li) << 4) + (g & 1) * 8;

  f32x4 xr[2][4];   // rows that are not fetched keep whatever (finite) values the registers held: their LayerNorm is discarded
#pragma unroll
  for (int it = 0; it < 2; ++it)
#pragma unroll
    for (int k = 0; k < 4; ++k) xr[it][k] = f32x4{0.f, 0.f, 0.f, 0.f};
  unsigned ok = 0;
  Win nxt;           // the window whose rows are in flight: decoded once, used again when it becomes the current one
  auto prefetch = [&](int widx) {
    ok = 0;
    const bool live = widx < p.n_win;
    nxt = win_decode(p, live ? widx : 0);
    const Win w = nxt;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int t = it * 32 + ln_row;
      int sy, sx;
      const int64_t tok = win_token(p, w, t, sy, sx);
      const bool rd = live && t < WT && tok >= 0;
      if (rd) {
        const float* src = p.x + tok * BC + lj * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) xr[it][k] = *reinterpret_cast<const f32x4*>(src + k * 32);
        ok |= 1u << it;
      }
    }
  };

  const int Hp = p.nWy * WS, Wp = p.nWx * WS;
  const float scale2 = 0.17677669529663687f * 1.4426950408889634f;   // 32^-0.5 * log2(e): base-2 softmax on (q.k + bias / scale)
  const float* be_head = p.bias_exp + head * 4096;   // uniform
  const int be_lane = li * 64 + g * 4;

  int w0 = blockIdx.x * 2;
  prefetch(w0 + grp);
  __syncthreads();   // the weight image is complete

  for (; w0 < p.n_win; w0 += gridDim.x * 2) {
    const int widx = w0 + grp;
    const bool valid = widx < p.n_win;
    const unsigned long long c0 = now();
    const Win win = nxt;
    const int wx = win.wx, wy = win.wy;

    // ---- LayerNorm -> bf16 tile (two-pass statistics; a row's 128 channels sit in 8 lanes x 16 values) -----------------------
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      float s1 = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) s1 += (xr[it][k][0] + xr[it][k][1]) + (xr[it][k][2] + xr[it][k][3]);
      const float mean = sum8(s1) * (1.0f / BC);
      float s2 = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = xr[it][k][e] - mean; s2 += d * d; }
      const float rstd = 1.0f / sqrtf(sum8(s2) * (1.0f / BC) + p.eps);
      const bool real = (ok >> it) & 1u;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const f32x4 g4 = *reinterpret_cast<const f32x4*>(v_g + k * 32 + lj * 4);
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(v_b + k * 32 + lj * 4);
        bf16x4 y = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
        if (real) {
#pragma unroll
          for (int e = 0; e < 4; ++e) y[e] = (bf16_t)((xr[it][k][e] - mean) * rstd * g4[e] + b4[e]);
        }
        *reinterpret_cast<bf16x4*>(tile + it * 32 * 256 + (ln_off ^ (k << 6))) = y;
      }
    }
    prefetch(w0 + (int)gridDim.x * 2 + grp);   // the next window's rows land while this one is computed
    const unsigned long long c1 = now();
    lds_barrier();   // B1: tile = LayerNorm(x) of the window
    const unsigned long long c2 = now();

    // ---- operand fragments of the tile: xf[tt][ks] = rows tt*16 + li, channels ks*32 + g*8 .. +8 --------------------------
    bf16x8 xf[4][4];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) xf[tt][ks] = *reinterpret_cast<const bf16x8*>(tile + tt * 4096 + cx[ks]);

    bf16x8 kf[4], qf[4], vf[2][2];
    {  // k^T [dim][token]
      f32x4 acc[2][4];
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) acc[dt][tt] = *reinterpret_cast<const f32x4*>(v_qb + BC + head * 32 + dt * 16 + g * 4);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wl + (BC + dt * 16) * 256 + cx[ks]);
#pragma unroll
          for (int tt = 0; tt < 4; ++tt) MM::mma(acc[dt][tt], wf, xf[tt][ks]);
        }
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) kf[tt] = pack8(acc[0][tt], acc[1][tt]);
    }
    {  // v [token][dim]
      f32x4 acc[4][2];
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) { const float bv = v_qb[2 * BC + head * 32 + dt * 16 + li]; acc[tt][dt] = f32x4{bv, bv, bv, bv}; }
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wl + (2 * BC + dt * 16) * 256 + cx[ks]);
#pragma unroll
          for (int tt = 0; tt < 4; ++tt) MM::mma(acc[tt][dt], xf[tt][ks], wf);
        }
#pragma unroll
      for (int ps = 0; ps < 2; ++ps)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) vf[ps][dt] = pack8(acc[2 * ps][dt], acc[2 * ps + 1][dt]);
    }
    {  // q^T [dim][token]
      f32x4 acc[2][4];
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) acc[dt][tt] = *reinterpret_cast<const f32x4*>(v_qb + head * 32 + dt * 16 + g * 4);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wl + dt * 16 * 256 + cx[ks]);
#pragma unroll
          for (int tt = 0; tt < 4; ++tt) MM::mma(acc[dt][tt], wf, xf[tt][ks]);
        }
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) qf[tt] = pack8(acc[0][tt], acc[1][tt]);
    }

    const unsigned long long c3 = now();
    // ---- attention of this head, as swin_attn_mfma_kernel<bf16, EXPB> (swin_attn.hip) --------------------------------------
    // SW-MSA: only the last window row / column of the padded grid mixes regions (swin_transformer.py:369-387)
    const bool edge = p.shift > 0 && (wy == p.nWy - 1 || wx == p.nWx - 1);
    unsigned long long krid = 0;   // region id (0..8) of this lane's 16 keys, 4 bits each
    if (edge) {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = kt * 16 + g * 4 + r;
          const int ty = (j * 37) >> 8, tx = j - ty * WS;
          const int ssy = wy * WS + ty, ssx = wx * WS + tx;
          const int ry = ssy < Hp - WS ? 0 : (ssy < Hp - p.shift ? 1 : 2);
          const int rx = ssx < Wp - WS ? 0 : (ssx < Wp - p.shift ? 1 : 2);
          krid |= (unsigned long long)(ry * 3 + rx) << ((kt * 4 + r) * 4);
        }
    }
    f32x4 oacc[2][4];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4) oacc[dt][t4] = f32x4{0.f, 0.f, 0.f, 0.f};
    // per-iteration copy of the lane offset: keeps the 16 bias addresses of a window out of loop-invariant registers
    int be_l;
    asm volatile("v_mov_b32 %0, %1" : "=v"(be_l) : "v"(be_lane));
    const float* be = be_head + be_l;
    f32x4 bnext[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) bnext[kt] = *reinterpret_cast<const f32x4*>(be + kt * 16);
    int64_t qtok[4];
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) {
      int i = t4 * 16 + li;
      const bool real_q = i < WT;
      if (i > WT - 1) i = WT - 1;            // clamped rows are never stored
      int sy, sx;
      const int64_t tok = win_token(p, win, i, sy, sx);
      qtok[t4] = (real_q && valid) ? tok : (int64_t)-1;
      int rid_i = 0;
      if (edge) {
        const int ry = sy < Hp - WS ? 0 : (sy < Hp - p.shift ? 1 : 2);
        const int rx = sx < Wp - WS ? 0 : (sx < Wp - p.shift ? 1 : 2);
        rid_i = ry * 3 + rx;
      }
      f32x4 bcur[4];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) bcur[kt] = bnext[kt];
      if (t4 < 3) {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) bnext[kt] = *reinterpret_cast<const f32x4*>(be + (t4 + 1) * 16 * 64 + kt * 16);
      }
      float sc[16];
      float mx = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        f32x4 st = bcur[kt];
        MM::mma(st, kf[kt], qf[t4]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float a = st[r] * scale2;   // (q.k + bias / scale) * scale * log2 e; padding keys are -inf through the seed
          if (edge && (int)((krid >> ((kt * 4 + r) * 4)) & 15) != rid_i) a += -100.0f * 1.4426950408889634f;
          sc[kt * 4 + r] = a;
          mx = fmaxf(mx, a);
        }
      }
      mx = quad_group_max(mx);
      float l = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        sc[k] = __builtin_amdgcn_exp2f(sc[k] - mx);
        l += sc[k];
      }
      l = quad_group_sum(l);
      const float inv = 1.0f / l;
#pragma unroll
      for (int k = 0; k < 16; ++k) sc[k] *= inv;
#pragma unroll
      for (int ps = 0; ps < 2; ++ps) {
        const bf16x8 pf = {(bf16_t)sc[ps * 8 + 0], (bf16_t)sc[ps * 8 + 1], (bf16_t)sc[ps * 8 + 2], (bf16_t)sc[ps * 8 + 3],
                           (bf16_t)sc[ps * 8 + 4], (bf16_t)sc[ps * 8 + 5], (bf16_t)sc[ps * 8 + 6], (bf16_t)sc[ps * 8 + 7]};
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) MM::mma(oacc[dt][t4], vf[ps][dt], pf);
      }
      }

    const unsigned long long c4 = now();
    lds_barrier();   // B2: every wave of the window has taken its xf fragments; the tile becomes O
    // O^T accumulators [dim dt*16 + 4g + r][query t4*16 + li] -> tile[query][head*32 + dim] (bf16)
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const f32x4 o = oacc[dt][t4];
        const bf16x4 ov = {(bf16_t)o[0], (bf16_t)o[1], (bf16_t)o[2], (bf16_t)o[3]};
        *reinterpret_cast<bf16x4*>(tile + t4 * 4096 + (dt ? o_off1 : o_off0)) = ov;
      }
    // residual rows of this lane's output pieces (L2: the window's rows were fetched one iteration ago)
    f32x4 res[4][2];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        res[tt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (qtok[tt] >= 0) res[tt][nt] = *reinterpret_cast<const f32x4*>(p.x + qtok[tt] * BC + head * 32 + nt * 16 + g * 4);
      }
    // proj: this wave produces output channels [32 head, +32) of its window: rows of Wp as A operands (L2-resident, 8 KB per wave)
    bf16x8 wp[2][4];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        wp[nt][ks] = *reinterpret_cast<const bf16x8*>(p.proj_w + (head * 32 + nt * 16 + li) * BC + ks * 32 + g * 8);
    lds_barrier();   // B3: tile = O of the window, all heads
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) xf[tt][ks] = *reinterpret_cast<const bf16x8*>(tile + tt * 4096 + cx[ks]);
    lds_barrier();   // B4: the tile may be overwritten by the next window's LayerNorm
    const unsigned long long c5 = now();

    // ---- out^T = Wp O^T + b + x: accumulator [channel head*32 + nt*16 + 4g + r][token tt*16 + li] -----------------------------
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        f32x4 acc = *reinterpret_cast<const f32x4*>(v_pb + head * 32 + nt * 16 + g * 4);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) MM::mma(acc, wp[nt][ks], xf[tt][ks]);
        if (qtok[tt] >= 0) {
          const f32x4 r = res[tt][nt];
          *reinterpret_cast<f32x4*>(p.out + qtok[tt] * BC + head * 32 + nt * 16 + g * 4) = f32x4{acc[0] + r[0], acc[1] + r[1], acc[2] + r[2], acc[3] + r[3]};
        }
      }
    if constexpr (TRACE) {
      const unsigned long long c6 = now();
      tr[1] += c1 - c0; tr[2] += c2 - c1; tr[3] += c3 - c2; tr[4] += c4 - c3; tr[5] += c5 - c4; tr[6] += c6 - c5;
    }
  }
  if constexpr (TRACE) {
    // wave 0: total | LayerNorm + prefetch issue | wait B1 | fragments + q k v | attention | B2, O, loads, B3, fragments, B4 | proj + stores | start
    if (threadIdx.x == 0 && p.trace != nullptr) {
      unsigned long long* t = p.trace + (long long)blockIdx.x * 8;
      t[0] = now() - t_start; t[1] = tr[1]; t[2] = tr[2]; t[3] = tr[3]; t[4] = tr[4]; t[5] = tr[5]; t[6] = tr[6]; t[7] = t_start;
    }
  }
}


// ---------------------------------------------------------------------------------------------
// The same block at C = 256 with 8 heads (Swin-B stage 1).  393 KB of qkv weights + 128 KB of proj weights fit neither LDS nor
// registers: a workgroup is ONE window x 8 heads, and every wave streams its head's operand fragments straight from the
// fragment-major image written by model/packing.py::pack_attn_block (1 KB contiguous per fragment, L2-resident, one k-step
// ahead of the matrix cores).  The products run k-step-outer so that only the four token fragments of one k-step are live.
// ---------------------------------------------------------------------------------------------
constexpr int C2 = 256, KS2 = 8;
constexpr int TILE2_BYTES = 64 * 512;
constexpr int VEC2_FLOATS = 3 * C2 + C2 + C2 + C2;   // qkv bias | proj bias | norm1 gamma | norm1 beta
constexpr int LDS2_BYTES = TILE2_BYTES + VEC2_FLOATS * 4;

struct SwinBlock256P {
  const float* x; float* out;
  const float* ln_g; const float* ln_b; float eps;
  const bf16_t* wpack; const float* qkv_b;
  const float* bias_exp;                 // [8 heads][64][64]
  const float* proj_b;
  int B, H, W, shift, nWy, nWx, n_win;
  unsigned long long* trace;
};

__device__ __forceinline__ float sum16(float v) {
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  v += dpp_mov<0x141>(v);   // row_half_mirror
  v += dpp_mov<0x140>(v);   // row_mirror: the other 8 of the 16-lane row
  return v;
}

template <bool TRACE>
__global__ __launch_bounds__(512, 1) void swin_block256_kernel(SwinBlock256P p) {
  typedef Mma<bf16_t> MM;
  unsigned long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  auto now = [&]() -> unsigned long long { if constexpr (TRACE) return __builtin_amdgcn_s_memtime(); else return 0ull; };
  const unsigned long long t_start = now();
  extern __shared__ __attribute__((aligned(16))) char lds[];   // bf16 tile [64][256], 16-byte chunk c of row r at slot c ^ (r & 15) | vectors
  const int tid = threadIdx.x, lane = tid & 63;
  const int head = __builtin_amdgcn_readfirstlane(tid >> 6);    // wave = head, 0..7
  const int li = lane & 15, g = lane >> 4;
  char* tile = lds;
  const float* vec = reinterpret_cast<const float*>(lds + TILE2_BYTES);
  const float* v_qb = vec; const float* v_pb = vec + 3 * C2; const float* v_g = vec + 4 * C2; const float* v_b = vec + 5 * C2;
  for (int idx = tid; idx < VEC2_FLOATS; idx += 512) {
    float v;
    if (idx < 3 * C2) v = p.qkv_b[idx];
    else if (idx < 4 * C2) v = p.proj_b[idx - 3 * C2];
    else if (idx < 5 * C2) v = p.ln_g[idx - 4 * C2];
    else v = p.ln_b[idx - 5 * C2];
    const_cast<float*>(vec)[idx] = v;
  }
  // fragment of rows R + li (R a multiple of 16), k-step ks: byte offset R * 512 + (cx0 ^ (ks << 6))
  const int cx0 = li * 512 + ((g ^ li) << 4);
  // LayerNorm: 16 lanes per token (lane j of the 16 holds the 16-byte pieces j, j + 16, j + 32, j + 48 of the 1 KB row); wave `head`
  // normalises tokens pass*32 + head*4 + tk
  const int tk = lane >> 4, lj = lane & 15;
  const int ln_row = head * 4 + tk;
  const int ln_off = ln_row * 512 + (((lj >> 1) ^ (ln_row & 15)) << 4) + (lj & 1) * 8;   // piece k enters as ^ (k << 7)
  const int o_off0 = li * 512 + (((head * 4 + (g >> 1)) ^ li) << 4) + (g & 1) * 8;
  const int o_off1 = li * 512 + (((head * 4 + 2 + (g >> 1)) ^ li) << 4) + (g & 1) * 8;
  const bf16_t* wq = p.wpack + (int64_t)head * (KS2 * 6 * 512) + lane * 8;                       // fragment (ks*6 + sel*2 + dt) of this head
  const bf16_t* wpj = p.wpack + 3 * C2 * C2 + (int64_t)head * (2 * KS2 * 512) + lane * 8;       // fragment (nt*KS2 + ks) of this wave's 32 output channels

  f32x4 xr[2][4];
#pragma unroll
  for (int it = 0; it < 2; ++it)
#pragma unroll
    for (int k = 0; k < 4; ++k) xr[it][k] = f32x4{0.f, 0.f, 0.f, 0.f};
  unsigned ok = 0;
  Win nxt;
  SwinBlockP geo;   // win_token reads the grid geometry from the stage-0 parameter block
  geo.H = p.H; geo.W = p.W; geo.shift = p.shift; geo.nWy = p.nWy; geo.nWx = p.nWx;
  auto prefetch = [&](int widx) {
    ok = 0;
    const bool live = widx < p.n_win;
    nxt = win_decode(geo, live ? widx : 0);
    const Win w = nxt;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int t = it * 32 + ln_row;
      int sy, sx;
      const int64_t tok = win_token(geo, w, t, sy, sx);
      const bool rd = live && t < WT && tok >= 0;
      if (rd) {
        const float* src = p.x + tok * C2 + lj * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) xr[it][k] = *reinterpret_cast<const f32x4*>(src + k * 64);
        ok |= 1u << it;
      }
    }
  };

  const int Hp = p.nWy * WS, Wp = p.nWx * WS;
  const float scale2 = 0.17677669529663687f * 1.4426950408889634f;
  const float* be_head = p.bias_exp + head * 4096;
  const int be_lane = li * 64 + g * 4;

  int w0 = blockIdx.x;
  prefetch(w0);
  __syncthreads();

  for (; w0 < p.n_win; w0 += gridDim.x) {
    const bool valid = true;
    const unsigned long long c0 = now();
    const Win win = nxt;
    const int wx = win.wx, wy = win.wy;

    // ---- LayerNorm -> bf16 tile ------------------------------------------------------------------------------------------------
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      float s1 = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) s1 += (xr[it][k][0] + xr[it][k][1]) + (xr[it][k][2] + xr[it][k][3]);
      const float mean = sum16(s1) * (1.0f / C2);
      float s2 = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = xr[it][k][e] - mean; s2 += d * d; }
      const float rstd = 1.0f / sqrtf(sum16(s2) * (1.0f / C2) + p.eps);
      const bool real = (ok >> it) & 1u;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const f32x4 g4 = *reinterpret_cast<const f32x4*>(v_g + k * 64 + lj * 4);
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(v_b + k * 64 + lj * 4);
        bf16x4 y = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
        if (real) {
#pragma unroll
          for (int e = 0; e < 4; ++e) y[e] = (bf16_t)((xr[it][k][e] - mean) * rstd * g4[e] + b4[e]);
        }
        *reinterpret_cast<bf16x4*>(tile + it * 32 * 512 + (ln_off ^ (k << 7))) = y;
      }
    }
    prefetch(w0 + (int)gridDim.x);
    const unsigned long long c1 = now();
    lds_barrier();   // B1: tile = LayerNorm(x) of the window
    const unsigned long long c2 = now();

    // ---- k^T, v, q^T of this head, k-step outer; the six weight fragments of the next k-step are in flight under the 24 products --
    bf16x8 kf[4], qf[4], vf[2][2];
    {
      f32x4 ka[2][4], qa[2][4], va[4][2];
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
          qa[dt][tt] = *reinterpret_cast<const f32x4*>(v_qb + head * 32 + dt * 16 + g * 4);
          ka[dt][tt] = *reinterpret_cast<const f32x4*>(v_qb + C2 + head * 32 + dt * 16 + g * 4);
          const float bv = v_qb[2 * C2 + head * 32 + dt * 16 + li];
          va[tt][dt] = f32x4{bv, bv, bv, bv};
        }
      bf16x8 wf[6];
#pragma unroll
      for (int f = 0; f < 6; ++f) wf[f] = *reinterpret_cast<const bf16x8*>(wq + f * 512);
#pragma unroll 1
      for (int ks = 0; ks < KS2; ++ks) {   // rolled: the fully unrolled form hoists all 48 fragment loads and spills
        bf16x8 wn[6];
        if (ks + 1 < KS2) {
#pragma unroll
          for (int f = 0; f < 6; ++f) wn[f] = *reinterpret_cast<const bf16x8*>(wq + ((ks + 1) * 6 + f) * 512);
        }
        bf16x8 xf[4];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) xf[tt] = *reinterpret_cast<const bf16x8*>(tile + tt * 8192 + (cx0 ^ (ks << 6)));
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
          MM::mma(qa[0][tt], wf[0], xf[tt]);
          MM::mma(qa[1][tt], wf[1], xf[tt]);
          MM::mma(ka[0][tt], wf[2], xf[tt]);
          MM::mma(ka[1][tt], wf[3], xf[tt]);
          MM::mma(va[tt][0], xf[tt], wf[4]);
          MM::mma(va[tt][1], xf[tt], wf[5]);
        }
        if (ks + 1 < KS2) {
#pragma unroll
          for (int f = 0; f < 6; ++f) wf[f] = wn[f];
        }
      }
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) { kf[tt] = pack8(ka[0][tt], ka[1][tt]); qf[tt] = pack8(qa[0][tt], qa[1][tt]); }
#pragma unroll
      for (int ps = 0; ps < 2; ++ps)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) vf[ps][dt] = pack8(va[2 * ps][dt], va[2 * ps + 1][dt]);
    }
    const unsigned long long c3 = now();
    // ---- attention of this head, as swin_attn_mfma_kernel<bf16, EXPB> (swin_attn.hip) --------------------------------------
    // SW-MSA: only the last window row / column of the padded grid mixes regions (swin_transformer.py:369-387)
    const bool edge = p.shift > 0 && (wy == p.nWy - 1 || wx == p.nWx - 1);
    unsigned long long krid = 0;   // region id (0..8) of this lane's 16 keys, 4 bits each
    if (edge) {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = kt * 16 + g * 4 + r;
          const int ty = (j * 37) >> 8, tx = j - ty * WS;
          const int ssy = wy * WS + ty, ssx = wx * WS + tx;
          const int ry = ssy < Hp - WS ? 0 : (ssy < Hp - p.shift ? 1 : 2);
          const int rx = ssx < Wp - WS ? 0 : (ssx < Wp - p.shift ? 1 : 2);
          krid |= (unsigned long long)(ry * 3 + rx) << ((kt * 4 + r) * 4);
        }
    }
    f32x4 oacc[2][4];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4) oacc[dt][t4] = f32x4{0.f, 0.f, 0.f, 0.f};
    // per-iteration copy of the lane offset: keeps the 16 bias addresses of a window out of loop-invariant registers
    int be_l;
    asm volatile("v_mov_b32 %0, %1" : "=v"(be_l) : "v"(be_lane));
    const float* be = be_head + be_l;
    f32x4 bnext[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) bnext[kt] = *reinterpret_cast<const f32x4*>(be + kt * 16);
    int64_t qtok[4];
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) {
      int i = t4 * 16 + li;
      const bool real_q = i < WT;
      if (i > WT - 1) i = WT - 1;            // clamped rows are never stored
      int sy, sx;
      const int64_t tok = win_token(geo, win, i, sy, sx);
      qtok[t4] = (real_q && valid) ? tok : (int64_t)-1;
      int rid_i = 0;
      if (edge) {
        const int ry = sy < Hp - WS ? 0 : (sy < Hp - p.shift ? 1 : 2);
        const int rx = sx < Wp - WS ? 0 : (sx < Wp - p.shift ? 1 : 2);
        rid_i = ry * 3 + rx;
      }
      f32x4 bcur[4];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) bcur[kt] = bnext[kt];
      if (t4 < 3) {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) bnext[kt] = *reinterpret_cast<const f32x4*>(be + (t4 + 1) * 16 * 64 + kt * 16);
      }
      float sc[16];
      float mx = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        f32x4 st = bcur[kt];
        MM::mma(st, kf[kt], qf[t4]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float a = st[r] * scale2;   // (q.k + bias / scale) * scale * log2 e; padding keys are -inf through the seed
          if (edge && (int)((krid >> ((kt * 4 + r) * 4)) & 15) != rid_i) a += -100.0f * 1.4426950408889634f;
          sc[kt * 4 + r] = a;
          mx = fmaxf(mx, a);
        }
      }
      mx = quad_group_max(mx);
      float l = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        sc[k] = __builtin_amdgcn_exp2f(sc[k] - mx);
        l += sc[k];
      }
      l = quad_group_sum(l);
      const float inv = 1.0f / l;
#pragma unroll
      for (int k = 0; k < 16; ++k) sc[k] *= inv;
#pragma unroll
      for (int ps = 0; ps < 2; ++ps) {
        const bf16x8 pf = {(bf16_t)sc[ps * 8 + 0], (bf16_t)sc[ps * 8 + 1], (bf16_t)sc[ps * 8 + 2], (bf16_t)sc[ps * 8 + 3],
                           (bf16_t)sc[ps * 8 + 4], (bf16_t)sc[ps * 8 + 5], (bf16_t)sc[ps * 8 + 6], (bf16_t)sc[ps * 8 + 7]};
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) MM::mma(oacc[dt][t4], vf[ps][dt], pf);
      }
      }

    const unsigned long long c4 = now();
    lds_barrier();   // B2: every head has taken its token fragments; the tile becomes O
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const f32x4 o = oacc[dt][t4];
        const bf16x4 ov = {(bf16_t)o[0], (bf16_t)o[1], (bf16_t)o[2], (bf16_t)o[3]};
        *reinterpret_cast<bf16x4*>(tile + t4 * 8192 + (dt ? o_off1 : o_off0)) = ov;
      }
    f32x4 res[4][2];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        res[tt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (qtok[tt] >= 0) res[tt][nt] = *reinterpret_cast<const f32x4*>(p.x + qtok[tt] * C2 + head * 32 + nt * 16 + g * 4);
      }
    bf16x8 pw[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) pw[nt] = *reinterpret_cast<const bf16x8*>(wpj + (nt * KS2) * 512);
    lds_barrier();   // B3: tile = O of the window, all heads

    // ---- out^T = Wp O^T + b + x, k-step outer -----------------------------------------------------------------------------------
    f32x4 pa[2][4];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) pa[nt][tt] = *reinterpret_cast<const f32x4*>(v_pb + head * 32 + nt * 16 + g * 4);
#pragma unroll 1
    for (int ks = 0; ks < KS2; ++ks) {
      bf16x8 pn[2];
      if (ks + 1 < KS2) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) pn[nt] = *reinterpret_cast<const bf16x8*>(wpj + (nt * KS2 + ks + 1) * 512);
      }
      bf16x8 of[4];
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) of[tt] = *reinterpret_cast<const bf16x8*>(tile + tt * 8192 + (cx0 ^ (ks << 6)));
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) MM::mma(pa[nt][tt], pw[nt], of[tt]);
      if (ks + 1 < KS2) { pw[0] = pn[0]; pw[1] = pn[1]; }
    }
    lds_barrier();   // B4: the tile may be overwritten by the next window's LayerNorm
    const unsigned long long c5 = now();
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
        if (qtok[tt] >= 0) {
          const f32x4 r = res[tt][nt], a = pa[nt][tt];
          *reinterpret_cast<f32x4*>(p.out + qtok[tt] * C2 + head * 32 + nt * 16 + g * 4) = f32x4{a[0] + r[0], a[1] + r[1], a[2] + r[2], a[3] + r[3]};
        }
    if constexpr (TRACE) {
      const unsigned long long c6 = now();
      tr[1] += c1 - c0; tr[2] += c2 - c1; tr[3] += c3 - c2; tr[4] += c4 - c3; tr[5] += c5 - c4; tr[6] += c6 - c5;
    }
  }
  if constexpr (TRACE) {
    if (threadIdx.x == 0 && p.trace != nullptr) {
      unsigned long long* t = p.trace + (long long)blockIdx.x * 8;
      t[0] = now() - t_start; t[1] = tr[1]; t[2] = tr[2]; t[3] = tr[3]; t[4] = tr[4]; t[5] = tr[5]; t[6] = tr[6]; t[7] = t_start;
    }
  }
}

}  // namespace

// Per-device launch state, safe under concurrent pipeline lanes: the CU count of the CURRENT device and "dynamic LDS
// limit raised" flags, in atomics indexed by device ordinal.  A lost race only repeats an idempotent query / attribute call.
namespace {
constexpr int MAX_DEVS = 64;
inline int current_device_cus(int* dev_out) {
  static std::atomic<int> cus[MAX_DEVS];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVS) return -1;
  *dev_out = dev;
  int v = cus[dev].load(std::memory_order_relaxed);
  if (v == 0) {
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    cus[dev].store(v, std::memory_order_relaxed);
  }
  return v;
}
template <typename K>
inline bool raise_lds_limit_once(std::atomic<bool>* done, int dev, K kern_a, K kern_b) {
  if (done[dev].load(std::memory_order_acquire)) return true;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern_a), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern_b), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
    return false;
  done[dev].store(true, std::memory_order_release);
  return true;
}
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
}  // namespace

// x (fp32 [B*H*W, 128]) -> out = x + proj(W-MSA / SW-MSA(LayerNorm(x))) for Swin-B stage-0 geometry (C = 128, 4 heads, window 7);
// out may be x.  Replaces LayerNorm + qkv GEMM + window attention + proj GEMM of the bf16 engine (swin_transformer.py:196-253).
extern "C" int omp_swin_attn_block(const void* x, void* out, const float* ln_g, const float* ln_b, float eps, const void* qkv_w,
                                   const float* qkv_b, const float* bias_expanded, const void* proj_w, const float* proj_b, int B, int H,
                                   int W, int C, int nH, int window, int shift, omp_stream_t s) {
  OMP_CHECK_ARG(x && out && ln_g && ln_b && qkv_w && qkv_b && bias_expanded && proj_w && proj_b, "omp_swin_attn_block: null pointer");
  OMP_CHECK_ARG(C == BC && nH == 4, "omp_swin_attn_block: built for C = 128 with 4 heads (got C=%d nH=%d)", C, nH);
  OMP_CHECK_ARG(window == WS, "omp_swin_attn_block: only window 7 is built (got %d)", window);
  OMP_CHECK_ARG(shift >= 0 && shift < WS, "omp_swin_attn_block: bad shift %d", shift);
  OMP_CHECK_ARG(B > 0 && H > 0 && W > 0, "omp_swin_attn_block: bad shape");
  SwinBlockP p;
  p.x = (const float*)x; p.out = (float*)out; p.ln_g = ln_g; p.ln_b = ln_b; p.eps = eps;
  p.qkv_w = (const bf16_t*)qkv_w; p.qkv_b = qkv_b; p.bias_exp = bias_expanded; p.proj_w = (const bf16_t*)proj_w; p.proj_b = proj_b;
  p.B = B; p.H = H; p.W = W; p.shift = shift;
  p.nWy = (H + WS - 1) / WS; p.nWx = (W + WS - 1) / WS;
  const int64_t nw = (int64_t)B * p.nWy * p.nWx;
  OMP_CHECK_ARG(nw < (int64_t)1 << 30, "omp_swin_attn_block: too many windows");
  p.n_win = (int)nw;
  OMP_CHECK_ARG(aligned16(x) && aligned16(out) && aligned16(qkv_w) && aligned16(proj_w) && aligned16(bias_expanded),
                "omp_swin_attn_block: x / out / qkv_w / proj_w / bias_expanded must be 16-byte aligned (the kernel moves 16-byte vectors)");
  int dev = 0;
  const int n_cu = current_device_cus(&dev);
  static std::atomic<bool> lds_done[MAX_DEVS];
  if (n_cu <= 0 || !raise_lds_limit_once(lds_done, dev, swin_block_kernel<false>, swin_block_kernel<true>)) {
    omp_set_error("omp_swin_attn_block: cannot query the device / raise the dynamic LDS limit");
    return OMP_ERR_LAUNCH;
  }
  const int pairs = (int)((nw + 1) / 2);
  const int grid = pairs < n_cu ? pairs : n_cu;
  p.trace = omp_cur().mlp_trace;   // omp_debug_swin_mlp_trace: the development buffer also takes this kernel's phase sums
  if (p.trace != nullptr) hipLaunchKernelGGL(swin_block_kernel<true>, dim3((unsigned)grid), dim3(512), LDS_BYTES, (hipStream_t)s, p);
  else hipLaunchKernelGGL(swin_block_kernel<false>, dim3((unsigned)grid), dim3(512), LDS_BYTES, (hipStream_t)s, p);
  OMP_CHECK_LAUNCH("omp_swin_attn_block");
  return OMP_OK;
}

// The same block for C = 256 with 8 heads (Swin-B stage 1); wpack = model/packing.py::pack_attn_block(qkv.weight, proj.weight, 8).
extern "C" int omp_swin_attn_block_packed(const void* x, void* out, const float* ln_g, const float* ln_b, float eps, const void* wpack,
                                          const float* qkv_b, const float* bias_expanded, const float* proj_b, int B, int H, int W, int C,
                                          int nH, int window, int shift, omp_stream_t s) {
  OMP_CHECK_ARG(x && out && ln_g && ln_b && wpack && qkv_b && bias_expanded && proj_b, "omp_swin_attn_block_packed: null pointer");
  OMP_CHECK_ARG(C == C2 && nH == 8, "omp_swin_attn_block_packed: built for C = 256 with 8 heads (got C=%d nH=%d)", C, nH);
  OMP_CHECK_ARG(window == WS, "omp_swin_attn_block_packed: only window 7 is built (got %d)", window);
  OMP_CHECK_ARG(shift >= 0 && shift < WS, "omp_swin_attn_block_packed: bad shift %d", shift);
  OMP_CHECK_ARG(B > 0 && H > 0 && W > 0, "omp_swin_attn_block_packed: bad shape");
  SwinBlock256P p;
  p.x = (const float*)x; p.out = (float*)out; p.ln_g = ln_g; p.ln_b = ln_b; p.eps = eps;
  p.wpack = (const bf16_t*)wpack; p.qkv_b = qkv_b; p.bias_exp = bias_expanded; p.proj_b = proj_b;
  p.B = B; p.H = H; p.W = W; p.shift = shift;
  p.nWy = (H + WS - 1) / WS; p.nWx = (W + WS - 1) / WS;
  const int64_t nw = (int64_t)B * p.nWy * p.nWx;
  OMP_CHECK_ARG(nw < (int64_t)1 << 30, "omp_swin_attn_block_packed: too many windows");
  p.n_win = (int)nw;
  OMP_CHECK_ARG(aligned16(x) && aligned16(out) && aligned16(wpack) && aligned16(bias_expanded),
                "omp_swin_attn_block_packed: x / out / wpack / bias_expanded must be 16-byte aligned (the kernel moves 16-byte vectors)");
  int dev = 0;
  const int n_cu = current_device_cus(&dev);
  if (n_cu <= 0) {
    omp_set_error("omp_swin_attn_block_packed: cannot query the device");
    return OMP_ERR_LAUNCH;
  }
  const int grid = (int)(nw < n_cu ? nw : n_cu);
  p.trace = omp_cur().mlp_trace;
  if (p.trace != nullptr) hipLaunchKernelGGL(swin_block256_kernel<true>, dim3((unsigned)grid), dim3(512), LDS2_BYTES, (hipStream_t)s, p);
  else hipLaunchKernelGGL(swin_block256_kernel<false>, dim3((unsigned)grid), dim3(512), LDS2_BYTES, (hipStream_t)s, p);
  OMP_CHECK_LAUNCH("omp_swin_attn_block_packed");
  return OMP_OK;
}
