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

}  // namespace

extern "C" int omp_swin_window_attn(const void* qkv, const float* qkv_bias, const float* rel_bias_table,
                                    void* out, int dtype, int B, int H, int W, int C, int nH, int window,
                                    int shift, omp_stream_t s) {
  OMP_CHECK_ARG(qkv && qkv_bias && rel_bias_table && out, "omp_swin_window_attn: null pointer");
  OMP_CHECK_ARG(window == WS, "omp_swin_window_attn: only window 7 is built (got %d)", window);
  OMP_CHECK_ARG(shift >= 0 && shift < WS, "omp_swin_window_attn: bad shift %d", shift);
  OMP_CHECK_ARG(nH > 0 && C == nH * HD, "omp_swin_window_attn: head_dim must be 32 (C=%d nH=%d)", C, nH);
  OMP_CHECK_ARG(B > 0 && H > 0 && W > 0, "omp_swin_window_attn: bad shape");
  const int nWy = (H + WS - 1) / WS, nWx = (W + WS - 1) / WS;
  dim3 grid((unsigned)((int64_t)B * nWy * nWx), (unsigned)((nH + 3) / 4));
  if (dtype == OMP_F32)
    hipLaunchKernelGGL((swin_attn_kernel<float>), grid, dim3(256), 0, (hipStream_t)s, (const float*)qkv,
                       qkv_bias, rel_bias_table, (float*)out, B, H, W, C, nH, shift, nWy, nWx);
  else if (dtype == OMP_BF16)
    hipLaunchKernelGGL((swin_attn_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)s, (const bf16_t*)qkv,
                       qkv_bias, rel_bias_table, (bf16_t*)out, B, H, W, C, nH, shift, nWy, nWx);
  else { omp_set_error("omp_swin_window_attn: bad dtype %d", dtype); return OMP_ERR_INVALID; }
  OMP_CHECK_LAUNCH("omp_swin_window_attn");
  return OMP_OK;
}
