// MGP-STR specific kernels (BASELINE config 5; reference OCR/MGP-STR/modules/mgp_str.py:64-94).
//
// The ViT-B encoder of MGP-STR runs on the kernels the OmniParser path already has -- LayerNorm, the DMA GEMM
// (q / k / v / proj / fc1+GELU / fc2 with fused bias, activation and residual) and, for the 257-token
// self-attention, the blocked-K / blocked-V^T cross-attention kernels (a ViT layer's keys and values are written
// by the k / v projection epilogues straight into the slabs those kernels stream; an image's 257 tokens are 5
// row groups that share its slab).  What has no counterpart there lives in this file:
//   vit_patch_embed_kernel   4x4/4 patchify (a K = 48 dot product per output) + bias + cls token + pos_embed
//   a3_pool_kernel           the A^3 module's token softmax and weighted pooling (token_learner.py:27-31)
//   row_argmax_prob_kernel   greedy id and max-softmax probability of every logits row (test_final.py:145-170)
#include "common.h"

namespace {

constexpr int VPE_TOK = 16;   // patch tokens per workgroup

// out[b, 0, :]     = cls + pos[0]
// out[b, 1 + p, e] = sum_{c,ky,kx} img[b, c, 4py+ky, 4px+kx] * w[e, c, ky, kx] + bias[e] + pos[1 + p, e]
// (timm PatchEmbed = Conv2d(3, E, 4, 4) -> flatten(2).transpose(1, 2); cls concat + pos add: mgp_str.py:66-70)
template <typename TO>
__global__ __launch_bounds__(256) void vit_patch_embed_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                              const float* __restrict__ bias, const float* __restrict__ cls,
                                                              const float* __restrict__ pos, TO* __restrict__ out, int B,
                                                              int H, int W, int Hp, int Wp, int E) {
  __shared__ __attribute__((aligned(16))) float patch[VPE_TOK * 48];
  const int tid = threadIdx.x;
  const int xb = blockIdx.x * VPE_TOK, ty = blockIdx.y, b = blockIdx.z;
  const int T = Hp * Wp + 1;
  for (int idx = tid; idx < VPE_TOK * 48; idx += 256) {
    const int t = idx / 48, e = idx - t * 48;
    const int ch = e >> 4, ky = (e >> 2) & 3, kx = e & 3;
    const int py = ty * 4 + ky, px = (xb + t) * 4 + kx;
    patch[idx] = (xb + t < Wp) ? img[(((int64_t)b * 3 + ch) * H + py) * W + px] : 0.f;
  }
  __syncthreads();
  for (int e = tid; e < E; e += 256) {
    float wr[48];
#pragma unroll
    for (int i = 0; i < 48; i += 4) {
      const f32x4 t4 = *reinterpret_cast<const f32x4*>(w + (int64_t)e * 48 + i);
      wr[i] = t4[0]; wr[i + 1] = t4[1]; wr[i + 2] = t4[2]; wr[i + 3] = t4[3];
    }
    const float bv = bias[e];
#pragma unroll 4
    for (int t = 0; t < VPE_TOK; ++t) {
      if (xb + t >= Wp) break;
      float a = 0.f;
#pragma unroll
      for (int i = 0; i < 48; i += 4) {
        const f32x4 p4 = *reinterpret_cast<const f32x4*>(patch + t * 48 + i);   // wave-wide broadcast
        a = fmaf(wr[i], p4[0], a);
        a = fmaf(wr[i + 1], p4[1], a);
        a = fmaf(wr[i + 2], p4[2], a);
        a = fmaf(wr[i + 3], p4[3], a);
      }
      const int tok = 1 + ty * Wp + xb + t;
      out[((int64_t)b * T + tok) * E + e] = from_f32<TO>(a + bv + pos[(int64_t)tok * E + e]);
    }
    if (blockIdx.x == 0 && ty == 0) out[(int64_t)b * T * E + e] = from_f32<TO>(cls[e] + pos[e]);
  }
}

// A^3 pooling of one image: maps[s, i] = softmax_i(sel[i, s]);  pooled[s, c] = sum_i maps[s, i] * feat[i, c].
// sel: fp32 [B*T, ld_sel] (token-major, S <= 28 columns used), feat: [B*T, C], pooled: fp32 [B*S, C],
// attn (optional): fp32 [B, S, T].  One workgroup per image; the S x T weights live in LDS (token-major, 28
// floats per token so that the inner loop reads them as 7 broadcast 16-byte loads), thread c owns channels
// c, c+256, c+512, ... and all S outputs of each.
constexpr int A3_SP = 28;
template <typename T, int CPT>
__global__ __launch_bounds__(256) void a3_pool_kernel(const float* __restrict__ sel, int ld_sel, const T* __restrict__ feat,
                                                      float* __restrict__ pooled, float* __restrict__ attn, int Tk, int S,
                                                      int C) {
  extern __shared__ __attribute__((aligned(16))) float pw[];   // [Tk][A3_SP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
  const float* sb = sel + (int64_t)b * Tk * ld_sel;
  for (int idx = tid; idx < Tk * A3_SP; idx += 256) {
    const int i = idx / A3_SP, s = idx - i * A3_SP;
    pw[idx] = s < S ? sb[(int64_t)i * ld_sel + s] : 0.f;
  }
  __syncthreads();
  for (int s = wave; s < S; s += 4) {   // softmax over the tokens, one wave per map
    float mx = -INFINITY;
    for (int i = lane; i < Tk; i += 64) mx = fmaxf(mx, pw[i * A3_SP + s]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int i = lane; i < Tk; i += 64) {
      const float e = expf(pw[i * A3_SP + s] - mx);
      pw[i * A3_SP + s] = e;
      sum += e;
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int i = lane; i < Tk; i += 64) {
      const float v = pw[i * A3_SP + s] * inv;
      pw[i * A3_SP + s] = v;
      if (attn != nullptr) attn[((int64_t)b * S + s) * Tk + i] = v;
    }
  }
  __syncthreads();
  float acc[A3_SP][CPT];
#pragma unroll
  for (int s = 0; s < A3_SP; ++s)
#pragma unroll
    for (int k = 0; k < CPT; ++k) acc[s][k] = 0.f;
  const T* fb = feat + (int64_t)b * Tk * C;
#pragma unroll 2
  for (int i = 0; i < Tk; ++i) {
    float f[CPT];
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
      const int c = tid + 256 * k;
      f[k] = c < C ? to_f32(fb[(int64_t)i * C + c]) : 0.f;
    }
#pragma unroll
    for (int s4 = 0; s4 < A3_SP; s4 += 4) {
      const f32x4 p4 = *reinterpret_cast<const f32x4*>(pw + i * A3_SP + s4);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int k = 0; k < CPT; ++k) acc[s4 + r][k] = fmaf(p4[r], f[k], acc[s4 + r][k]);
    }
  }
#pragma unroll
  for (int s = 0; s < A3_SP; ++s) {
    if (s < S) {
#pragma unroll
      for (int k = 0; k < CPT; ++k) {
        const int c = tid + 256 * k;
        if (c < C) pooled[((int64_t)b * S + s) * C + c] = acc[s][k];
      }
    }
  }
}

// ids[r] = argmax_v logits[r, v] (lowest index on ties), prob[r] = softmax(logits[r])[ids[r]].  One wave per row.
__global__ __launch_bounds__(256) void row_argmax_prob_kernel(const float* __restrict__ logits, int64_t ld, int R, int V,
                                                              int32_t* __restrict__ ids, float* __restrict__ prob) {
  const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const float* lg = logits + (int64_t)r * ld;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int t = lane; t < V; t += 64) {
    const float v = lg[t];
    if (v > best) { best = v; bi = t; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  float sum = 0.f;
  for (int t = lane; t < V; t += 64) sum += expf(lg[t] - best);
  sum = wave_sum(sum);
  if (lane == 0) {
    ids[r] = bi;
    prob[r] = 1.0f / sum;
  }
}

}  // namespace

extern "C" int omp_vit_patch_embed(const float* img, const float* w, const float* bias, const float* cls,
                                   const float* pos, void* out, int out_dtype, int B, int H, int W, int E,
                                   omp_stream_t s) {
  OMP_CHECK_ARG(img && w && bias && cls && pos && out, "omp_vit_patch_embed: null pointer");
  OMP_CHECK_ARG(B > 0 && H > 0 && W > 0 && E > 0, "omp_vit_patch_embed: bad shape");
  OMP_CHECK_ARG(H % 4 == 0 && W % 4 == 0, "omp_vit_patch_embed: image %dx%d is not a multiple of the 4x4 patch", H, W);
  const int Hp = H / 4, Wp = W / 4;
  dim3 grid((Wp + VPE_TOK - 1) / VPE_TOK, Hp, B);
  if (out_dtype == OMP_F32)
    hipLaunchKernelGGL((vit_patch_embed_kernel<float>), grid, dim3(256), 0, (hipStream_t)s, img, w, bias, cls, pos,
                       (float*)out, B, H, W, Hp, Wp, E);
  else if (out_dtype == OMP_BF16)
    hipLaunchKernelGGL((vit_patch_embed_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)s, img, w, bias, cls, pos,
                       (bf16_t*)out, B, H, W, Hp, Wp, E);
  else { omp_set_error("omp_vit_patch_embed: bad dtype %d", out_dtype); return OMP_ERR_INVALID; }
  OMP_CHECK_LAUNCH("omp_vit_patch_embed");
  return OMP_OK;
}

template <typename T>
static int launch_a3(const float* sel, int ld_sel, const void* feat, float* pooled, float* attn, int B, int Tk, int S,
                     int C, hipStream_t st) {
  const size_t smem = (size_t)Tk * A3_SP * sizeof(float);
  const int cpt = (C + 255) / 256;
  const T* f = reinterpret_cast<const T*>(feat);
#define A3_LAUNCH(CPT)                                                                                              \
  do {                                                                                                              \
    auto kern = a3_pool_kernel<T, CPT>;                                                                             \
    if (smem > 48 * 1024 &&                                                                                         \
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,        \
                            160 * 1024) != hipSuccess) {                                                            \
      omp_set_error("omp_a3_pool: cannot raise dynamic LDS limit");                                                 \
      return OMP_ERR_LAUNCH;                                                                                        \
    }                                                                                                               \
    hipLaunchKernelGGL(kern, dim3(B), dim3(256), smem, st, sel, ld_sel, f, pooled, attn, Tk, S, C);                 \
  } while (0)
  if (cpt == 1) A3_LAUNCH(1);
  else if (cpt == 2) A3_LAUNCH(2);
  else if (cpt == 3) A3_LAUNCH(3);
  else A3_LAUNCH(4);
#undef A3_LAUNCH
  return OMP_OK;
}

extern "C" int omp_a3_pool(const float* sel, int ld_sel, const void* feat, int dtype, float* pooled, float* attn,
                           int B, int T, int S, int C, omp_stream_t s) {
  OMP_CHECK_ARG(sel && feat && pooled, "omp_a3_pool: null pointer");
  OMP_CHECK_ARG(B > 0 && T > 0 && S > 0 && S <= A3_SP && ld_sel >= S, "omp_a3_pool: bad shape (S <= %d)", A3_SP);
  OMP_CHECK_ARG(C > 0 && C <= 1024, "omp_a3_pool: C must be in 1..1024 (got %d)", C);
  OMP_CHECK_ARG((size_t)T * A3_SP * sizeof(float) <= 150 * 1024, "omp_a3_pool: too many tokens (%d)", T);
  int rc;
  if (dtype == OMP_F32) rc = launch_a3<float>(sel, ld_sel, feat, pooled, attn, B, T, S, C, (hipStream_t)s);
  else if (dtype == OMP_BF16) rc = launch_a3<bf16_t>(sel, ld_sel, feat, pooled, attn, B, T, S, C, (hipStream_t)s);
  else { omp_set_error("omp_a3_pool: bad dtype %d", dtype); return OMP_ERR_INVALID; }
  if (rc != OMP_OK) return rc;
  OMP_CHECK_LAUNCH("omp_a3_pool");
  return OMP_OK;
}

extern "C" int omp_row_argmax_prob(const float* logits, int64_t ld, int R, int V, int32_t* ids, float* prob,
                                   omp_stream_t s) {
  OMP_CHECK_ARG(logits && ids && prob, "omp_row_argmax_prob: null pointer");
  OMP_CHECK_ARG(R > 0 && V > 0 && ld >= V, "omp_row_argmax_prob: bad shape");
  hipLaunchKernelGGL(row_argmax_prob_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)s, logits, ld, R, V, ids, prob);
  OMP_CHECK_LAUNCH("omp_row_argmax_prob");
  return OMP_OK;
}
