// MGP-STR specific kernels (BASELINE config 5; reference OCR/MGP-STR/modules/mgp_str.py:64-94).
//
// The ViT-B encoder of MGP-STR runs on the kernels the OmniParser path already has -- LayerNorm, the DMA GEMM
// (q / k / v / proj / fc1+GELU / fc2 with fused bias, activation and residual) and, for the 257-token
// self-attention, the blocked-K / blocked-V^T cross-attention kernels (a ViT layer's keys and values are written
// by the k / v projection epilogues straight into the slabs those kernels stream; an image's 257 tokens are 5
// row groups that share its slab).  What has no counterpart there lives in this file:
//   vit_attn_kernel          bf16 self-attention of one (image, head) per workgroup: the 257 keys / values of the
//                            head stay in LDS, every wave owns pairs of 16-query tiles (see the kernel)
//   vit_patch_embed_kernel   4x4/4 patchify (a K = 48 dot product per output) + bias + cls token + pos_embed
//   a3_pool_kernel           the A^3 module's token softmax and weighted pooling (token_learner.py:27-31)
//   row_argmax_prob_kernel   greedy id and max-softmax probability of every logits row (test_final.py:145-170)
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------------
// ViT self-attention, bf16 (timm Attention.forward: softmax(q k^T / 8) v per head; modules/mgp_str.py:71-74 runs it
// 12 times on 257 tokens).  One workgroup = one (image, head):
//   * the head's K rows [Mpad][64] and V^T blocks [Mpad/32][64][32] (the slabs the k / v projection epilogues write,
//     same layout as the decoder's cross-attention) are copied ONCE into LDS by DMA, 128-byte rows with the 16-byte
//     chunk index XOR-swizzled on the source address so that the fragment ds_read_b128s are conflict-free;
//   * a wave owns 16-query tiles, two at a time (each K / V^T fragment read from LDS feeds two matrix-core
//     products): S^T = K Q^T for 96 keys at a time into registers, masked running max / exp / sum in registers
//     (a query's keys sit in one lane of each of the 4 lane groups: two shuffles per chunk), P^T repacked to bf16
//     in the order the V^T slab stores its keys, O^T = alpha O^T + V^T P^T, scaled by 1 / sum on the way out.
// Through the blocked cross-attention kernels the same layer cost 5 row groups x (K + V^T streamed from L2) per
// (image, head) and ran its last group for one query.
// QKV (round 6): q, K and V are the three column groups of ONE token-major projection [B T, 3 nH 64] (timm's fused qkv Linear, mgp_str.py:72-73) --
// K = q + nH 64, Vt = q + 2 nH 64, row pitch ldq.  The key rows go to LDS by the same DMA (source rows ldq apart, rows beyond T clamped: masked
// below), and the V^T image is BUILT here: every thread loads 16-byte pieces of value rows (8 dims of a key) and scatters them as eight 2-byte LDS
// writes into the blocked, slot-permuted layout the P V product reads (keys beyond T: zeros).  The blocked slabs cost the projections dearly: the
// V^T slab's store path degenerates to one 2-byte store with two integer divisions per element when the tokens per image (257) are not a multiple
// of 4 -- 540 us per block against 200 for the plain product of the same size (profiles/r06j_mgp_str_kernel_shapes.txt).
template <int NB, bool QKV = false>   // 16-key score blocks (Mpad = 16 NB; NB even)
__global__ __launch_bounds__(256, 2) void vit_attn_kernel(const bf16_t* __restrict__ q, int64_t ldq, const bf16_t* __restrict__ K,
                                                          const bf16_t* __restrict__ Vt, bf16_t* __restrict__ out, int64_t ldo,
                                                          int T, int nH) {
  typedef bf16x8 frag;
  constexpr int MP = NB * 16, KBYTES = MP * 128, ROUNDS = MP / 32;
  constexpr int CB = 6;   // score blocks (16 keys each) per softmax chunk: 2 x 6 x 4 score registers per lane
  static_assert(NB % CB == 0 && CB % 2 == 0, "chunking");
  extern __shared__ __attribute__((aligned(16))) char lds[];   // K image | V^T image, MP rows of 128 B each
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), li = lane & 15, g = lane >> 4;
  const int h = blockIdx.x, b = blockIdx.y;
  const int64_t slab = ((int64_t)b * nH + h) * MP * 64;
  if constexpr (QKV) {
    const int dr = lane >> 3, dc = (lane & 7) ^ dr;
    const bf16_t* kb_ = K + (int64_t)b * T * ldq + h * 64 + dc * 8;
#pragma unroll
    for (int i = 0; i < ROUNDS; ++i) {
      int key = i * 32 + wave * 8 + dr;
      if (key > T - 1) key = T - 1;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kb_ + (int64_t)key * ldq),
                                       (__attribute__((address_space(3))) void*)(lds + (i * 32 + wave * 8) * 128), 16, 0, 0);
    }
    // V^T image: element (key j = 32 s + kl, dim d) -> row 32 s + d / 2, byte (d & 1) * 64 + 2 slot(kl), 16-byte chunks XOR-swizzled by the row
    constexpr int PIECES = MP * 8, PER = (PIECES + 255) / 256;
    const bf16_t* vb_ = Vt + (int64_t)b * T * ldq + h * 64;
    frag vr[PER];
#pragma unroll
    for (int it = 0; it < PER; ++it) {
      const int idx = it * 256 + (int)threadIdx.x, j = idx >> 3, c8 = idx & 7;
      frag z;
#pragma unroll
      for (int u = 0; u < 8; ++u) z[u] = (bf16_t)0.f;
      vr[it] = (idx < PIECES && j < T) ? *reinterpret_cast<const frag*>(vb_ + (int64_t)j * ldq + c8 * 8) : z;
    }
#pragma unroll
    for (int it = 0; it < PER; ++it) {
      const int idx = it * 256 + (int)threadIdx.x, j = idx >> 3, c8 = idx & 7;
      if (idx < PIECES) {
        const int s_ = j >> 5, kl = j & 31;
        const int slot = ((kl & 15) >> 2) * 8 + (kl >> 4) * 4 + (kl & 3);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int d = c8 * 8 + u, rw = d >> 1, byte = (d & 1) * 64 + slot * 2;
          *reinterpret_cast<bf16_t*>(lds + KBYTES + (s_ * 32 + rw) * 128 + ((((byte >> 4) ^ (rw & 7))) << 4) + (byte & 15)) = vr[it][u];
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  } else {
    const int dr = lane >> 3, dc = (lane & 7) ^ dr;
    const bf16_t* ks = K + slab + (int64_t)(wave * 8 + dr) * 64 + dc * 8;
    const bf16_t* vs = Vt + slab + (int64_t)(wave * 8 + dr) * 64 + dc * 8;
#pragma unroll
    for (int i = 0; i < ROUNDS; ++i) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ks + i * 32 * 64),
                                       (__attribute__((address_space(3))) void*)(lds + (i * 32 + wave * 8) * 128), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vs + i * 32 * 64),
                                       (__attribute__((address_space(3))) void*)(lds + KBYTES + (i * 32 + wave * 8) * 128), 16, 0, 0);
    }
  }
  // fragment byte offsets: K block kb, k-step st -> row kb*16 + li, chunk st*4 + g; V^T key step s, dim tile dt ->
  // row s*32 + dt*8 + (li >> 1), chunk (li & 1)*4 + g (a 128-byte row holds two dims x 32 key slots)
  const int koff0 = li * 128 + (((0 * 4 + g) ^ (li & 7)) << 4), koff1 = li * 128 + (((1 * 4 + g) ^ (li & 7)) << 4);
  int voff[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    const int rw = dt * 8 + (li >> 1), c = (li & 1) * 4 + g;
    voff[dt] = KBYTES + rw * 128 + ((c ^ (rw & 7)) << 4);
  }
  const int nq = (T + 15) >> 4;          // query tiles
  const int first_dead = T >> 4;         // first score block with a masked key
  const bf16_t* qb = q + (int64_t)b * T * ldq + h * 64 + g * 8;
  bf16_t* ob = out + (int64_t)b * T * ldo + h * 64 + g * 4;

  auto tile = [&](auto NQ_, int f0) {
    constexpr int NQ = decltype(NQ_)::value;
    constexpr float L2E = 1.4426950408889634f;
    frag qf[NQ][2];
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
      int qi = (f0 + n) * 16 + li;
      if (qi > T - 1) qi = T - 1;
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        float tmp[8];
        unpack16(*reinterpret_cast<const frag*>(qb + (int64_t)qi * ldq + st * 32), tmp);
#pragma unroll
        for (int i = 0; i < 8; ++i) tmp[i] *= 0.125f;   // 1 / sqrt(64), exact in bf16
        pack16(tmp, qf[n][st]);
      }
    }
    float mrun[NQ], lrun[NQ];
    f32x4 ot[NQ][4];
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
      mrun[n] = -INFINITY; lrun[n] = 0.f;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) ot[n][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int c0 = 0; c0 < NB; c0 += CB) {
      // compiler barrier: the LDS images are loop-invariant, and hipcc otherwise hoists every fragment read of a tile
      // (36 + 36 x 4 registers) out of the tile loop and spills them
      asm volatile("" ::: "memory");
      f32x4 sc[NQ][CB];
#pragma unroll
      for (int kb = 0; kb < CB; ++kb) {
        const frag k0 = *reinterpret_cast<const frag*>(lds + (c0 + kb) * 2048 + koff0);
        const frag k1 = *reinterpret_cast<const frag*>(lds + (c0 + kb) * 2048 + koff1);
#pragma unroll
        for (int n = 0; n < NQ; ++n) {
          f32x4 a = {0.f, 0.f, 0.f, 0.f};
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, qf[n][0], a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, qf[n][1], a, 0, 0, 0);
          sc[n][kb] = a;
        }
      }
      // keys >= T (the zero rows that pad the slab) take no weight
#pragma unroll
      for (int kb = 0; kb < CB; ++kb)
        if (c0 + kb >= first_dead) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool dead = (c0 + kb) * 16 + g * 4 + r >= T;
#pragma unroll
            for (int n = 0; n < NQ; ++n) sc[n][kb][r] = dead ? -INFINITY : sc[n][kb][r];
          }
        }
#pragma unroll
      for (int n = 0; n < NQ; ++n) {
        float m = sc[n][0][0];
#pragma unroll
        for (int kb = 0; kb < CB; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r) m = fmaxf(m, sc[n][kb][r]);
        m = quad_group_max(m);
        const float mn = fmaxf(mrun[n], m);
        const float mref = (mn == -INFINITY) ? 0.f : mn;    // a chunk of padding only: exp(-inf - 0) = 0, not NaN
        const float alpha = __builtin_amdgcn_exp2f((mrun[n] - mref) * L2E);
        mrun[n] = mn;
        const float ml = mref * L2E;
        float l = 0.f;
#pragma unroll
        for (int kb = 0; kb < CB; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e = __builtin_amdgcn_exp2f(fmaf(sc[n][kb][r], L2E, -ml));
            sc[n][kb][r] = e;
            l += e;
          }
        lrun[n] = lrun[n] * alpha + l;
        if (c0 > 0) {
#pragma unroll
          for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) ot[n][dt][r] *= alpha;
        }
      }
#pragma unroll
      for (int s = 0; s < CB / 2; ++s) {
        frag pf[NQ];
#pragma unroll
        for (int n = 0; n < NQ; ++n)
#pragma unroll
          for (int r = 0; r < 4; ++r) { pf[n][r] = (bf16_t)sc[n][2 * s][r]; pf[n][4 + r] = (bf16_t)sc[n][2 * s + 1][r]; }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const frag vc = *reinterpret_cast<const frag*>(lds + (c0 / 2 + s) * 4096 + voff[dt]);
#pragma unroll
          for (int n = 0; n < NQ; ++n) ot[n][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vc, pf[n], ot[n][dt], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
      const float l = quad_group_sum(lrun[n]);
      const float inv = 1.0f / l;
      const int qi = (f0 + n) * 16 + li;
      if (qi < T) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
          *reinterpret_cast<bf16x4*>(ob + (int64_t)qi * ldo + dt * 16) =
              bf16x4{(bf16_t)(ot[n][dt][0] * inv), (bf16_t)(ot[n][dt][1] * inv), (bf16_t)(ot[n][dt][2] * inv),
                     (bf16_t)(ot[n][dt][3] * inv)};
      }
    }
  };

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  // pairs of query tiles round-robin over the waves; an odd last tile goes, alone, to the wave after the last pair
  const int npair = nq >> 1;
  for (int pr = wave; pr < npair; pr += 4) tile(std::integral_constant<int, 2>(), 2 * pr);
  if ((nq & 1) && (npair & 3) == wave) tile(std::integral_constant<int, 1>(), nq - 1);
}

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

// The tiles of a row (omp_gemm_bias_act, OMP_STORE_ROWSTAT) -> greedy id and its softmax probability.  One wave per row: the row's maximum over the
// tiles' maxima (lowest column on ties: tiles ascend with the columns), then sum_t s_t exp(m_t - max).
__global__ __launch_bounds__(256) void row_stat_merge_kernel(const f32x4* __restrict__ stats, int R, int nt, int32_t* __restrict__ ids,
                                                             float* __restrict__ prob) {
  const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const f32x4* sr = stats + (int64_t)r * nt;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int t = lane; t < nt; t += 64) {
    const f32x4 v = sr[t];
    const int i = __float_as_int(v[2]);
    if (v[0] > best || (v[0] == best && i < bi)) { best = v[0]; bi = i; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  float sum = 0.f;
  for (int t = lane; t < nt; t += 64) {
    const f32x4 v = sr[t];
    sum += v[1] * expf(v[0] - best);
  }
  sum = wave_sum(sum);
  if (lane == 0) {
    ids[r] = bi;
    prob[r] = 1.0f / sum;
  }
}

}  // namespace

extern "C" int omp_row_stat_merge(const float* stats, int R, int n_tiles, int32_t* ids, float* prob, omp_stream_t s) {
  OMP_CHECK_ARG(stats && ids && prob, "omp_row_stat_merge: null pointer");
  OMP_CHECK_ARG(R > 0 && n_tiles > 0 && ((uintptr_t)stats % 16) == 0, "omp_row_stat_merge: bad shape / alignment");
  hipLaunchKernelGGL(row_stat_merge_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)s, reinterpret_cast<const f32x4*>(stats), R, n_tiles, ids, prob);
  OMP_CHECK_LAUNCH("omp_row_stat_merge");
  return OMP_OK;
}

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

extern "C" int omp_vit_attn(const void* q, int64_t ldq, const void* K, const void* Vt, int Mpad, void* out, int64_t ldo,
                            int dtype, int B, int T, int nH, omp_stream_t s) {
  OMP_CHECK_ARG(q && K && Vt && out, "omp_vit_attn: null pointer");
  OMP_CHECK_ARG(B > 0 && T > 0 && nH > 0 && T <= Mpad, "omp_vit_attn: bad shape B=%d T=%d nH=%d Mpad=%d", B, T, nH, Mpad);
  OMP_CHECK_ARG(ldq % 8 == 0 && ldo % 4 == 0, "omp_vit_attn: ldq %% 8, ldo %% 4");
  if (dtype != OMP_BF16 || Mpad != 288) {
    omp_set_error("omp_vit_attn: built for bf16 and Mpad = 288 (257 tokens); use omp_dec_cross_attn_step otherwise");
    return OMP_ERR_UNSUPPORTED;
  }
  constexpr int NB = 18;
  constexpr size_t smem = 2 * (size_t)NB * 16 * 128;
  auto kern = vit_attn_kernel<NB>;
  static bool done = false;
  if (!done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
      omp_set_error("omp_vit_attn: cannot raise dynamic LDS limit");
      return OMP_ERR_LAUNCH;
    }
    done = true;
  }
  hipLaunchKernelGGL(kern, dim3(nH, B), dim3(256), smem, (hipStream_t)s, (const bf16_t*)q, ldq, (const bf16_t*)K,
                     (const bf16_t*)Vt, (bf16_t*)out, ldo, T, nH);
  OMP_CHECK_LAUNCH("omp_vit_attn");
  return OMP_OK;
}

extern "C" int omp_vit_attn_qkv(const void* qkv, int64_t ld, void* out, int64_t ldo, int dtype, int B, int T, int nH, omp_stream_t s) {
  OMP_CHECK_ARG(qkv && out, "omp_vit_attn_qkv: null pointer");
  OMP_CHECK_ARG(B > 0 && T > 0 && nH > 0 && ld >= 3 * (int64_t)nH * 64 && ld % 8 == 0 && ldo % 4 == 0 && ((uintptr_t)qkv % 16) == 0,
                "omp_vit_attn_qkv: qkv is [B T, 3 nH 64] with a row pitch that is a multiple of 8 elements (B=%d T=%d nH=%d ld=%lld)", B, T, nH, (long long)ld);
  if (dtype != OMP_BF16 || T > 288) {
    omp_set_error("omp_vit_attn_qkv: built for bf16 and at most 288 tokens per image (MGP-STR's ViT: 257)");
    return OMP_ERR_UNSUPPORTED;
  }
  constexpr int NB = 18;
  constexpr size_t smem = 2 * (size_t)NB * 16 * 128;
  auto kern = vit_attn_kernel<NB, true>;
  static bool done = false;
  if (!done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
      omp_set_error("omp_vit_attn_qkv: cannot raise dynamic LDS limit");
      return OMP_ERR_LAUNCH;
    }
    done = true;
  }
  const bf16_t* q = reinterpret_cast<const bf16_t*>(qkv);
  hipLaunchKernelGGL(kern, dim3(nH, B), dim3(256), smem, (hipStream_t)s, q, ld, q + nH * 64, q + 2 * nH * 64, (bf16_t*)out, ldo, T, nH);
  OMP_CHECK_LAUNCH("omp_vit_attn_qkv");
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
