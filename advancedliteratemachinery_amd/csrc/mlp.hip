// Fused Swin MLP for gfx950 (bf16):   y = x + fc2(GELU(fc1(LayerNorm(x))))
//
// Replaces, in one launch, `x = x + self.drop_path(self.mlp(self.norm2(x)))` of SwinTransformerBlock.forward
// (reference OCR/OmniParser/model/backbone/swin_transformer.py:250) with Mlp.forward (:30-36) inlined -- three launches
// (LayerNorm, fc1 + GELU, fc2 + residual) and a [tokens, 4C] hidden tensor through HBM in round 1.  For the early
// stages that tensor WAS the cost: at C = 128 the two GEMMs move 1.48 GB per 8 images for 137 GFLOP; fused, the
// kernel reads x once (twice counting the residual, an L2 hit) and writes y: 268 MB.
//
// Structure ("row-stationary"): a wave owns RG groups of 16 token rows for its whole life.
//   * the rows' LayerNorm'ed values live in REGISTERS as matrix-core B-operand fragments (lane l: row l & 15,
//     k-chunk l >> 4), loaded straight from global memory and normalised in place with xor-shuffles over the four
//     lanes that share a row;
//   * the weights stream through LDS in sub-chunks of 32 hidden units: one contiguous, host-packed image per
//     sub-chunk (model/packing.py: fc1 rows in swizzled 128-byte K-tile rows | fc2 columns in matrix-core order |
//     fc1 bias slice), copied by DMA (global_load_lds_dwordx4) into an NS-stage ring, one raw s_barrier per
//     sub-chunk, counted vmcnt -- the weight stream never touches a VGPR;
//   * first product  D1[hidden i][row j] = sum_k W1[i][k] * xn[j][k]  (A = fc1 fragment from LDS, B = row fragment);
//     a lane of the accumulator then holds hidden units 4g..4g+3 (g = lane >> 4) of its row j = lane & 15 for both
//     16-wide hidden tiles of the sub-chunk.  +bias, GELU (erf form, common.h), round to bf16: those 8 values, in the
//     order fc2 was packed in, ARE the B fragment of the second product
//        D2[feature n][row j] += sum_h W2[n][h] * gelu[j][h],
//     so the hidden activations never leave registers -- no LDS round trip, no barrier between the two products;
//   * epilogue: + fc2 bias + residual (re-read from x, an L2 hit), 8-byte stores (a lane owns 4 consecutive
//     features of one row).
// The fp32 engine (the parity gate) keeps the unfused path; the matrix-core operands of this kernel are bf16.
// TX = type of the residual stream x / y: bf16, or float (round 3: the bf16 engine carries its residual stream in fp32 so
// that 24 blocks of residual adds do not each round to 8 mantissa bits; the rows are then loaded as fp32, normalised with
// exact two-pass statistics in registers and only the normalised values are rounded to bf16 fragments).
#include <type_traits>

#include "common.h"

namespace {

struct MlpP {
  const void* X; int64_t ldx;
  const float* ln_g; const float* ln_b; float eps;
  const char* Wp;      // packed sub-chunks, C*128 + 1024 bytes each
  const float* b2;
  void* Y; int64_t ldy;
  int64_t M; int nsub;
  unsigned long long* trace;   // development (TRACE instantiation): [workgroup][8] cycle sums of wave 0
  int x_f32;                   // X / Y are fp32 rows (fp32 residual stream of the bf16 engine)
};

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// WPS = waves per SIMD the register allocation is held to (VGPR + AGPR <= 512 / WPS)
template <typename TX, int C, int RG, int NW, int NS, int WPS, bool TRACE = false>
__global__ __launch_bounds__(64 * NW, WPS) void mlp_fused_kernel(MlpP p) {
  constexpr bool XF32 = sizeof(TX) == 4;
  const TX* X = reinterpret_cast<const TX*>(p.X);
  TX* Y = reinterpret_cast<TX*>(p.Y);
  // TRACE: s_memtime sums per phase (0 whole, 1 rows + LayerNorm, 2 DMA wait + barrier, 3 first product, 4 GELU,
  // 5 second product, 6 epilogue); the stamps serialise the LDS queue, so the TOTAL is pessimistic, the split is the point
  unsigned long long tr[7] = {0, 0, 0, 0, 0, 0, 0};
  auto now = [&]() -> unsigned long long { if constexpr (TRACE) return __builtin_amdgcn_s_memtime(); else return 0ull; };
  const unsigned long long t_start = now();
  constexpr int KS = C / 32;                 // k-steps of the first product
  constexpr int NT = C / 16;                 // feature tiles of the second product
  constexpr int BLK = C * 128 + 1024;        // bytes per sub-chunk image
  constexpr int PIECES = BLK / 1024;         // 1 KB DMA wave-instructions per sub-chunk
  constexpr int PPW = (PIECES + NW - 1) / NW;  // per wave (the same count for every wave: counted vmcnt)
  static_assert(NS >= 2 && NS <= 4, "2..4 ring stages");
  static_assert(PPW * (NS - 2) <= 63, "vmcnt immediate");
  extern __shared__ __attribute__((aligned(16))) char smem[];   // NS * BLK

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int64_t m0 = ((int64_t)blockIdx.x * NW + wave) * (RG * 16);

  // ---- weight ring: issue / wait -------------------------------------------------------------------------------
  auto issue = [&](int hc, int stage) {
    const char* src = p.Wp + (int64_t)hc * BLK + lane * 16;
    char* dst = smem + stage * BLK;
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
      int pc = wave + q * NW;
      if (pc > PIECES - 1) pc = PIECES - 1;   // surplus slots re-copy the last piece (same bytes, same place)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + pc * 1024),
                                       (__attribute__((address_space(3))) void*)(dst + pc * 1024), 16, 0, 0);
    }
  };

  // ---- the rows: load, LayerNorm in registers, keep as B fragments -----------------------------------------------
  bf16x8 xf[RG][KS];
  if constexpr (!XF32) {
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) {
      int64_t m = m0 + rg * 16 + li;
      if (m > p.M - 1) m = p.M - 1;            // clamped rows are computed and never stored
      const bf16_t* xr = X + m * p.ldx + lg * 8;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) xf[rg][ks] = *reinterpret_cast<const bf16x8*>(xr + ks * 32);
    }
    // the first ring stages do not depend on the rows: request them now, under the LayerNorm arithmetic
#pragma unroll
    for (int t = 0; t < NS - 1; ++t)
      if (t < p.nsub) issue(t, t);
    float mean[RG], rstd[RG];
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) {
      float s = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) s += (float)xf[rg][ks][e];
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      mean[rg] = s / (float)C;
      float q = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = (float)xf[rg][ks][e] - mean[rg]; q += d * d; }
      q += __shfl_xor(q, 16, 64);
      q += __shfl_xor(q, 32, 64);
      rstd[rg] = 1.0f / sqrtf(q / (float)C + p.eps);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(p.ln_g + ks * 32 + lg * 8);
      const f32x4 g1 = *reinterpret_cast<const f32x4*>(p.ln_g + ks * 32 + lg * 8 + 4);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.ln_b + ks * 32 + lg * 8);
      const f32x4 b1 = *reinterpret_cast<const f32x4*>(p.ln_b + ks * 32 + lg * 8 + 4);
#pragma unroll
      for (int rg = 0; rg < RG; ++rg) {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] = (bf16_t)(((float)xf[rg][ks][e] - mean[rg]) * rstd[rg] * g0[e] + b0[e]);
          o[e + 4] = (bf16_t)(((float)xf[rg][ks][e + 4] - mean[rg]) * rstd[rg] * g1[e] + b1[e]);
        }
        xf[rg][ks] = o;
      }
    }
  } else {
    // fp32 residual stream: one row group at a time (the fp32 values of a group live only until its fragments are made)
#pragma unroll
    for (int t = 0; t < NS - 1; ++t)
      if (t < p.nsub) issue(t, t);
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) {
      int64_t m = m0 + rg * 16 + li;
      if (m > p.M - 1) m = p.M - 1;
      const float* xr = reinterpret_cast<const float*>(X) + m * p.ldx + lg * 8;
      f32x4 v[KS][2];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        v[ks][0] = *reinterpret_cast<const f32x4*>(xr + ks * 32);
        v[ks][1] = *reinterpret_cast<const f32x4*>(xr + ks * 32 + 4);
      }
      float s = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int e = 0; e < 4; ++e) s += v[ks][0][e] + v[ks][1][e];
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      const float mean = s / (float)C;
      float q = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d0 = v[ks][0][e] - mean, d1 = v[ks][1][e] - mean;
          q += d0 * d0 + d1 * d1;
        }
      q += __shfl_xor(q, 16, 64);
      q += __shfl_xor(q, 32, 64);
      const float rstd = 1.0f / sqrtf(q / (float)C + p.eps);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(p.ln_g + ks * 32 + lg * 8);
        const f32x4 g1 = *reinterpret_cast<const f32x4*>(p.ln_g + ks * 32 + lg * 8 + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.ln_b + ks * 32 + lg * 8);
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(p.ln_b + ks * 32 + lg * 8 + 4);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] = (bf16_t)((v[ks][0][e] - mean) * rstd * g0[e] + b0[e]);
          o[e + 4] = (bf16_t)((v[ks][1][e] - mean) * rstd * g1[e] + b1[e]);
        }
        xf[rg][ks] = o;
      }
    }
  }

  tr[1] = now() - t_start;
  f32x4 acc2[NT][RG];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) acc2[nt][rg] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int w2_off = C * 64 + (li * 4 + (lg ^ ((-(li >> 2)) & 3))) * 16;   // this lane's fragment inside a 16-row block of the fc2 image

  auto compute = [&](int stage) {
    const char* w1 = smem + stage * BLK;
    const float* b1s = reinterpret_cast<const float*>(w1 + C * 128);
    const unsigned long long c0 = now();
    f32x4 a1[2][RG];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int rg = 0; rg < RG; ++rg) a1[t][rg] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int r = 16 * t + li;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int kt = ks >> 1, c = (ks & 1) * 4 + lg;
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(w1 + ((kt * 32 + r) * 8 + (c ^ (r & 7))) * 16);
#pragma unroll
        for (int rg = 0; rg < RG; ++rg)
          a1[t][rg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, xf[rg][ks], a1[t][rg], 0, 0, 0);
      }
    }
    if constexpr (TRACE) { asm volatile("" :: "v"(a1[0][0]), "v"(a1[1][RG - 1])); }
    const unsigned long long c1 = now();
    const f32x4 bb0 = *reinterpret_cast<const f32x4*>(b1s + 4 * lg);
    const f32x4 bb1 = *reinterpret_cast<const f32x4*>(b1s + 16 + 4 * lg);
    bf16x8 hf[RG];
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) {
      float hv[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) { hv[e] = a1[0][rg][e] + bb0[e]; hv[e + 4] = a1[1][rg][e] + bb1[e]; }
      gelu_fast_n<8>(hv);
#pragma unroll
      for (int e = 0; e < 8; ++e) hf[rg][e] = (bf16_t)hv[e];
    }
    if constexpr (TRACE) { asm volatile("" :: "v"(hf[0]), "v"(hf[RG - 1])); }
    const unsigned long long c2 = now();
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(w1 + w2_off + nt * 1024);
#pragma unroll
      for (int rg = 0; rg < RG; ++rg)
        acc2[nt][rg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, hf[rg], acc2[nt][rg], 0, 0, 0);
    }
    if constexpr (TRACE) {
      asm volatile("" :: "v"(acc2[0][0]), "v"(acc2[NT - 1][RG - 1]));
      const unsigned long long c3 = now();
      tr[3] += c1 - c0; tr[4] += c2 - c1; tr[5] += c3 - c2;
    }
  };

  // ---- ring over the hidden sub-chunks (same protocol as gemm_dma: wait own DMA of chunk hc, barrier -- everybody's
  // part has landed and everybody is done with chunk hc-1 -- refill that stage, multiply chunk hc) -------------------
  int st_c = 0, st_i = NS - 1;
  for (int hc = 0; hc < p.nsub; ++hc) {
    const int after = p.nsub - 1 - hc;
    const unsigned long long w0 = now();
    if (NS == 2 || after == 0) wait_vm<0>();
    else if (NS == 3 || after == 1) wait_vm<PPW>();
    else wait_vm<PPW * (NS > 3 ? 2 : 1)>();
    __builtin_amdgcn_s_barrier();
    tr[2] += now() - w0;
    if (hc + NS - 1 < p.nsub) issue(hc + NS - 1, st_i);
    compute(st_c);
    st_c = (st_c + 1 == NS) ? 0 : st_c + 1;
    st_i = (st_i + 1 == NS) ? 0 : st_i + 1;
  }

  // ---- epilogue: + fc2 bias + residual, 8-byte stores ------------------------------------------------------------
  const unsigned long long e0 = now();
#pragma unroll
  for (int rg = 0; rg < RG; ++rg) {
    const int64_t m = m0 + rg * 16 + li;
    if (m < p.M) {
      const TX* xr = X + m * p.ldx + lg * 4;
      TX* yr = Y + m * p.ldy + lg * 4;
      constexpr int EG = 4;   // feature tiles per batch of residual loads (bounds the live registers)
      typedef typename std::conditional<XF32, f32x4, bf16x4>::type xv4;
#pragma unroll
      for (int n0 = 0; n0 < NT; n0 += EG) {
        xv4 res[EG];
        f32x4 bo[EG];
#pragma unroll
        for (int u = 0; u < EG; ++u) {
          res[u] = *reinterpret_cast<const xv4*>(xr + (n0 + u) * 16);
          bo[u] = *reinterpret_cast<const f32x4*>(p.b2 + (n0 + u) * 16 + lg * 4);
        }
#pragma unroll
        for (int u = 0; u < EG; ++u) {
          xv4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (TX)(acc2[n0 + u][rg][e] + bo[u][e] + (float)res[u][e]);
          *reinterpret_cast<xv4*>(yr + (n0 + u) * 16) = o;
        }
      }
    }
  }
  if constexpr (TRACE) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t_end = now();
    if (threadIdx.x == 0 && p.trace != nullptr) {
      unsigned long long* t = p.trace + (long long)blockIdx.x * 8;
      t[0] = t_end - t_start; t[1] = tr[1]; t[2] = tr[2]; t[3] = tr[3]; t[4] = tr[4]; t[5] = tr[5]; t[6] = t_end - e0; t[7] = t_start;
    }
  }
}

template <typename TX, int C, int RG, int NW, int NS, int WPS, bool TRACE = false>
int launch_mlp_t(const MlpP& p, hipStream_t st) {
  constexpr size_t smem = (size_t)NS * (C * 128 + 1024);
  auto kern = mlp_fused_kernel<TX, C, RG, NW, NS, WPS, TRACE>;
  static bool done = false;   // per template instantiation
  if (!done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
      omp_set_error("omp_swin_mlp_fused: cannot raise dynamic LDS limit");
      return OMP_ERR_LAUNCH;
    }
    done = true;
  }
  const int64_t rows_per_wg = (int64_t)NW * RG * 16;
  hipLaunchKernelGGL(kern, dim3((unsigned)ceil_div64(p.M, rows_per_wg)), dim3(64 * NW), smem, st, p);
  OMP_CHECK_LAUNCH("omp_swin_mlp_fused");
  return OMP_OK;
}

template <int C, int RG, int NW, int NS, int WPS, bool TRACE = false>
int launch_mlp(const MlpP& p, hipStream_t st) {
  if (p.x_f32) return launch_mlp_t<float, C, RG, NW, NS, WPS, TRACE>(p, st);
  return launch_mlp_t<bf16_t, C, RG, NW, NS, WPS, TRACE>(p, st);
}

int dispatch_mlp(const MlpP& p, int C, int v, hipStream_t st);

}  // namespace

extern "C" int omp_debug_swin_mlp_variant(int v) {
  omp_cur().mlp_variant = v;
  return OMP_OK;
}

extern "C" int omp_debug_swin_mlp_trace(void* buffer) {
  omp_cur().mlp_trace = reinterpret_cast<unsigned long long*>(buffer);
  return OMP_OK;
}

extern "C" int omp_swin_mlp_fused(const void* x, int64_t ldx, const float* ln_gamma, const float* ln_beta, float eps,
                                  const void* wpack, const float* b2, void* y, int64_t ldy, int64_t M, int C, int hidden,
                                  omp_stream_t s) {
  return omp_swin_mlp_fused2(x, OMP_BF16, ldx, ln_gamma, ln_beta, eps, wpack, b2, y, ldy, M, C, hidden, s);
}

extern "C" int omp_swin_mlp_fused2(const void* x, int x_dtype, int64_t ldx, const float* ln_gamma, const float* ln_beta, float eps,
                                   const void* wpack, const float* b2, void* y, int64_t ldy, int64_t M, int C, int hidden,
                                   omp_stream_t s) {
  OMP_CHECK_ARG(x_dtype == OMP_BF16 || x_dtype == OMP_F32, "omp_swin_mlp_fused: x_dtype must be bf16 or f32");
  OMP_CHECK_ARG(x && ln_gamma && ln_beta && wpack && b2 && y, "omp_swin_mlp_fused: null pointer");
  OMP_CHECK_ARG(M > 0 && M < (1ll << 31), "omp_swin_mlp_fused: bad M=%lld", (long long)M);
  OMP_CHECK_ARG(hidden > 0 && hidden % 32 == 0, "omp_swin_mlp_fused: hidden=%d must be a multiple of 32", hidden);
  OMP_CHECK_ARG(ldx % 8 == 0 && ldy % 8 == 0 && ldx >= C && ldy >= C, "omp_swin_mlp_fused: row pitches must be multiples of 8 elements");
  OMP_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)wpack % 16) == 0 && ((uintptr_t)b2 % 16) == 0 &&
                    ((uintptr_t)ln_gamma % 16) == 0 && ((uintptr_t)ln_beta % 16) == 0,
                "omp_swin_mlp_fused: pointers must be 16-byte aligned");
  MlpP p;
  p.X = x; p.ldx = ldx; p.x_f32 = x_dtype == OMP_F32 ? 1 : 0;
  p.ln_g = ln_gamma; p.ln_b = ln_beta; p.eps = eps;
  p.Wp = reinterpret_cast<const char*>(wpack); p.b2 = b2;
  p.Y = y; p.ldy = ldy;
  p.M = M; p.nsub = hidden / 32; p.trace = omp_cur().mlp_trace;
  hipStream_t st = (hipStream_t)s;
  const int v = omp_cur().mlp_variant;
  const int slot = omp_prof_active(OMP_PROF_MLP) ? omp_prof_begin(OMP_PROF_MLP, st, 4.0 * (double)M * C * hidden, (x_dtype == OMP_F32 ? 8.0 : 4.0) * (double)M * C + (double)(hidden / 32) * (C * 128 + 1024)) : -1;
  const int rc = dispatch_mlp(p, C, v, st);
  if (slot >= 0) omp_prof_end(OMP_PROF_MLP, slot, st);
  return rc;
}

namespace {
int dispatch_mlp(const MlpP& p, int C, int v, hipStream_t st) {
  switch (C) {   // variant 0 = the fastest measured (profiles/r02c_kbench_mlp.txt)
    case 128:
      if (v == 100) return launch_mlp<128, 2, 4, 3, 3, true>(p, st);
      if (v == 1) return launch_mlp<128, 2, 4, 4, 2>(p, st);
      if (v == 2) return launch_mlp<128, 2, 8, 4, 2>(p, st);
      if (v == 3) return launch_mlp<128, 4, 4, 3, 2>(p, st);
      return launch_mlp<128, 2, 4, 3, 3>(p, st);
    case 256:
      if (v == 100) return launch_mlp<256, 2, 4, 2, 2, true>(p, st);
      if (v == 1) return launch_mlp<256, 1, 4, 2, 2>(p, st);
      if (v == 2) return launch_mlp<256, 1, 8, 2, 2>(p, st);
      return launch_mlp<256, 2, 4, 2, 2>(p, st);
    case 512:
      if (v == 100) return launch_mlp<512, 1, 8, 2, 2, true>(p, st);
      if (v == 1) return launch_mlp<512, 1, 4, 2, 2>(p, st);
      return launch_mlp<512, 1, 8, 2, 2>(p, st);
    default:
      omp_set_error("omp_swin_mlp_fused: C=%d not built (128, 256, 512)", C);
      return OMP_ERR_UNSUPPORTED;
  }
}
}  // namespace
