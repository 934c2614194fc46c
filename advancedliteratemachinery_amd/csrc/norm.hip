// LayerNorm family (HBM-bound): plain LN, PatchMerging gather+LN, PatchEmbed conv4x4+LN.
//
// Layout: a row of C channels is covered by LPR = 2^k <= 64 lanes, each lane owning IT 16-byte
// chunks (chunk c = sub + it*LPR), so a wave processes 64/LPR rows at once and every global access
// is a full 16-byte vector.  Row statistics are two-pass in registers (mean, then centred
// variance -- same numerics as ATen's LayerNorm), reduced with xor-shuffles inside the LPR group.
#include "common.h"

namespace {

template <typename TO, int NV>
__device__ __forceinline__ void store_vals(TO* p, const float* v) {
  if constexpr (sizeof(TO) == 4) {
#pragma unroll
    for (int i = 0; i < NV; i += 4) {
      f32x4 o = {v[i], v[i + 1], v[i + 2], v[i + 3]};
      *reinterpret_cast<f32x4*>(p + i) = o;
    }
  } else {
    if constexpr (NV == 8) {
      bf16x8 o;
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = (bf16_t)v[i];
      *reinterpret_cast<bf16x8*>(p) = o;
    } else {
      bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
      *reinterpret_cast<bf16x4*>(p) = o;
    }
  }
}

struct LnP {
  const void* x; const float* g; const float* b; void* y; float* yf;
  int64_t rows; int C; int lpr_log2; float eps;
  int split;   // y is bf16 and holds split pairs: row pitch 2C, hi at column c, lo at column C + c (OMP_BF16X2)
  // gather mode (PatchMerging): x is [B,H,W,Cin], row = (b, y2, x2), C = 4*Cin
  int gather; int H, W, Cin, H2, W2;
};

template <typename TI, typename TO, int IT>
__global__ __launch_bounds__(256) void ln_kernel(LnP p) {
  constexpr int NV = Vec16<TI>::N;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int LPR = 1 << p.lpr_log2;
  const int rpw = 64 >> p.lpr_log2;
  const int64_t row = ((int64_t)blockIdx.x * 4 + wave) * rpw + (lane >> p.lpr_log2);
  const int sub = lane & (LPR - 1);
  const bool rvalid = row < p.rows;
  const int nchunks = p.C / NV;
  const TI* X = reinterpret_cast<const TI*>(p.x);

  // gather bookkeeping
  int gb = 0, gy = 0, gx = 0, cpp = 1;
  if (p.gather) {
    int64_t r = rvalid ? row : 0;
    gx = (int)(r % p.W2); r /= p.W2;
    gy = (int)(r % p.H2); gb = (int)(r / p.H2);
    cpp = p.Cin / NV;  // chunks per part
  }

  float v[IT][NV];
  float s = 0.f;
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int c = sub + it * LPR;
    bool ok = rvalid && c < nchunks;
    const TI* src = nullptr;
    if (ok) {
      if (p.gather) {
        const int part = c / cpp, cc = c - part * cpp;
        const int yy = 2 * gy + (part & 1), xx = 2 * gx + (part >> 1);
        if (yy < p.H && xx < p.W)
          src = X + (((int64_t)gb * p.H + yy) * p.W + xx) * p.Cin + cc * NV;
        else
          ok = false;  // zero padding (F.pad), still counted in the statistics
      } else {
        src = X + row * p.C + c * NV;
      }
    }
    if (ok) {
      unpack16(ld16<TI>(src), v[it]);
    } else {
#pragma unroll
      for (int i = 0; i < NV; ++i) v[it][i] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) s += v[it][i];
  }
  for (int o = LPR >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mean = s / (float)p.C;
  float q = 0.f;
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int c = sub + it * LPR;
    if (c < nchunks) {
#pragma unroll
      for (int i = 0; i < NV; ++i) { float d = v[it][i] - mean; q += d * d; }
    }
  }
  for (int o = LPR >> 1; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
  const float rstd = 1.0f / sqrtf(q / (float)p.C + p.eps);
  if (!rvalid) return;
  TO* Y = reinterpret_cast<TO*>(p.y);
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int c = sub + it * LPR;
    if (c < nchunks) {
      float o[NV];
#pragma unroll
      for (int i = 0; i < NV; i += 4) {
        f32x4 gg = *reinterpret_cast<const f32x4*>(p.g + c * NV + i);
        f32x4 bb = *reinterpret_cast<const f32x4*>(p.b + c * NV + i);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[i + j] = (v[it][i + j] - mean) * rstd * gg[j] + bb[j];
      }
      if constexpr (sizeof(TO) == 2) {
        if (Y != nullptr && p.split) {
          float hi[NV], lo[NV];
#pragma unroll
          for (int i = 0; i < NV; ++i) { hi[i] = (float)(bf16_t)o[i]; lo[i] = o[i] - hi[i]; }
          store_vals<TO, NV>(Y + row * 2 * p.C + c * NV, hi);
          store_vals<TO, NV>(Y + row * 2 * p.C + p.C + c * NV, lo);
        } else if (Y != nullptr) {
          store_vals<TO, NV>(Y + row * p.C + c * NV, o);
        }
      } else {
        if (Y != nullptr) store_vals<TO, NV>(Y + row * p.C + c * NV, o);
      }
      if (p.yf != nullptr) store_vals<float, NV>(p.yf + row * p.C + c * NV, o);
    }
  }
}

template <typename TI, typename TO>
int launch_ln(LnP p, hipStream_t st, const char* name) {
  constexpr int NV = Vec16<TI>::N;
  OMP_CHECK_ARG(p.C % NV == 0 && p.C % 4 == 0, "%s: C=%d must be a multiple of %d", name, p.C, NV);
  const int nchunks = p.C / NV;
  int lg = 0;
  while ((1 << lg) < nchunks && lg < 6) ++lg;
  p.lpr_log2 = lg;
  const int LPR = 1 << lg;
  const int it = (nchunks + LPR - 1) / LPR;
  const int rpw = 64 >> lg;
  const int64_t blocks = ceil_div64(p.rows, 4 * rpw);
  dim3 grid((unsigned)blocks), block(256);
  if (it <= 1) hipLaunchKernelGGL((ln_kernel<TI, TO, 1>), grid, block, 0, st, p);
  else if (it <= 2) hipLaunchKernelGGL((ln_kernel<TI, TO, 2>), grid, block, 0, st, p);
  else if (it <= 4) hipLaunchKernelGGL((ln_kernel<TI, TO, 4>), grid, block, 0, st, p);
  else if (it <= 8) hipLaunchKernelGGL((ln_kernel<TI, TO, 8>), grid, block, 0, st, p);
  else if (it <= 16) hipLaunchKernelGGL((ln_kernel<TI, TO, 16>), grid, block, 0, st, p);
  else { omp_set_error("%s: C=%d too large", name, p.C); return OMP_ERR_UNSUPPORTED; }
  OMP_CHECK_LAUNCH(name);
  return OMP_OK;
}

int dispatch_ln(LnP p, int x_dtype, int y_dtype, hipStream_t st, const char* name) {
  p.split = 0;
  if (y_dtype == OMP_BF16X2) { p.split = 1; y_dtype = OMP_BF16; }
  if (x_dtype == OMP_F32 && y_dtype == OMP_F32) return launch_ln<float, float>(p, st, name);
  if (x_dtype == OMP_F32 && y_dtype == OMP_BF16) return launch_ln<float, bf16_t>(p, st, name);
  if (x_dtype == OMP_BF16 && y_dtype == OMP_BF16) return launch_ln<bf16_t, bf16_t>(p, st, name);
  if (x_dtype == OMP_BF16 && y_dtype == OMP_F32) return launch_ln<bf16_t, float>(p, st, name);
  omp_set_error("%s: bad dtypes %d -> %d", name, x_dtype, y_dtype);
  return OMP_ERR_INVALID;
}

// ---------------------------------------------------------------------------------------------
// PatchEmbed: one block = TOK consecutive tokens of one token-row; thread c = output channel.
// ---------------------------------------------------------------------------------------------
constexpr int PE_TOK = 16;   // tokens per workgroup: a thread's 48 weights are loaded once per workgroup

template <typename TO>
__global__ void patch_embed_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                   const float* __restrict__ bias, const float* __restrict__ g,
                                   const float* __restrict__ be, TO* __restrict__ out, int B, int H,
                                   int W, int Hp, int Wp, int E, float eps) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* patch = sm;               // [PE_TOK][48]
  float* vals = sm + PE_TOK * 48;  // [PE_TOK][E]
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int xb = blockIdx.x * PE_TOK, ty = blockIdx.y, b = blockIdx.z;
  for (int idx = tid; idx < PE_TOK * 48; idx += nthr) {
    const int t = idx / 48, e = idx - t * 48;
    const int ch = e >> 4, ky = (e >> 2) & 3, kx = e & 3;
    const int py = ty * 4 + ky, px = (xb + t) * 4 + kx;
    float v = 0.f;
    if (py < H && px < W && xb + t < Wp) v = img[(((int64_t)b * 3 + ch) * H + py) * W + px];
    patch[idx] = v;
  }
  __syncthreads();
  if (tid < E) {
    float wr[48];
#pragma unroll
    for (int i = 0; i < 48; i += 4) {
      f32x4 t4 = *reinterpret_cast<const f32x4*>(w + tid * 48 + i);
      wr[i] = t4[0]; wr[i + 1] = t4[1]; wr[i + 2] = t4[2]; wr[i + 3] = t4[3];
    }
    const float bv = bias[tid];
#pragma unroll 4
    for (int t = 0; t < PE_TOK; ++t) {
      float a = 0.f;
#pragma unroll
      for (int e = 0; e < 48; e += 4) {   // wave-wide broadcast reads, 16 bytes each (the kernel is LDS-issue bound)
        const f32x4 p4 = *reinterpret_cast<const f32x4*>(patch + t * 48 + e);
        a = fmaf(wr[e], p4[0], a);
        a = fmaf(wr[e + 1], p4[1], a);
        a = fmaf(wr[e + 2], p4[2], a);
        a = fmaf(wr[e + 3], p4[3], a);
      }
      vals[t * E + tid] = a + bv;
    }
  }
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6, nw = nthr >> 6;
  for (int t = wave; t < PE_TOK; t += nw) {
    if (xb + t >= Wp) continue;
    float s = 0.f;
    for (int c = lane; c < E; c += 64) s += vals[t * E + c];
    s = wave_sum(s);
    const float mean = s / (float)E;
    float q = 0.f;
    for (int c = lane; c < E; c += 64) { float d = vals[t * E + c] - mean; q += d * d; }
    q = wave_sum(q);
    const float rstd = 1.0f / sqrtf(q / (float)E + eps);
    TO* o = out + (((int64_t)b * Hp + ty) * Wp + xb + t) * E;
    for (int c = lane; c < E; c += 64) o[c] = from_f32<TO>((vals[t * E + c] - mean) * rstd * g[c] + be[c]);
  }
}

// fp32 rows -> split-bf16 pair rows (the operand format of the bf16x3 products): hi = bf16(x), lo = bf16(x - hi).
// triple = 0: [hi | lo] (A operands, read with a_wrap), 1: [hi | hi | lo] (the W-side image of an activation)
__global__ __launch_bounds__(256) void split_bf16_kernel(const float* __restrict__ x, int64_t ldx, bf16_t* __restrict__ y, int64_t ldy,
                                                         int64_t rows, int C, int triple) {
  const int cpr = C >> 2;   // 4-element chunks per row
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= rows * cpr) return;
  const int64_t r = idx / cpr;
  const int c = (int)(idx - r * cpr) * 4;
  const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * ldx + c);
  bf16x4 hi, lo;
#pragma unroll
  for (int i = 0; i < 4; ++i) { hi[i] = (bf16_t)v[i]; lo[i] = (bf16_t)(v[i] - (float)hi[i]); }
  bf16_t* yr = y + r * ldy + c;
  *reinterpret_cast<bf16x4*>(yr) = hi;
  if (triple) {
    *reinterpret_cast<bf16x4*>(yr + C) = hi;
    *reinterpret_cast<bf16x4*>(yr + 2 * C) = lo;
  } else {
    *reinterpret_cast<bf16x4*>(yr + C) = lo;
  }
}

// PatchEmbed, one THREAD per token (E <= 128): the 48 inputs of a token are 12 sixteen-byte loads (lane t reads the four kx
// values of token t: consecutive lanes read consecutive 16 bytes of an image row -- fully coalesced), the E x 48 weights are
// wave-uniform and arrive through the scalar cache, the LayerNorm over the token's E channels is in-thread arithmetic (no LDS,
// no cross-lane traffic, no barrier).  The workgroup-per-16-tokens kernel above staged every patch value through LDS and was
// LDS-issue bound at 0.12 of the HBM roofline (949 us per 32-image chunk, profiles/r02zl_*); kept for E > 128.
template <typename TO, int EMAX>
__global__ __launch_bounds__(256) void patch_embed_tok_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                              const float* __restrict__ bias, const float* __restrict__ g,
                                                              const float* __restrict__ be, TO* __restrict__ out, int B, int H, int W,
                                                              int Hp, int Wp, int E, float eps) {
  const int64_t tok = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t ntok = (int64_t)B * Hp * Wp;
  const bool live = tok < ntok;
  const int64_t tk = live ? tok : ntok - 1;
  const int tx = (int)(tk % Wp), ty = (int)((tk / Wp) % Hp), b = (int)(tk / ((int64_t)Wp * Hp));
  float x[48];
  const bool vec = (W & 3) == 0 && tx * 4 + 3 < W;
#pragma unroll
  for (int ch = 0; ch < 3; ++ch)
#pragma unroll
    for (int ky = 0; ky < 4; ++ky) {
      const int py = ty * 4 + ky;
      const float* row = img + (((int64_t)b * 3 + ch) * H + (py < H ? py : H - 1)) * W + tx * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (py < H) {
        if (vec) v = *reinterpret_cast<const f32x4*>(row);
        else {
#pragma unroll
          for (int kx = 0; kx < 4; ++kx)
            if (tx * 4 + kx < W) v[kx] = row[kx];
        }
      }
#pragma unroll
      for (int kx = 0; kx < 4; ++kx) x[ch * 16 + ky * 4 + kx] = v[kx];
    }
  float y[EMAX];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < EMAX; ++c) {
    float a = 0.f;
    if (c < E) {                       // wave-uniform
      const float* wc = w + c * 48;    // uniform address: scalar loads
      a = bias[c];
#pragma unroll
      for (int e = 0; e < 48; ++e) a = fmaf(wc[e], x[e], a);
    }
    y[c] = a;
    s += a;
  }
  const float mean = s / (float)E;
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < EMAX; ++c)
    if (c < E) { const float d = y[c] - mean; q += d * d; }
  const float rstd = 1.0f / sqrtf(q / (float)E + eps);
  if (!live) return;
  TO* o = out + tok * E;
  constexpr int NV = 16 / (int)sizeof(TO);
#pragma unroll
  for (int c = 0; c < EMAX; c += NV) {
    if (c < E) {
      float t[NV];
#pragma unroll
      for (int i = 0; i < NV; ++i) t[i] = (y[c + i] - mean) * rstd * g[c + i] + be[c + i];
      store_vals<TO, NV>(o + c, t);
    }
  }
}


// PatchEmbed on the fp32 matrix cores (round 4; E = 16 NB <= 128, fp32 residual stream out).  The thread-per-token kernel above spends
// 6144 scalar-operand FMAs per token and writes every token's 512-byte row from ONE lane, 16 bytes at a time (64 different rows per
// store instruction): 1.49-1.6 ms per 32-image chunk = 0.12 of the HBM roofline (profiles/r03m_*).  Here a wave owns 16 consecutive
// tokens: D[channel][token] = W[channel][k] x[token][k] over k = 48 patch values + 1 (the bias rides along as k = 48 against a
// constant 1), 13 steps of v_mfma_f32_16x16x4_f32 per 16-channel block; the weights (13 registers per block) stay in registers
// across the persistent wave's tiles; a step's B operand is ONE coalesced dword load (lane (token, kx) reads pixel kx of the
// token's patch row: 64 consecutive floats of an image row per instruction); the LayerNorm statistics are in-register sums + two
// register swaps across the four lane groups; the normalised tile goes through a wave-private LDS transpose so that every store
// instruction writes 1 KB of consecutive output (two whole token rows).
template <int NB>
__global__ __launch_bounds__(256) void patch_embed_mfma_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                               const float* __restrict__ bias, const float* __restrict__ g,
                                                               const float* __restrict__ be, float* __restrict__ out, int B, int H, int W,
                                                               int Hp, int Wp, float eps, int ntiles) {
  constexpr int E = 16 * NB, PITCH = E + 4;
  extern __shared__ __attribute__((aligned(16))) float pe_lds[];   // [4 waves][16 tokens][PITCH]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, kq = lane >> 4;
  float* T = pe_lds + wave * 16 * PITCH;
  // A operand: channel cb * 16 + li, k = 4 s + kq (s = 12: the bias against x = 1 in lane group 0)
  float wr[NB][13];
#pragma unroll
  for (int cb = 0; cb < NB; ++cb) {
    const float* wc = w + (cb * 16 + li) * 48 + kq;
#pragma unroll
    for (int s_ = 0; s_ < 12; ++s_) wr[cb][s_] = wc[4 * s_];
    wr[cb][12] = kq == 0 ? bias[cb * 16 + li] : 0.f;
  }
  const int64_t ntok = (int64_t)B * Hp * Wp;
  const float one = kq == 0 ? 1.f : 0.f;
  const int nwaves = gridDim.x * 4;
  for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += nwaves) {
    unsigned tok = (unsigned)tile * 16u + (unsigned)li;   // the host sends token counts >= 2^31 to the scalar kernel
    if (tok > (unsigned)(ntok - 1)) tok = (unsigned)(ntok - 1);
    const unsigned trow = tok / (unsigned)Wp;
    const int tx = (int)(tok - trow * (unsigned)Wp), b = (int)(trow / (unsigned)Hp), ty = (int)(trow - (unsigned)b * (unsigned)Hp);
    const int px = tx * 4 + kq;
    float x[12];
#pragma unroll
    for (int s_ = 0; s_ < 12; ++s_) {
      const int ch = s_ >> 2, py = ty * 4 + (s_ & 3);
      const bool in = py < H && px < W;
      x[s_] = in ? img[(((int64_t)b * 3 + ch) * H + py) * W + px] : 0.f;
    }
    f32x4 acc[NB];
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) {
      acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s_ = 0; s_ < 12; ++s_) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[cb][s_], x[s_], acc[cb], 0, 0, 0);
      acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[cb][12], one, acc[cb], 0, 0, 0);
    }
    // LayerNorm statistics of token li: this lane holds channels cb * 16 + 4 kq + r
    float sm = 0.f;
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) sm += (acc[cb][0] + acc[cb][1]) + (acc[cb][2] + acc[cb][3]);
    const float mean = quad_group_sum(sm) * (1.0f / (float)E);
    float q = 0.f;
#pragma unroll
    for (int cb = 0; cb < NB; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float d = acc[cb][r] - mean; q += d * d; }
    const float rstd = 1.0f / sqrtf(quad_group_sum(q) * (1.0f / (float)E) + eps);
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) {
      f32x4 n;
#pragma unroll
      for (int r = 0; r < 4; ++r) n[r] = (acc[cb][r] - mean) * rstd;
      *reinterpret_cast<f32x4*>(T + li * PITCH + cb * 16 + kq * 4) = n;
    }
    // wave-private tile: LDS operations of one wave complete in order.  Read back row-contiguous, affine, 16-byte stores
    constexpr int CPR = E / 4;   // 16-byte chunks per token row
#pragma unroll
    for (int it = 0; it < 16 * CPR / 64; ++it) {
      const int idx = it * 64 + lane, t = idx / CPR, c4 = (idx - t * CPR) * 4;
      const f32x4 v = *reinterpret_cast<const f32x4*>(T + t * PITCH + c4);
      const f32x4 gg = *reinterpret_cast<const f32x4*>(g + c4), bb = *reinterpret_cast<const f32x4*>(be + c4);
      const int64_t to = (int64_t)tile * 16 + t;
      if (to < ntok) *reinterpret_cast<f32x4*>(out + to * E + c4) = f32x4{v[0] * gg[0] + bb[0], v[1] * gg[1] + bb[1], v[2] * gg[2] + bb[2], v[3] * gg[3] + bb[3]};
    }
  }
}

}  // namespace

extern "C" int omp_split_bf16(const float* x, int64_t ldx, void* y, int64_t ldy, int64_t rows, int C, int triple, omp_stream_t s) {
  OMP_CHECK_ARG(x && y, "omp_split_bf16: null pointer");
  OMP_CHECK_ARG(rows > 0 && C > 0 && C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && ldx >= C && ldy >= (triple ? 3 : 2) * (int64_t)C,
                "omp_split_bf16: bad shape rows=%lld C=%d", (long long)rows, C);
  OMP_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 8) == 0, "omp_split_bf16: misaligned pointers");
  const int64_t n = rows * (C / 4);
  hipLaunchKernelGGL(split_bf16_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, (hipStream_t)s, x, ldx, (bf16_t*)y, ldy, rows, C,
                     triple ? 1 : 0);
  OMP_CHECK_LAUNCH("omp_split_bf16");
  return OMP_OK;
}

extern "C" int omp_layernorm(const void* x, int x_dtype, const float* gamma, const float* beta, void* y,
                             int y_dtype, float* y_f32, int64_t rows, int C, float eps, omp_stream_t s) {
  OMP_CHECK_ARG(x && gamma && beta && (y || y_f32), "omp_layernorm: null pointer");
  OMP_CHECK_ARG(rows > 0 && C > 0, "omp_layernorm: bad shape rows=%lld C=%d", (long long)rows, C);
  LnP p{};
  p.x = x; p.g = gamma; p.b = beta; p.y = y; p.yf = y_f32; p.rows = rows; p.C = C; p.eps = eps;
  p.gather = 0;
  if (y == nullptr) y_dtype = OMP_F32;
  return dispatch_ln(p, x_dtype, y_dtype, (hipStream_t)s, "omp_layernorm");
}

extern "C" int omp_patch_merge_gather_ln(const void* x, const float* gamma, const float* beta, void* y,
                                         int dtype, int B, int H, int W, int C, float eps,
                                         omp_stream_t s) {
  return omp_patch_merge_gather_ln2(x, dtype, gamma, beta, y, dtype, B, H, W, C, eps, s);
}

extern "C" int omp_patch_merge_gather_ln2(const void* x, int x_dtype, const float* gamma, const float* beta, void* y,
                                          int y_dtype, int B, int H, int W, int C, float eps, omp_stream_t s) {
  const int dtype = x_dtype;
  OMP_CHECK_ARG(x && gamma && beta && y, "omp_patch_merge_gather_ln: null pointer");
  OMP_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0, "omp_patch_merge_gather_ln: bad shape");
  LnP p{};
  p.x = x; p.g = gamma; p.b = beta; p.y = y; p.yf = nullptr; p.eps = eps;
  p.gather = 1; p.H = H; p.W = W; p.Cin = C; p.H2 = (H + 1) / 2; p.W2 = (W + 1) / 2;
  p.rows = (int64_t)B * p.H2 * p.W2; p.C = 4 * C;
  const int nv = dtype == OMP_F32 ? 4 : 8;
  OMP_CHECK_ARG(C % nv == 0, "omp_patch_merge_gather_ln: C=%d must be a multiple of %d", C, nv);
  return dispatch_ln(p, x_dtype, y_dtype, (hipStream_t)s, "omp_patch_merge_gather_ln");
}

extern "C" int omp_patch_embed_ln(const float* img, const float* w, const float* b, const float* gamma,
                                  const float* beta, void* out, int out_dtype, int B, int H, int W, int E,
                                  float eps, omp_stream_t s) {
  OMP_CHECK_ARG(img && w && b && gamma && beta && out, "omp_patch_embed_ln: null pointer");
  OMP_CHECK_ARG(B > 0 && H > 0 && W > 0 && E > 0 && E <= 1024, "omp_patch_embed_ln: bad shape");
  const int Hp = (H + 3) / 4, Wp = (W + 3) / 4;
  if ((E == 128 || E == 96) && out_dtype == OMP_F32 && (int64_t)B * Hp * Wp < (1ll << 31) - 16 && ((uintptr_t)out % 16) == 0 && ((uintptr_t)gamma % 16) == 0 && ((uintptr_t)beta % 16) == 0 &&
      getenv("OMP355_PATCH_EMBED_SCALAR") == nullptr) {
    // fp32 matrix cores, 16 tokens per wave, persistent waves (csrc/norm.hip: patch_embed_mfma_kernel)
    const int64_t ntok = (int64_t)B * Hp * Wp;
    const int ntiles = (int)ceil_div64(ntok, 16);
    int nwg = (int)ceil_div64(ntiles, 4);
    if (nwg > 512) nwg = 512;   // two resident workgroups per CU (172 registers): the waves are persistent, the weights are read once each
    if (E == 128)
      hipLaunchKernelGGL((patch_embed_mfma_kernel<8>), dim3(nwg), dim3(256), 4 * 16 * 132 * sizeof(float), (hipStream_t)s, img, w, b, gamma, beta, (float*)out, B,
                         H, W, Hp, Wp, eps, ntiles);
    else
      hipLaunchKernelGGL((patch_embed_mfma_kernel<6>), dim3(nwg), dim3(256), 4 * 16 * 100 * sizeof(float), (hipStream_t)s, img, w, b, gamma, beta, (float*)out, B,
                         H, W, Hp, Wp, eps, ntiles);
    OMP_CHECK_LAUNCH("omp_patch_embed_ln");
    return OMP_OK;
  }
  if (E <= 128 && E % 8 == 0 && (out_dtype == OMP_F32 || out_dtype == OMP_BF16)) {   // thread-per-token kernel
    const int64_t ntok = (int64_t)B * Hp * Wp;
    const dim3 tgrid((unsigned)ceil_div64(ntok, 256));
    if (out_dtype == OMP_F32)
      hipLaunchKernelGGL((patch_embed_tok_kernel<float, 128>), tgrid, dim3(256), 0, (hipStream_t)s, img, w, b, gamma, beta, (float*)out, B, H,
                         W, Hp, Wp, E, eps);
    else
      hipLaunchKernelGGL((patch_embed_tok_kernel<bf16_t, 128>), tgrid, dim3(256), 0, (hipStream_t)s, img, w, b, gamma, beta, (bf16_t*)out, B,
                         H, W, Hp, Wp, E, eps);
    OMP_CHECK_LAUNCH("omp_patch_embed_ln");
    return OMP_OK;
  }
  const int nthr = ((E + 63) / 64) * 64;
  dim3 grid((Wp + PE_TOK - 1) / PE_TOK, Hp, B);
  const size_t smem = (PE_TOK * 48 + PE_TOK * E) * sizeof(float);
  OMP_CHECK_ARG(smem <= 64 * 1024, "omp_patch_embed_ln: E=%d needs %zu bytes of LDS (> 64 KB)", E, smem);   // E <= 976
  if (out_dtype == OMP_F32)
    hipLaunchKernelGGL((patch_embed_kernel<float>), grid, dim3(nthr), smem, (hipStream_t)s, img, w, b,
                       gamma, beta, (float*)out, B, H, W, Hp, Wp, E, eps);
  else if (out_dtype == OMP_BF16)
    hipLaunchKernelGGL((patch_embed_kernel<bf16_t>), grid, dim3(nthr), smem, (hipStream_t)s, img, w, b,
                       gamma, beta, (bf16_t*)out, B, H, W, Hp, Wp, E, eps);
  else { omp_set_error("omp_patch_embed_ln: bad dtype %d", out_dtype); return OMP_ERR_INVALID; }
  OMP_CHECK_LAUNCH("omp_patch_embed_ln");
  return OMP_OK;
}
