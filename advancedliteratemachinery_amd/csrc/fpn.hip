// FPN top-down/resample/concat fused with the stride-2 sampling of input_proj, and the sine
// position embedding.  Both are pure gather/elementwise (HBM-bound) kernels on token-major maps.
#include "common.h"

namespace {

struct FpnP {
  const void* l[4];  // lateral maps p2..p5 sources: l2,l3,l4,l5 (token-major, 256 ch)
  int h[4], w[4];
  void* out;
  int B, ho, wo, stride;
};

constexpr int FC = 256;  // fpn channels per level

// PyTorch 'nearest': src = min(floor(dst * (float)in/out), in-1)
__device__ __forceinline__ int nearest_idx(int dst, int in_sz, int out_sz) {
  const float scale = (float)in_sz / (float)out_sz;
  int s = (int)floorf((float)dst * scale);
  return s < in_sz - 1 ? s : in_sz - 1;
}
// PyTorch bilinear, align_corners=False
__device__ __forceinline__ void bilinear_idx(int dst, int in_sz, int out_sz, int& i0, int& i1, float& lam) {
  const float scale = (float)in_sz / (float)out_sz;
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  i0 = (int)src;
  if (i0 > in_sz - 1) i0 = in_sz - 1;
  i1 = i0 + 1 < in_sz ? i0 + 1 : in_sz - 1;
  lam = src - (float)i0;
  lam = fminf(fmaxf(lam, 0.f), 1.f);
}

template <typename T, int NV>
__device__ __forceinline__ void load_add(const FpnP& p, int lvl, int b, int y, int x, int c0, float* acc) {
  const T* base = reinterpret_cast<const T*>(p.l[lvl]) + (((int64_t)b * p.h[lvl] + y) * p.w[lvl] + x) * FC + c0;
  float t[NV];
  unpack16(ld16<T>(base), t);
#pragma unroll
  for (int i = 0; i < NV; ++i) acc[i] += t[i];
}

// value of the top-down pyramid level `lvl` (0 = p2 .. 3 = p5) at (y, x): lateral + nearest-upsampled
// coarser levels (fpn.py:25-36).  Rounded to T after every add like the reference's tensors are.
template <typename T, int NV>
__device__ __forceinline__ void pyramid_at(const FpnP& p, int lvl, int b, int y, int x, int c0, float* v) {
  // gather coordinates down to p5
  int ys[4], xs[4];
  ys[lvl] = y; xs[lvl] = x;
  for (int k = lvl + 1; k < 4; ++k) {
    ys[k] = nearest_idx(ys[k - 1], p.h[k], p.h[k - 1]);
    xs[k] = nearest_idx(xs[k - 1], p.w[k], p.w[k - 1]);
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = 0.f;
  // coarse to fine so that the add order matches p_k = conv(c_k) + up(p_{k+1})
  for (int k = 3; k >= lvl; --k) {
    float t[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) t[i] = 0.f;
    load_add<T, NV>(p, k, b, ys[k], xs[k], c0, t);
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = to_f32(from_f32<T>(t[i] + v[i]));
  }
}

template <typename T>
__global__ __launch_bounds__(256) void fpn_fuse_kernel(FpnP p) {
  constexpr int NV = Vec16<T>::N;
  constexpr int CPL = FC / NV;
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t ntok = (int64_t)p.B * p.ho * p.wo;
  const int64_t tok = g / (4 * CPL);
  if (tok >= ntok) return;
  const int rem = (int)(g % (4 * CPL));
  const int lvl = rem / CPL, c0 = (rem % CPL) * NV;
  int t = (int)(tok % ((int64_t)p.ho * p.wo));
  const int b = (int)(tok / ((int64_t)p.ho * p.wo));
  const int y3 = (t / p.wo) * p.stride, x3 = (t % p.wo) * p.stride;
  float o[NV];
  if (lvl == 1) {
    pyramid_at<T, NV>(p, 1, b, y3, x3, c0, o);
  } else {
    int y0, y1, x0, x1; float ly, lx;
    bilinear_idx(y3, p.h[lvl], p.h[1], y0, y1, ly);
    bilinear_idx(x3, p.w[lvl], p.w[1], x0, x1, lx);
    float v00[NV], v01[NV], v10[NV], v11[NV];
    pyramid_at<T, NV>(p, lvl, b, y0, x0, c0, v00);
    pyramid_at<T, NV>(p, lvl, b, y0, x1, c0, v01);
    pyramid_at<T, NV>(p, lvl, b, y1, x0, c0, v10);
    pyramid_at<T, NV>(p, lvl, b, y1, x1, c0, v11);
    const float hy = 1.f - ly, hx = 1.f - lx;
#pragma unroll
    for (int i = 0; i < NV; ++i) o[i] = hy * (hx * v00[i] + lx * v01[i]) + ly * (hx * v10[i] + lx * v11[i]);
  }
  typename Vec16<T>::type pv;
  pack16(o, pv);
  st16<T>(reinterpret_cast<T*>(p.out) + tok * (4 * FC) + lvl * FC + c0, pv);
}

template <typename T>
__global__ __launch_bounds__(256) void sine_pos_kernel(const uint8_t* __restrict__ mask, T* __restrict__ pos,
                                                       int B, int h, int w, int npf, float temperature) {
  __shared__ float e[2];
  const int tok = blockIdx.x;  // (b, y, x)
  const int x = tok % w, y = (tok / w) % h, b = tok / (w * h);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint8_t* mb = mask + (int64_t)b * h * w;
  const float two_pi = 6.283185307179586f;
  if (wave == 0) {
    float cum = 0.f, tot = 0.f;
    for (int yy = lane; yy < h; yy += 64) {
      const float nm = mb[yy * w + x] ? 0.f : 1.f;
      tot += nm;
      if (yy <= y) cum += nm;
    }
    cum = wave_sum(cum); tot = wave_sum(tot);
    if (lane == 0) e[0] = cum / (tot + 1e-6f) * two_pi;
  } else if (wave == 1) {
    float cum = 0.f, tot = 0.f;
    for (int xx = lane; xx < w; xx += 64) {
      const float nm = mb[y * w + xx] ? 0.f : 1.f;
      tot += nm;
      if (xx <= x) cum += nm;
    }
    cum = wave_sum(cum); tot = wave_sum(tot);
    if (lane == 0) e[1] = cum / (tot + 1e-6f) * two_pi;
  }
  __syncthreads();
  T* o = pos + (int64_t)tok * 2 * npf;
  for (int c = threadIdx.x; c < 2 * npf; c += blockDim.x) {
    const int i = c < npf ? c : c - npf;
    const float emb = c < npf ? e[0] : e[1];
    const float dim_t = powf(temperature, (float)(2 * (i / 2)) / (float)npf);
    const float v = emb / dim_t;
    o[c] = from_f32<T>((i & 1) ? cosf(v) : sinf(v));
  }
}

__global__ __launch_bounds__(256) void mask_nearest_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int B, int H,
                                                           int W, int h, int w) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)B * h * w) return;
  const int x = (int)(idx % w), y = (int)((idx / w) % h), b = (int)(idx / ((int64_t)w * h));
  const float sy = (float)H / (float)h, sx = (float)W / (float)w;   // ATen: compute_scales_value<float>
  int yy = (int)floorf((float)y * sy); if (yy > H - 1) yy = H - 1;
  int xx = (int)floorf((float)x * sx); if (xx > W - 1) xx = W - 1;
  out[idx] = in[((int64_t)b * H + yy) * W + xx] ? 1 : 0;
}

}  // namespace

extern "C" int omp_mask_nearest(const uint8_t* in, uint8_t* out, int B, int H, int W, int h, int w, omp_stream_t s) {
  OMP_CHECK_ARG(in && out, "omp_mask_nearest: null pointer");
  OMP_CHECK_ARG(B > 0 && H > 0 && W > 0 && h > 0 && w > 0, "omp_mask_nearest: bad shape");
  const int64_t total = (int64_t)B * h * w;
  hipLaunchKernelGGL(mask_nearest_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, (hipStream_t)s, in, out, B, H, W, h, w);
  OMP_CHECK_LAUNCH("omp_mask_nearest");
  return OMP_OK;
}

extern "C" int omp_fpn_fuse(const void* l2, const void* l3, const void* l4, const void* l5, void* out,
                            int dtype, int B, int h2, int w2, int h3, int w3, int h4, int w4, int h5,
                            int w5, int stride, omp_stream_t s) {
  OMP_CHECK_ARG(l2 && l3 && l4 && l5 && out, "omp_fpn_fuse: null pointer");
  OMP_CHECK_ARG(stride >= 1 && B > 0, "omp_fpn_fuse: bad stride/B");
  FpnP p;
  p.l[0] = l2; p.l[1] = l3; p.l[2] = l4; p.l[3] = l5;
  p.h[0] = h2; p.h[1] = h3; p.h[2] = h4; p.h[3] = h5;
  p.w[0] = w2; p.w[1] = w3; p.w[2] = w4; p.w[3] = w5;
  p.out = out; p.B = B; p.stride = stride;
  p.ho = (h3 + stride - 1) / stride; p.wo = (w3 + stride - 1) / stride;
  const int nv = dtype == OMP_F32 ? 4 : 8;
  const int64_t total = (int64_t)B * p.ho * p.wo * 4 * (FC / nv);
  dim3 grid((unsigned)ceil_div64(total, 256));
  if (dtype == OMP_F32) hipLaunchKernelGGL((fpn_fuse_kernel<float>), grid, dim3(256), 0, (hipStream_t)s, p);
  else if (dtype == OMP_BF16) hipLaunchKernelGGL((fpn_fuse_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)s, p);
  else { omp_set_error("omp_fpn_fuse: bad dtype %d", dtype); return OMP_ERR_INVALID; }
  OMP_CHECK_LAUNCH("omp_fpn_fuse");
  return OMP_OK;
}

extern "C" int omp_sine_posembed(const uint8_t* mask, void* pos, int dtype, int B, int h, int w, int npf,
                                 float temperature, omp_stream_t s) {
  OMP_CHECK_ARG(mask && pos, "omp_sine_posembed: null pointer");
  OMP_CHECK_ARG(B > 0 && h > 0 && w > 0 && npf > 0, "omp_sine_posembed: bad shape");
  dim3 grid((unsigned)((int64_t)B * h * w));
  if (dtype == OMP_F32)
    hipLaunchKernelGGL((sine_pos_kernel<float>), grid, dim3(256), 0, (hipStream_t)s, mask, (float*)pos, B, h, w, npf, temperature);
  else if (dtype == OMP_BF16)
    hipLaunchKernelGGL((sine_pos_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)s, mask, (bf16_t*)pos, B, h, w, npf, temperature);
  else { omp_set_error("omp_sine_posembed: bad dtype %d", dtype); return OMP_ERR_INVALID; }
  OMP_CHECK_LAUNCH("omp_sine_posembed");
  return OMP_OK;
}
