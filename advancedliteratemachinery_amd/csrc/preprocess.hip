// Test-time image pre-processing on the device: the step right before the hot path.
//
// Reference (OCR/OmniParser): dataset/transforms.py:249-298 (aspect-preserving resize = torchvision F.resize of a
// PIL image = Pillow's bilinear ImagingResample), :312-322 (ToTensor, Normalize), utils/nested_tensor.py:37-54
// (zero-pad to the batch maximum, mask True on padding).  One launch per image writes that image's slice of
// the batch tensor: resized + normalised pixels inside (oh, ow), zeros and mask = 1 outside.
//
// Bit-exactness with Pillow's 8-bit resampler is part of the contract (the oracle is pinned to PIL itself):
//   * the coefficient tables (triangle filter widened by the down-scale factor, normalised in double, quantised
//     to 22 fractional bits) are computed on the host exactly as Resample.c does and passed in;
//   * horizontal pass first, its result ROUNDED TO uint8 (Pillow materialises a uint8 temporary), then the
//     vertical pass; a thread recomputes the few horizontally-filtered pixels of its column footprint instead of
//     round-tripping a temporary image through HBM;
//   * ToTensor/Normalize are a 3 x 256 float table built on the host with the reference's own float32 operations
//     ((p / 255 - mean) / std), so no device rounding mode enters the result.
#include "common.h"

namespace {

struct PreP {
  const uint8_t* src; int64_t src_pitch;   // HWC uint8, bytes per row
  int in_h, in_w;
  const int32_t* xb; const int32_t* kx; int ksx;   // [ow][2] (first, count), [ow][ksx]
  const int32_t* yb; const int32_t* ky; int ksy;   // [oh][2], [oh][ksy]
  int need_h, need_v;
  const float* lut;                        // [3][256]
  float* dst; int64_t plane;               // this image's [3][Hd][Wd] slice, plane = Hd * Wd
  uint8_t* mask;                           // [Hd][Wd] or nullptr
  int out_h, out_w, Hd, Wd;
};

constexpr int PRE_BITS = 32 - 8 - 2;

__device__ __forceinline__ int clip8(int acc) {
  const int v = acc >> PRE_BITS;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

__global__ __launch_bounds__(256) void resize_norm_pad_kernel(PreP p) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= p.Wd || y >= p.Hd) return;
  const int64_t o = (int64_t)y * p.Wd + x;
  if (x >= p.out_w || y >= p.out_h) {
    p.dst[o] = 0.f; p.dst[p.plane + o] = 0.f; p.dst[2 * p.plane + o] = 0.f;
    if (p.mask != nullptr) p.mask[o] = 1;
    return;
  }
  const int x0 = p.need_h ? p.xb[2 * x] : x, nx = p.need_h ? p.xb[2 * x + 1] : 1;
  const int y0 = p.need_v ? p.yb[2 * y] : y, ny = p.need_v ? p.yb[2 * y + 1] : 1;
  const int32_t* kx = p.kx + (int64_t)x * p.ksx;
  const int32_t* ky = p.ky + (int64_t)y * p.ksy;
  const int half = 1 << (PRE_BITS - 1);
  int v0 = half, v1 = half, v2 = half;
  int h0 = 0, h1 = 0, h2 = 0;
  for (int r = 0; r < ny; ++r) {
    const uint8_t* row = p.src + (int64_t)(y0 + r) * p.src_pitch + (int64_t)x0 * 3;
    if (p.need_h) {
      int a0 = half, a1 = half, a2 = half;
      for (int t = 0; t < nx; ++t) {
        const int k = kx[t];
        a0 += (int)row[3 * t] * k; a1 += (int)row[3 * t + 1] * k; a2 += (int)row[3 * t + 2] * k;
      }
      h0 = clip8(a0); h1 = clip8(a1); h2 = clip8(a2);
    } else {
      h0 = row[0]; h1 = row[1]; h2 = row[2];
    }
    if (p.need_v) {
      const int k = ky[r];
      v0 += h0 * k; v1 += h1 * k; v2 += h2 * k;
    }
  }
  if (p.need_v) { h0 = clip8(v0); h1 = clip8(v1); h2 = clip8(v2); }
  p.dst[o] = p.lut[h0];
  p.dst[p.plane + o] = p.lut[256 + h1];
  p.dst[2 * p.plane + o] = p.lut[512 + h2];
  if (p.mask != nullptr) p.mask[o] = 0;
}

}  // namespace

extern "C" int omp_resize_normalize_pad(const uint8_t* src, int64_t src_pitch, int in_h, int in_w, const int32_t* xbounds,
                                        const int32_t* xcoef, int ksx, const int32_t* ybounds, const int32_t* ycoef,
                                        int ksy, const float* lut, float* dst, uint8_t* mask, int out_h, int out_w,
                                        int dst_h, int dst_w, omp_stream_t s) {
  OMP_CHECK_ARG(src && lut && dst, "omp_resize_normalize_pad: null pointer");
  OMP_CHECK_ARG(in_h > 0 && in_w > 0 && out_h > 0 && out_w > 0 && dst_h >= out_h && dst_w >= out_w,
                "omp_resize_normalize_pad: bad sizes in %dx%d out %dx%d dst %dx%d", in_h, in_w, out_h, out_w, dst_h, dst_w);
  OMP_CHECK_ARG(src_pitch >= (int64_t)in_w * 3, "omp_resize_normalize_pad: src_pitch %lld < 3 * width", (long long)src_pitch);
  PreP p;
  p.src = src; p.src_pitch = src_pitch; p.in_h = in_h; p.in_w = in_w;
  p.need_h = out_w != in_w; p.need_v = out_h != in_h;
  OMP_CHECK_ARG(!p.need_h || (xbounds && xcoef && ksx > 0), "omp_resize_normalize_pad: width changes but no x coefficients");
  OMP_CHECK_ARG(!p.need_v || (ybounds && ycoef && ksy > 0), "omp_resize_normalize_pad: height changes but no y coefficients");
  p.xb = xbounds; p.kx = xcoef; p.ksx = ksx; p.yb = ybounds; p.ky = ycoef; p.ksy = ksy;
  p.lut = lut; p.dst = dst; p.plane = (int64_t)dst_h * dst_w; p.mask = mask;
  p.out_h = out_h; p.out_w = out_w; p.Hd = dst_h; p.Wd = dst_w;
  dim3 grid((dst_w + 63) / 64, (dst_h + 3) / 4);
  hipLaunchKernelGGL(resize_norm_pad_kernel, grid, dim3(256), 0, (hipStream_t)s, p);
  OMP_CHECK_LAUNCH("omp_resize_normalize_pad");
  return OMP_OK;
}
