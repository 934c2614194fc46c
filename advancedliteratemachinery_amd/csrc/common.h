// Shared device/host helpers for libomp355 (gfx950 / CDNA4 only).
//
// Conventions used by every kernel in this directory:
//   * wave = 64 lanes, hard-coded.
//   * activations are token-major (rows = tokens, channels contiguous), dtype T in {float, bf16};
//     biases / LayerNorm affine / tables are always fp32; accumulation is always fp32.
//   * a "fragment" is the 16 bytes one lane feeds to the matrix core per k-step:
//       bf16 : 8 elements -> one v_mfma_f32_16x16x32_bf16
//       f32  : 4 elements -> four v_mfma_f32_16x16x4_f32 (element j of every lane forms k-slice j)
//     In both cases lane l supplies row (l & 15) and the k-chunk (l >> 4); the k permutation is
//     the same for both operands, so the dot product is unaffected.
//   * D layout of the 16x16 MFMA: acc[r] <-> (i = (l >> 4) * 4 + r, j = l & 15).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/omp355.h"
#include "omp355_debug.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define OMP_WAVE 64

// ---------------------------------------------------------------------------------------------
// host-side error plumbing
// ---------------------------------------------------------------------------------------------
void omp_set_error(const char* fmt, ...);
#define OMP_CHECK_ARG(cond, ...)        \
  do {                                  \
    if (!(cond)) {                      \
      omp_set_error(__VA_ARGS__);       \
      return OMP_ERR_INVALID;           \
    }                                   \
  } while (0)
#define OMP_CHECK_LAUNCH(name)                                                   \
  do {                                                                           \
    hipError_t e__ = hipGetLastError();                                          \
    if (e__ != hipSuccess) {                                                     \
      omp_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));      \
      return OMP_ERR_LAUNCH;                                                     \
    }                                                                            \
  } while (0)

// measurement hooks (api.hip): hipEvent brackets around the eagerly launched kernels of one class
// GEMM = encoder-sized products (M >= 32768 rows: the Swin / FPN / projection GEMMs of a chunk, as in rounds 1-3); GEMM_DEC = the same kernels
// on the decoder phases' rows (round 4: the 10240-row polygon / recognition products moved from the unbracketed 64x64 kernel to these)
// ROWS = the row-owner chains of the decoders' many-row phases (round 5, csrc/dec_rows.hip)
enum { OMP_PROF_CROSS = 0, OMP_PROF_GEMM = 1, OMP_PROF_MLP = 2, OMP_PROF_GEMM_DEC = 3, OMP_PROF_ROWS = 4, OMP_PROF_NCLASS = 5 };

// ---------------------------------------------------------------------------------------------
// omp_ctx: ALL mutable library state (kernel selectors, development trace buffers, the table of captured decoder-step
// hipGraphs, the measurement brackets).  A host thread works on its current context (omp_ctx_make_current; the process
// default context until then); entry points read it once per call.  Nothing else in the library is writable at run
// time except the thread-local error string and the thread-local "capturing" flag of omp_decoder_run.
// ---------------------------------------------------------------------------------------------
constexpr int OMP_MAX_GRAPH_SLOTS = 4096;   // fixed table (no reallocation): pipeline lanes drive their own slots from their own threads
constexpr int OMP_GRAPH_RUN = 8;   // sampling steps per multi-step graph (omp_decoder_run)
struct OmpGraphSlot {
  hipGraph_t graph = nullptr;       // ONE sampling step
  hipGraphExec_t exec = nullptr;
  hipGraph_t graph_n = nullptr;     // OMP_GRAPH_RUN consecutive sampling steps (positions come from the device counter: the steps are identical)
  hipGraphExec_t exec_n = nullptr;
};
struct OmpProfClass;   // api.hip
struct omp_ctx {
  // kernel selectors (omp_debug_*): 0 = the measured default everywhere
  int force_gemm = 0;        // 0 auto, 3 rows, 4 small split-K, 5 dma 128x128, 6 dma 64x64 ring, 9 = 256x256 phase-interleaved, 15 = 5 + timestamps
  int cross_q4 = 1;          // 1 = LDS-ring kernel for 33..64 rows per image, 64-key chunks, non-temporal DMA; 2 = one block per step; 4 = chunks, temporal; 0 = register streaming
  int cross_nt = 1;          // non-temporal K / V^T loads: 1 = always (the slabs are read once per launch and would evict the decoder weights from the
                             // MALL: 8-image point phase 39.8 -> 37.2 ms, profiles/r03f_ab_*), 2 = only from 32 groups per launch (round 2), 0 = never
  int self_attn_impl = 0;    // 0 auto, 1 one wave per (row, head), 2 one wave per row
  int dec_fused = 0;         // 0 = fused few-row decoder kernels where they apply (csrc/decoder.hip: fused_step_ok), 1 = the launch-per-op path everywhere
  int swin_impl = 0;         // 0 matrix cores, 1 scalar cross-check kernel, 2 matrix cores with per-score table lookups
  int mlp_variant = 0;       // alternative instantiations of the fused MLP; 100 = traced default
  int gemm_choice_only = 0;  // omp_debug_gemm_choice: launch_gemm records the selector it would take in gemm_last_choice and launches nothing
  int gemm_last_choice = 0;
  unsigned long long* gemm_trace = nullptr;   // device buffers of the TRACE instantiations
  long long gemm_trace_cap = 0;
  unsigned long long* mlp_trace = nullptr;
  int rows_rtt = 0;          // decoder chains: 16-row tiles per workgroup, 0 = by row count (csrc/dec_rows.hip rows_rtt), 2..5 forced (omp_debug_rows_tile)
  OmpGraphSlot slots[OMP_MAX_GRAPH_SLOTS];
  int prof_mask = 0;
  OmpProfClass* prof = nullptr;   // [OMP_PROF_NCLASS], owned by the context (api.hip)
};
omp_ctx& omp_cur();   // the calling thread's current context
int omp_device_cus(); // compute units of the current device (cached per device; 256 when no device answers): api.hip

bool omp_prof_active(int cls);
int omp_prof_begin(int cls, hipStream_t st, double work, double bytes = 0.0);   // -> slot; bytes = algorithmic HBM bytes of the launch
void omp_prof_end(int cls, int slot, hipStream_t st);

// ---------------------------------------------------------------------------------------------
// scalar conversions
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }
template <typename T>
__device__ __forceinline__ T from_f32(float v);
template <>
__device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <>
__device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }  // RNE

// 16-byte vector of T
template <typename T>
struct Vec16;
template <>
struct Vec16<float> {
  typedef f32x4 type;
  static constexpr int N = 4;
};
template <>
struct Vec16<bf16_t> {
  typedef bf16x8 type;
  static constexpr int N = 8;
};

template <typename T>
__device__ __forceinline__ typename Vec16<T>::type ld16(const T* p) {
  return *reinterpret_cast<const typename Vec16<T>::type*>(p);
}
template <typename T>
__device__ __forceinline__ void st16(T* p, typename Vec16<T>::type v) {
  *reinterpret_cast<typename Vec16<T>::type*>(p) = v;
}

// unpack a 16-byte vector into floats
__device__ __forceinline__ void unpack16(f32x4 v, float* o) {
  o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
}
__device__ __forceinline__ void unpack16(bf16x8 v, float* o) {
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = (float)v[i];
}
__device__ __forceinline__ void pack16(const float* o, f32x4& v) {
  v[0] = o[0]; v[1] = o[1]; v[2] = o[2]; v[3] = o[3];
}
__device__ __forceinline__ void pack16(const float* o, bf16x8& v) {
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (bf16_t)o[i];
}

// ---------------------------------------------------------------------------------------------
// matrix-core fragment traits
// ---------------------------------------------------------------------------------------------
template <typename T>
struct Mma;
template <>
struct Mma<bf16_t> {
  typedef bf16x8 frag;
  static constexpr int KPL = 8;    // k elements per lane per step
  static constexpr int KSTEP = 32; // k elements per wave per step
  __device__ static __forceinline__ void mma(f32x4& acc, frag a, frag b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
  }
};
template <>
struct Mma<float> {
  typedef f32x4 frag;
  static constexpr int KPL = 4;
  static constexpr int KSTEP = 16;
  __device__ static __forceinline__ void mma(f32x4& acc, frag a, frag b) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc, 0, 0, 0);
  }
};

// ---------------------------------------------------------------------------------------------
// wave-level reductions (64 lanes)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// reduce over groups of `W` consecutive lanes (W power of two <= 64)
template <int W>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = W / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// max / sum over the 4 lanes {l, l^16, l^32, l^48} (the lane groups of a 16x16 matrix-core tile hold different
// contraction slices of the same output column): two register swaps instead of two ds_bpermute round trips.
// v_permlane32_swap exchanges the upper half of its first operand with the lower half of the second,
// v_permlane16_swap the odd 16-lane rows of the first with the even rows of the second; fed the same value twice,
// the two results are the value and its partner's.
// (inline asm, with the two wait states the swap needs after a VALU write of its operands inside the string: fed the
// same SSA value twice, hipcc / ROCm 7.2 folds the builtin's two results into one -- it emitted max(r0, r0).)
__device__ __forceinline__ void lane_swap32(float& a, float& b) {
  asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void lane_swap16(float& a, float& b) {
  asm volatile("v_nop\n\tv_nop\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float quad_group_max(float v) {
  float a = v, b = v;
  lane_swap32(a, b);
  a = fmaxf(a, b);
  b = a;
  lane_swap16(a, b);
  return fmaxf(a, b);
}
__device__ __forceinline__ float quad_group_sum(float v) {
  float a = v, b = v;
  lane_swap32(a, b);
  a = a + b;
  b = a;
  lane_swap16(a, b);
  return a + b;
}

// erf(a), branch-free: both ranges are evaluated and selected, so a wave never diverges (the library erff
// takes two divergent paths and costs several hundred cycles per element in a GEMM epilogue).  Minimax
// polynomials for |a| <= 0.9277 and 1 - exp(poly) above; max error < 1 ulp (8.8e-8 relative, checked against
// scipy.special.erf over [-6, 6] on 2M points, tests/test_host_logic.py holds the numpy twin).
__device__ __forceinline__ float erf_fast(float a) {
  const float t = fabsf(a), s = a * a;
  float r = fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
  const float u = fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
  r = fmaf(r, s, u);
  r = fmaf(r, t, -1.06777877e-1f);
  r = fmaf(r, t, -6.34846687e-1f);
  r = fmaf(r, t, -1.28717512e-1f);
  r = fmaf(r, t, -t);
  const float big = copysignf(1.0f - __expf(r), a);
  float q = -5.96761703e-4f;
  q = fmaf(q, s, 4.99119423e-3f);
  q = fmaf(q, s, -2.67681349e-2f);
  q = fmaf(q, s, 1.12819925e-1f);
  q = fmaf(q, s, -3.76125336e-1f);
  q = fmaf(q, s, 1.28379166e-1f);
  const float small = fmaf(q, a, a);
  return t > 0.927734375f ? big : small;
}
// nn.GELU() (exact erf form), swin_transformer.py:21
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752440f)); }

// GELU for the bf16 engine (the GELU cost of a GEMM epilogue is VALU issue slots: 42 % of the fused MLP at C = 128,
// profiles/r02f_mlp_trace.txt).  One range, no select:  gelu(x) = max(x, 0) - 0.5 |x| erfc(|x| / sqrt 2)  with
// erfc(t / sqrt 2) = 2^(t q(t)), q a degree-5 polynomial fitted on [0, 6] (|x| is clamped there: erfc < 2e-9).  Two
// elements per call so that the polynomial runs on v_pk_fma_f32.  Max |error| 3.2e-7 over [-8, 8] in fp32 arithmetic
// (numpy twin in tests/test_host_logic.py), i.e. 1/2000 of a bf16 ulp at |x| ~ 0.1 -- the fp32 engine keeps erf_fast.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_fast2(f32x2 x) {
  const f32x2 a = {fabsf(x[0]), fabsf(x[1])};
  const f32x2 t = {fminf(a[0], 6.0f), fminf(a[1], 6.0f)};
  f32x2 q = {2.992467125e-05f, 2.992467125e-05f};
  q = __builtin_elementwise_fma(q, t, f32x2{-7.398781599e-04f, -7.398781599e-04f});
  q = __builtin_elementwise_fma(q, t, f32x2{7.977474481e-03f, 7.977474481e-03f});
  q = __builtin_elementwise_fma(q, t, f32x2{-5.323820189e-02f, -5.323820189e-02f});
  q = __builtin_elementwise_fma(q, t, f32x2{-4.589156806e-01f, -4.589156806e-01f});
  q = __builtin_elementwise_fma(q, t, f32x2{-1.151147127e+00f, -1.151147127e+00f});
  const f32x2 p = q * t;
  const f32x2 e = {__builtin_amdgcn_exp2f(p[0]), __builtin_amdgcn_exp2f(p[1])};
  const f32x2 h = a * f32x2{-0.5f, -0.5f};
  const f32x2 r = {fmaxf(x[0], 0.0f), fmaxf(x[1], 0.0f)};
  return __builtin_elementwise_fma(h, e, r);
}
// n (even) values in place
template <int N>
__device__ __forceinline__ void gelu_fast_n(float* v) {
#pragma unroll
  for (int i = 0; i < N; i += 2) {
    const f32x2 g = gelu_fast2(f32x2{v[i], v[i + 1]});
    v[i] = g[0]; v[i + 1] = g[1];
  }
}

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
