// Probe (development aid, not product): at what rate can ONE compute unit pull a weight stream that every other compute
// unit pulls at the same time (L2 / MALL resident), and how much of it survives matrix-core work fed from LDS?
// It is the inner loop of csrc/dec_rows.hip (a workgroup owns RT rows of the decoders' many-row phases, every wave walks its own
// linear stream of 1 KB operand fragments) with nothing around it:
//   mode 0: the stream alone (global_load_dwordx4 -> registers, PF fragments in flight per wave, xor-folded)
//   mode 1: every fragment feeds RTT matrix-core instructions whose other operand is read from LDS (the real loop)
//   mode 2: the same stream through LDS DMA into a wave-private ring (no consumption)
// usage: probe_stream            (prints one line per configuration)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int NW, int PF, int RTT, int MODE, bool NT>
__global__ __launch_bounds__(NW * 64) void probe_kernel(const char* __restrict__ W, int nfrag_per_wave, float* __restrict__ out, int stride_wg) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // every wave walks its OWN contiguous stream of 1 KB fragments (stride_wg = 0: every workgroup the same bytes; > 0: its own copy)
  const char* src = W + (size_t)blockIdx.x * stride_wg + (size_t)wave * nfrag_per_wave * 1024 + lane * 16;
  const size_t step = 1024;
  u32x4 ring[PF];
  // hipcc does not keep a register ring of loads in flight (it sinks the loads to their uses and drains vmcnt(0)): asm loads, hand-counted waits
  auto ld = [&](u32x4& dst, int i) {
    const char* p = src + (size_t)i * step;
    if constexpr (NT) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(dst) : "v"(p) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
  };
  auto wait_use = [&](u32x4& r) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(r) : "n"(PF - 1) : "memory"); };
  if constexpr (MODE == 2) {
    char* mine = lds + wave * PF * 1024;
    for (int i = 0; i < nfrag_per_wave; ++i) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)i * step),
                                       (__attribute__((address_space(3))) void*)(mine + (i % PF) * 1024), 16, 0, NT ? 2 : 0);
      if ((i % PF) == PF - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PF / 2) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0 && wave == 0) out[blockIdx.x] = reinterpret_cast<float*>(lds)[0];
    return;
  } else {
    // LDS "activation" tile for mode 1: RTT row tiles x 16 rows x 512 k, pitch 1056 bytes
    constexpr int PITCH = 1056;
    if constexpr (MODE == 1) {
      for (int i = threadIdx.x; i < RTT * 16 * PITCH / 4; i += NW * 64) reinterpret_cast<float*>(lds)[i] = 0.001f * (float)(i & 255);
      __syncthreads();
    }
    const int li = lane & 15, g = lane >> 4;
#pragma unroll
    for (int u = 0; u < PF; ++u) ld(ring[u], u);
    u32x4 fold = {0, 0, 0, 0};
    f32x4 acc[4][RTT];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int r = 0; r < RTT; ++r) acc[a][r] = f32x4{0.f, 0.f, 0.f, 0.f};
    // fragments are consumed in groups of 4 (the 4 feature tiles of one k-step); per k-step the RTT activation fragments are read once
    const int ngroups = nfrag_per_wave / PF;
    for (int gi = 0; gi < ngroups; ++gi) {
      bf16x8 bfr[RTT];
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int idx = gi * PF + u;
        wait_use(ring[u]);
        const u32x4 w = ring[u];
        if constexpr (MODE == 0) {
          fold ^= w;
        } else {
          const int ks = (idx >> 2) & 15;
          constexpr int dummy = 0; (void)dummy;
          bf16x8 wf;
          __builtin_memcpy(&wf, &w, 16);
          if ((u & 3) == 0) {   // one k-step = 4 feature tiles: the activation fragments are read once per k-step
#pragma unroll
            for (int r = 0; r < RTT; ++r) bfr[r] = *reinterpret_cast<const bf16x8*>(lds + (r * 16 + li) * PITCH + ks * 64 + g * 16);
          }
#pragma unroll
          for (int r = 0; r < RTT; ++r) acc[u & 3][r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, bfr[r], acc[u & 3][r], 0, 0, 0);
        }
        ld(ring[u], idx + PF);   // unconditional: the buffer has PF fragments of slack behind every stream
      }
    }
    float s = 0.f;
    if constexpr (MODE == 0) {
      s = (float)(fold[0] ^ fold[1] ^ fold[2] ^ fold[3]);
    } else {
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < RTT; ++r) s += acc[a][r][0] + acc[a][r][1] + acc[a][r][2] + acc[a][r][3];
    }
    if (s == 123.456f) out[blockIdx.x * NW * 64 + threadIdx.x] = s;
  }
}

template <int NW, int PF, int RTT, int MODE, bool NT>
void run(const char* name, const char* W, size_t bytes, int wgs, float* out, bool own_copy = false) {
  const int nfrag_per_wave = (int)(bytes / 1024 / NW) / PF * PF;
  size_t smem = MODE == 1 ? (size_t)RTT * 16 * 1056 : (MODE == 2 ? (size_t)NW * PF * 1024 : 0);
  auto kern = probe_kernel<NW, PF, RTT, MODE, NT>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int stride = own_copy ? (int)bytes : 0;
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(wgs), dim3(NW * 64), smem, 0, W, nfrag_per_wave, out, stride);
  CK(hipDeviceSynchronize());
  const int iters = 20;
  CK(hipEventRecord(a, 0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(wgs), dim3(NW * 64), smem, 0, W, nfrag_per_wave, out, stride);
  CK(hipEventRecord(b, 0));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  const double us = ms * 1e3 / iters;
  const double per_wg = (double)nfrag_per_wave * NW * 1024;
  const double flops = MODE == 1 ? (double)nfrag_per_wave * NW * RTT * 16 * 16 * 32 * 2 * wgs : 0.0;
  printf("%-34s NW=%-2d PF=%-2d RTT=%d nt=%d wgs=%-4d stream/WG=%5.2f MB : %8.1f us  %6.1f GB/s per WG  %6.2f TB/s aggregate  %7.1f TF/s\n", name, NW, PF, RTT, (int)NT, wgs,
         per_wg / 1e6, us, per_wg / us / 1e3, per_wg * wgs / us / 1e6, flops / us / 1e6);
  fflush(stdout);
}

int main() {
  const size_t MAXB = 8u << 20;
  char* W;
  float* out;
  CK(hipMalloc(&W, MAXB * 40));   // room for per-workgroup private copies (the HBM-streaming reference point)
  CK(hipMemset(W, 1, MAXB * 40));
  CK(hipMalloc(&out, 1 << 24));
  hipDeviceProp_t pr;
  CK(hipGetDeviceProperties(&pr, 0));
  printf("%s, %d CUs\n", pr.name, pr.multiProcessorCount);
  // the stream alone: depth and wave count
  run<8, 4, 1, 0, false>("stream only, shared 7.3 MB", W, 7340032, 256, out);
  run<8, 8, 1, 0, false>("stream only, shared 7.3 MB", W, 7340032, 256, out);
  run<8, 16, 1, 0, false>("stream only, shared 7.3 MB", W, 7340032, 256, out);
  run<16, 8, 1, 0, false>("stream only, shared 7.3 MB", W, 7340032, 256, out);
  run<4, 16, 1, 0, false>("stream only, shared 7.3 MB", W, 7340032, 256, out);
  run<8, 8, 1, 0, true>("stream only, shared 7.3 MB", W, 7340032, 256, out);
  run<8, 8, 1, 0, false>("stream only, shared 2 MB", W, 2097152, 256, out);
  run<8, 8, 1, 0, false>("stream only, shared 0.5 MB", W, 524288, 256, out);
  run<8, 8, 1, 0, false>("stream only, shared 7.3 MB", W, 7340032, 128, out);
  run<8, 8, 1, 0, false>("stream only, shared 7.3 MB", W, 7340032, 512, out);
  run<8, 8, 1, 0, false>("stream only, OWN 7.3 MB copy (HBM)", W, 7340032, 40, out, true);
  // LDS DMA transport
  run<8, 8, 1, 2, false>("LDS DMA ring, shared 7.3 MB", W, 7340032, 256, out);
  run<8, 8, 1, 2, true>("LDS DMA ring, shared 7.3 MB", W, 7340032, 256, out);
  // the real loop: RTT matrix-core instructions per fragment, the other operand from LDS
  run<8, 8, 4, 1, false>("stream + MFMA (64 rows)", W, 7340032, 256, out);
  run<8, 8, 5, 1, false>("stream + MFMA (80 rows)", W, 7340032, 256, out);
  run<8, 8, 6, 1, false>("stream + MFMA (96 rows)", W, 7340032, 256, out);
  run<8, 4, 5, 1, false>("stream + MFMA (80 rows)", W, 7340032, 256, out);
  run<8, 16, 5, 1, false>("stream + MFMA (80 rows)", W, 7340032, 256, out);
  run<8, 8, 5, 1, true>("stream + MFMA (80 rows)", W, 7340032, 256, out);
  run<16, 8, 5, 1, false>("stream + MFMA (80 rows)", W, 7340032, 256, out);
  run<8, 8, 5, 1, false>("stream + MFMA (80 rows)", W, 7340032, 128, out);
  run<8, 8, 2, 1, false>("stream + MFMA (32 rows)", W, 7340032, 512, out);
  run<8, 8, 3, 1, false>("stream + MFMA (48 rows)", W, 7340032, 256, out);
  return 0;
}
