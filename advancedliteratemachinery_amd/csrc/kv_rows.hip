// Cross-attention memory projection as a row-owner stream kernel (round 5).
//
// Reference: OCR/OmniParser/model/transformer.py:88-96, 442-446 -- nn.MultiheadAttention projects the image memory to keys and values inside
// every decoder layer of every step; here K = (memory + pos) Wk^T + bk and V = memory Wv^T + bv of all 12 (decoder, layer) pairs are computed
// ONCE per engine call into the head-blocked slabs the cross-attention kernels stream:
//     K   [slab][image][head][Mpad][64]            V^T [slab][image][head][Mpad / 32][64][32 key slots]
// Until round 5 these were two tiled GEMMs (655 360 x 6144 x 512 at 160 images: 256 x 256 output tiles, 8 K-tiles each, 128 KB of output per
// tile with nothing to overlap it with): 6.2 + 5.9 ms = 0.27 of the matrix-core peak, 2.6 % of the headline.  The product is row-local with a
// short K, i.e. exactly the shape of the row-owner chains (csrc/dec_rows.hip): a workgroup of eight waves keeps 64 memory rows (two 32-key
// blocks of one image) as a bf16 operand tile in LDS and streams the 12 x 512 x 512 weights of all slabs through the matrix cores -- per-wave
// fragment streams in consumption order (model/packing.py::pack_kv_rows_k / _v), the asm register ring of rows_common.inc.  Wave w owns the 64
// features of HEAD w of every slab, and the two slab layouts fall out of the accumulators without a transpose:
//   * K (A operand = weight fragment, D[feature][row]): the packer permutes a head's dims so that matrix-core row 4 g + r of feature tile ft
//     is dim 32 (ft / 2) + 8 g + 4 (ft % 2) + r -- a lane then holds dims 8 g .. 8 g + 7 and 32 + 8 g .. of one key: two 16-byte stores, each of
//     which four lanes make 64 contiguous bytes (a first version with 16 consecutive dims per lane wrote 16-byte pieces at a 32-byte stride);
//   * V^T (operands swapped, D[row][feature]): a lane holds keys 4 g + r of a 16-key tile for ONE dim; the slot order of a 32-key block
//     (slot 8 g + 4 half + r <- key 16 half + 4 g + r, the order the first cross-attention product delivers P) puts the two tiles of a block
//     side by side: 8 consecutive slots = one 16-byte store, sixteen lanes x four g complete 16 dims x 64 bytes = 1 KB contiguous.
// Same MFMA, same ascending-k accumulation and the same rounding points as the tiled kernels: the slabs are bit-identical
// (tests/gpu_checks.py::check_kv_rows).
#include <type_traits>
#include <utility>

#include "common.h"

namespace {

#include "rows_common.inc"

// Weight fragments in flight per wave.  MEASURED (160 images: 655 360 rows, profiles/r05zc-r05zm_kbench_kv_rows*.txt): 3.9-4.1 ms per launch
// = 1.0 PF, against 5.7-6.2 ms for the tiled GEMMs -- and the same with a ring of 16, with 128 rows per workgroup, with the K stores as 64-byte
// runs instead of 16-byte pieces, and with a slab's stores issued one per ring revolution under the NEXT slab's product.  The phase clocks of
// wave 0 (p.trace): a workgroup holds its CU for 158 k cycles = products 114 k (issue floor of its 3072 MFMAs per wave at two waves per SIMD:
// 98 k) + pack and stores 44 k (786 KB at the ~10 B / clk a CU writes); with TWO workgroups per CU (half-slab passes, 106 registers) products
// take 207 k (floor 196 k), stores 82 k, the launch 4.4 ms.  The product phase is matrix-core-bound at the clock the chip sustains under it
// (s_memtime / s_memrealtime: 1.60-1.73 GHz, not the nominal 2.4); what is left is the store phase that nothing overlaps.
constexpr int PF = 8;
constexpr int A_PITCH = D * 2 + 32;    // operand tile row pitch, bytes (conflict-free b128 fragment reads, as csrc/dec_rows.hip)
constexpr int TILE_SLACK = 64;         // the operand prefetch reads one k-step past the last row

struct KvP {
  const bf16_t* a;        // [rows, 512] memory (+ pos) rows, bf16
  const char* wstream;    // packed weights of all slabs
  int64_t wave_stride;
  const float* bias;      // [n_slabs * 512]
  bf16_t* out;            // slab base
  int64_t rows;           // B * M
  int M, Mpad, B, n_slabs;
  unsigned long long* trace;   // development (omp_debug_swin_mlp_trace): [workgroup][8] shader clocks (s_memtime) of wave 0 -- 0 both, 2 products, 3 pack + stores; 4 = the same interval in 10 ns ticks (s_memrealtime)
};

constexpr int RTT = 4, RT = RTT * 16;  // 64 rows per workgroup: two 32-key blocks
constexpr int NP = 2 * RTT;            // 16-byte pieces per lane and slab

// A slab's results of this lane, rounded and packed, and where they go.
struct Pieces {
  bf16x8 v[NP];
  bf16_t* base;     // lane base of the slab
};

template <bool SWAP>
__device__ __forceinline__ int64_t piece_offset(int i) {   // elements from Pieces::base, static per piece
  if constexpr (!SWAP) return (int64_t)(i >> 1) * 16 * 64 + (i & 1) * 32;      // K: piece (rt, half): key tile rt, dims 8 g .. / 32 + 8 g ..
  else return ((int64_t)(i & 1) * 64 + (i >> 1) * 16) * 32;                     // V^T: piece (ft, b): key block b, dims 16 ft + li
}

// acc (+ bias), rounded to bf16 and packed into the lane's NP pieces of slab nl
template <bool SWAP>
__device__ __forceinline__ void pack_slab(Pieces& pc, const f32x4 (&acc)[4][RTT], const KvP& p, int nl, int image, int key0, int wave, int li_, int g_) {
  const int li = opaque(li_), g = opaque(g_);
  const int64_t head = ((int64_t)nl * p.B + image) * NW + wave;   // (slab, image, head): 8 heads = the 8 waves
  if constexpr (!SWAP) {
    // lane: key rt * 16 + li, dims 32 (ft / 2) + 8 g + 4 (ft % 2) + r: tiles 0, 1 are dims 8 g .. 8 g + 7, tiles 2, 3 the same + 32
    const float* bp = p.bias + nl * D + wave * 64 + g * 8;
    f32x4 bb[4];
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) bb[ft] = *reinterpret_cast<const f32x4*>(bp + (ft >> 1) * 32 + (ft & 1) * 4);
    pc.base = p.out + (head * p.Mpad + key0 + li) * 64 + g * 8;
#pragma unroll
    for (int rt = 0; rt < RTT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pc.v[2 * rt][r] = (bf16_t)(acc[0][rt][r] + bb[0][r]);          // four lanes: 64 contiguous bytes of the key's row
        pc.v[2 * rt][4 + r] = (bf16_t)(acc[1][rt][r] + bb[1][r]);
        pc.v[2 * rt + 1][r] = (bf16_t)(acc[2][rt][r] + bb[2][r]);
        pc.v[2 * rt + 1][4 + r] = (bf16_t)(acc[3][rt][r] + bb[3][r]);
      }
  } else {
    // lane: keys rt * 16 + 4 g + r, dim 16 ft + li; tiles 2 b and 2 b + 1 are the halves of key block b
    const float* bp = p.bias + nl * D + wave * 64 + li;
    pc.base = p.out + ((head * (p.Mpad >> 5) + (key0 >> 5)) * 64 + li) * 32 + g * 8;
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) {
      const float bv = bp[ft * 16];
#pragma unroll
      for (int b = 0; b < RTT / 2; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          pc.v[2 * ft + b][r] = (bf16_t)(acc[ft][2 * b][r] + bv);
          pc.v[2 * ft + b][4 + r] = (bf16_t)(acc[ft][2 * b + 1][r] + bv);
        }
    }
  }
}

// acc[ft][rt] += over K = 512 (16 k-steps); the wave's next 64 stream fragments, ordered (k-step, feature tile).  SWAP: the row fragment is the
// A operand (D[row][feature]) -- the register contents of both fragments are the same either way.
template <bool SWAP>
__device__ __forceinline__ void gemm_pass_kv(f32x4 (&acc)[4][RTT], const char* a_lane, u32x4 (&ring)[PF], Stream& st) {
  constexpr int NFT = 4, KS = 16, NG = NFT * KS / PF, KPG = PF / NFT;
  bf16x8 bfr[2][RTT];
#pragma unroll
  for (int rt = 0; rt < RTT; ++rt) bfr[0][rt] = *reinterpret_cast<const bf16x8*>(a_lane + rt * 16 * A_PITCH);
#pragma unroll 1
  for (int gi = 0; gi < NG; ++gi) {
    sfor<PF>([&](auto U) {
      constexpr int u = decltype(U)::value, ft = u % NFT, kk = u / NFT;
      if constexpr (ft == 0) {
        const char* ak = a_lane + (gi * KPG + kk + 1) * 64;
#pragma unroll
        for (int rt = 0; rt < RTT; ++rt) bfr[(kk + 1) & 1][rt] = *reinterpret_cast<const bf16x8*>(ak + rt * 16 * A_PITCH);
      }
      const bf16x8 wf = ws_take<u>(ring);
#pragma unroll
      for (int rt = 0; rt < RTT; ++rt) {
        if constexpr (SWAP) acc[ft][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[kk & 1][rt], wf, acc[ft][rt], 0, 0, 0);
        else acc[ft][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, bfr[kk & 1][rt], acc[ft][rt], 0, 0, 0);
      }
      ws_issue<u>(ring, st);
    });
  }
}

template <bool SWAP>
__device__ __forceinline__ void store_slab(const f32x4 (&acc)[4][RTT], const KvP& p, int nl, int image, int key0, int wave, int li, int g) {
  Pieces pc;
  pack_slab<SWAP>(pc, acc, p, nl, image, key0, wave, li, g);
#pragma unroll
  for (int i = 0; i < NP; ++i) *reinterpret_cast<bf16x8*>(pc.base + piece_offset<SWAP>(i)) = pc.v[i];
}

template <bool SWAP>
__global__ __launch_bounds__(NW * 64) void kv_rows_kernel(KvP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* tile = smem;                                               // RT x A_PITCH
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t r0 = (int64_t)blockIdx.x * RT;
  Stream st = stream_of_wave(p.wstream, p.wave_stride, wave, lane);
  u32x4 ring[PF];
  sfor<PF>([&](auto U) { ws_issue<decltype(U)::value>(ring, st); });
  // the workgroup's rows -> operand tile by LDS DMA: a row is 1 KB = one wave instruction (as csrc/dec_rows.hip stage_rows)
#pragma unroll
  for (int i = 0; i < RT / NW; ++i) {
    const int row = wave * (RT / NW) + i;
    int64_t r = r0 + row;
    if (r > p.rows - 1) r = p.rows - 1;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.a + r * D + lane * 8),
                                     (__attribute__((address_space(3))) void*)(tile + row * A_PITCH), 16, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  lds_barrier();
  const bool tracing = p.trace != nullptr && wave == 0;
  unsigned long long tr[4] = {0, 0, 0, 0};
  unsigned long long t_last = tracing ? __builtin_amdgcn_s_memtime() : 0ull;
  const unsigned long long rt0 = tracing ? __builtin_amdgcn_s_memrealtime() : 0ull;   // the constant 100 MHz counter: calibrates the shader clock
  auto lap = [&](int i) {
    if (tracing) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      tr[i] += t - t_last;
      t_last = t;
    }
  };
  const char* a_lane = tile + li * A_PITCH + g * 16;
  const int image = (int)(r0 / p.M), key0 = (int)(r0 - (int64_t)image * p.M);   // M % 64 == 0: a workgroup never straddles images
  f32x4 acc[4][RTT];
#pragma unroll 1
  for (int nl = 0; nl < p.n_slabs - 1; ++nl) {
    zero_acc(acc);
    gemm_pass_kv<SWAP>(acc, a_lane, ring, st);
    lap(2);
    store_slab<SWAP>(acc, p, nl, image, key0, wave, li, g);
    lap(3);
  }
  zero_acc(acc);
  gemm_pass_kv<SWAP>(acc, a_lane, ring, st);
  ws_drain(ring);   // the ring's run-ahead requests (PF fragments of slack behind the stream)
  lap(2);
  store_slab<SWAP>(acc, p, p.n_slabs - 1, image, key0, wave, li, g);
  if (tracing) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lap(3);
    if (lane == 0) {
      unsigned long long* o = p.trace + (int64_t)blockIdx.x * 8;
      o[0] = tr[2] + tr[3];
      o[2] = tr[2];
      o[3] = tr[3];
      o[4] = __builtin_amdgcn_s_memrealtime() - rt0;   // 10 ns ticks over the same interval as o[0]
    }
  }
}

template <bool SWAP>
int launch_kv(const KvP& p, hipStream_t st) {
  const size_t smem = (size_t)RT * A_PITCH + TILE_SLACK;
  auto kern = kv_rows_kernel<SWAP>;
  static bool done = false;   // per instantiation
  if (!done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
      omp_set_error("omp_kv_project_rows: cannot raise dynamic LDS limit");
      return OMP_ERR_LAUNCH;
    }
    done = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(p.rows / RT)), dim3(NW * 64), smem, st, p);
  OMP_CHECK_LAUNCH("omp_kv_project_rows");
  return OMP_OK;
}

}  // namespace

extern "C" int omp_kv_project_rows(const void* rows, const void* wstream, int64_t wave_stride, const float* bias, void* out, int B, int M, int Mpad,
                                   int n_slabs, int vt, omp_stream_t s) {
  OMP_CHECK_ARG(rows && wstream && bias && out, "omp_kv_project_rows: null pointer");
  OMP_CHECK_ARG(B > 0 && M > 0 && M % 64 == 0 && Mpad >= M && Mpad % 32 == 0 && n_slabs > 0, "omp_kv_project_rows: M must be a multiple of 64 (two 32-key blocks per workgroup), Mpad >= M a multiple of 32 (B=%d M=%d Mpad=%d slabs=%d)", B, M, Mpad, n_slabs);
  OMP_CHECK_ARG((int64_t)B * M / 64 < (1ll << 31), "omp_kv_project_rows: too many rows");
  OMP_CHECK_ARG(wave_stride >= (int64_t)n_slabs * 64 * 1024 && wave_stride % 16 == 0 && ((uintptr_t)wstream % 16) == 0 && ((uintptr_t)rows % 16) == 0 && ((uintptr_t)out % 16) == 0,
                "omp_kv_project_rows: a wave's stream holds %d fragments of 1 KB; 16-byte aligned pointers", n_slabs * 64);
  KvP p;
  p.a = reinterpret_cast<const bf16_t*>(rows); p.wstream = reinterpret_cast<const char*>(wstream); p.wave_stride = wave_stride;
  p.bias = bias; p.out = reinterpret_cast<bf16_t*>(out); p.rows = (int64_t)B * M; p.M = M; p.Mpad = Mpad; p.B = B; p.n_slabs = n_slabs;
  p.trace = omp_cur().mlp_trace;
  hipStream_t st = (hipStream_t)s;
  const double fl = 2.0 * (double)p.rows * D * D * n_slabs;
  const double by = (double)p.rows * D * 2 + (double)p.rows * D * n_slabs * 2 + (double)n_slabs * D * D * 2;
  const int slot = omp_prof_active(OMP_PROF_GEMM) ? omp_prof_begin(OMP_PROF_GEMM, st, fl, by) : -1;
  const int rc = vt ? launch_kv<true>(p, st) : launch_kv<false>(p, st);
  if (slot >= 0) omp_prof_end(OMP_PROF_GEMM, slot, st);
  return rc;
}
