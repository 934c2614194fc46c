"""Where does a gemm_dma<128,128,2> tile spend its time?  (development aid, round-2 starting point)

Runs the Swin-B GEMM shapes through the TRACE instantiation (omp_debug_force_gemm_kernel(15)): wave 0 of every
workgroup stamps s_memtime (tick = shader cycle) at start / first K tile landed / K loop done / accumulators in LDS /
stores retired.  Prints per shape the median and p90 of each phase in cycles, the spread of workgroup start times
(dispatch ramp) and the fraction of the kernel's wall time covered by the median workgroup.
    python tools/gemm_trace.py [--shapes all|qkv]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from advancedliteratemachinery_amd import _lib, ops  # noqa: E402

SHAPES = [(524288, 384, 128, 0), (524288, 128, 128, 1), (524288, 512, 128, 0), (524288, 128, 512, 1),
          (131072, 768, 256, 0), (131072, 256, 1024, 1), (32768, 1536, 512, 0), (32768, 512, 512, 1), (32768, 2048, 512, 0),
          (32768, 512, 2048, 1), (8192, 3072, 1024, 0), (8192, 1024, 4096, 1)]


def q(t, f):
    return float(torch.quantile(t.double(), f))


def main():
    dev = 'cuda'
    h = _lib.lib()
    for (M, N, K, res) in SHAPES:
        A = torch.randn(M, K, device=dev).to(torch.bfloat16)
        W = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
        bias = torch.randn(N, device=dev)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        r = torch.randn(M, N, device=dev).to(torch.bfloat16) if res else None
        nwg = ((M + 127) // 128) * ((N + 127) // 128)
        trace = torch.zeros(nwg, 8, dtype=torch.int64, device=dev)
        _lib.check(h.omp_debug_set_gemm_trace(ops.ptr(trace), nwg), 'omp_debug_set_gemm_trace')
        ops.force_gemm_kernel(15)
        for _ in range(2):   # second run: steady clocks, warm instruction cache
            trace.zero_()
            ops.gemm(A, W, bias, residual=r, out=out)
        torch.cuda.synchronize()
        ops.force_gemm_kernel(0)
        _lib.check(h.omp_debug_set_gemm_trace(None, 0), 'omp_debug_set_gemm_trace')
        t = trace.cpu()
        t0 = t[:, 0].min()
        wall = int(t[:, 4].max() - t0)
        ph = [(t[:, i + 1] - t[:, i]) for i in range(4)]
        names = ['prologue (launch -> first K tile)', 'K loop', 'acc -> LDS (+ residual request)', 'stores']
        print('gemm %dx%dx%d res=%d : %d workgroups, kernel wall %d cycles; start spread p50 %.0f p100 %.0f cycles; XCC ids seen %s'
              % (M, N, K, res, nwg, wall, q(t[:, 0] - t0, 0.5), float((t[:, 0] - t0).max()), sorted(set((t[:, 5] & 15).tolist()))[:9]))
        tot = (t[:, 4] - t[:, 0])
        for n_, p_ in zip(names, ph):
            print('    %-36s p50 %8.0f  p90 %8.0f cycles  (%.0f%% of the median workgroup)' % (n_, q(p_, 0.5), q(p_, 0.9), 100 * q(p_, 0.5) / q(tot, 0.5)))
        print('    %-36s p50 %8.0f cycles; %.1f workgroup lifetimes fit in the kernel wall time (>= rounds needed: %.1f)'
              % ('whole workgroup', q(tot, 0.5), wall / q(tot, 0.5), nwg / 512.0), flush=True)


if __name__ == '__main__':
    main()
