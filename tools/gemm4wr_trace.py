"""Where does a gemm_4w_r tile (256x256, selector 16) spend its time?  Selector 18 = the same kernel with s_memtime stamps of wave 0 at
start / X tile 0 landed / main loop done / every wave done / stores retired.  Prints per shape the median and p90 of each phase in
timer ticks, next to the tile's matrix-core time (K / 64 stages x 128 MFMAs x 16 cycles per wave).
    python tools/gemm4wr_trace.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from advancedliteratemachinery_amd import _lib, ops  # noqa: E402

# (M, N, K, fp32 residual + fp32 out)
SHAPES = [(131072, 1536, 512, 0), (131072, 512, 512, 1), (131072, 512, 2048, 1), (131072, 6144, 512, 0), (32768, 3072, 1024, 0), (32768, 1024, 4096, 1)]


def q(t, f):
    return float(torch.quantile(t.double(), f))


def main():
    dev = 'cuda'
    h = _lib.lib()
    for (M, N, K, res) in SHAPES:
        A = torch.randn(M, K, device=dev).to(torch.bfloat16)
        W = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
        bias = torch.randn(N, device=dev)
        odt = torch.float32 if res else torch.bfloat16
        out = torch.empty(M, N, device=dev, dtype=odt)
        r = torch.randn(M, N, device=dev).to(odt) if res else None
        nwg = ((M + 255) // 256) * ((N + 255) // 256)
        trace = torch.zeros(nwg, 8, dtype=torch.int64, device=dev)
        _lib.check(h.omp_debug_set_gemm_trace(ops.ptr(trace), nwg), 'omp_debug_set_gemm_trace')
        ops.force_gemm_kernel(18)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):   # last run: steady clocks, warm instruction cache
            trace.zero_()
            ev0.record()
            ops.gemm(A, W, bias, residual=r, out=out)
            ev1.record()
        torch.cuda.synchronize()
        us = ev0.elapsed_time(ev1) * 1e3
        ops.force_gemm_kernel(0)
        _lib.check(h.omp_debug_set_gemm_trace(None, 0), 'omp_debug_set_gemm_trace')
        t = trace.cpu()
        # s_memtime counters of different XCDs are not aligned: the span of the kernel is taken per XCD (median over the XCDs)
        xcc = t[:, 5] & 15
        walls = [int(t[xcc == x, 4].max() - t[xcc == x, 0].min()) for x in sorted(set(xcc.tolist()))]
        wall = sorted(walls)[len(walls) // 2]
        ph = [(t[:, i + 1] - t[:, i]) for i in range(4)]
        names = ['prologue (launch -> X tile 0 landed)', 'main loop (wave 0)', 'drain + barrier', 'epilogue (stores retired)']
        print('gemm_4w_r %dx%dx%d res=%d : %d tiles, %.1f us by events, span of one XCD %d ticks (%.0f ticks per us)'
              % (M, N, K, res, nwg, us, wall, wall / us))
        tot = (t[:, 4] - t[:, 0])
        for n_, p_ in zip(names, ph):
            print('    %-38s p50 %8.0f  p90 %8.0f ticks  (%.0f%% of the median tile)' % (n_, q(p_, 0.5), q(p_, 0.9), 100 * q(p_, 0.5) / q(tot, 0.5)))
        print('    %-38s p50 %8.0f ticks; matrix-core time of a tile: %d cycles; %.1f tile lifetimes fit in the kernel wall (rounds needed: %.1f)'
              % ('whole tile', q(tot, 0.5), K // 64 * 128 * 16, wall / q(tot, 0.5), nwg / 256.0), flush=True)


if __name__ == '__main__':
    main()
