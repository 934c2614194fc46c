"""Where does a fused-MLP workgroup spend its time?  (development aid)  Runs the TRACE instantiation (variant 100) of
csrc/mlp.hip at the Swin-B stage shapes (B = 8, 1024x1024) and prints the median cycle split of wave 0."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from advancedliteratemachinery_amd import _lib, ops  # noqa: E402
from advancedliteratemachinery_amd.model.packing import pack_mlp  # noqa: E402


def main():
    dev = 'cuda'
    h = _lib.lib()
    for (M, C, rows_per_wg) in ((524288, 128, 128), (131072, 256, 128), (32768, 512, 128)):
        Hd = 4 * C
        x = torch.randn(M, C, device=dev).to(torch.bfloat16)
        g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        w1 = (torch.randn(Hd, C, device=dev) / C ** 0.5).to(torch.bfloat16)
        w2 = (torch.randn(C, Hd, device=dev) / Hd ** 0.5).to(torch.bfloat16)
        b1, b2 = torch.randn(Hd, device=dev) * 0.1, torch.randn(C, device=dev) * 0.1
        pack = pack_mlp(w1, b1, w2)
        out = torch.empty_like(x)
        nwg = (M + rows_per_wg - 1) // rows_per_wg
        trace = torch.zeros(nwg, 8, dtype=torch.int64, device=dev)
        h.omp_debug_swin_mlp_trace(ops.ptr(trace))
        ops.swin_mlp_variant(100)
        for _ in range(2):
            ops.swin_mlp_fused(x, g, b, pack, b2, out=out)
        torch.cuda.synchronize()
        ops.swin_mlp_variant(0)
        h.omp_debug_swin_mlp_trace(None)
        t = trace.cpu().double()
        med = t.median(dim=0).values
        names = ['whole workgroup', 'rows + LayerNorm', 'DMA wait + barrier', 'first product', 'GELU', 'second product', 'epilogue']
        span = (t[:, 7].max() - t[:, 7].min() + med[0]).item()
        print('mlp C=%d M=%d: %d workgroups, %d sub-chunks; kernel span %.0f cycles' % (C, M, nwg, Hd // 32, span))
        for i, n in enumerate(names):
            print('    %-20s %9.0f cycles  (%5.1f%%)  per sub-chunk %7.0f' % (n, med[i].item(), 100 * med[i].item() / med[0].item(), med[i].item() / (Hd // 32)))
        sys.stdout.flush()


if __name__ == '__main__':
    main()
