#!/bin/bash
# the fused three-product bf16x3 kernel (selector 22): probe vs the three-pass kernel, check_gemm_4w, kbench against k9 / k10 / k20
OUT=gpurun_out/r04y; mkdir -p $OUT; export TMPDIR=/tmp
timeout 60 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); torch.cuda.synchronize(); assert float(x.sum()) == float(1 << 20)" > $OUT/canary.log 2>&1 || { echo canary failed; exit 3; }
timeout 150 python - > $OUT/probe.txt 2>&1 <<'P'
import torch
from advancedliteratemachinery_amd import ops
torch.manual_seed(0)
ok = True
for (M, N, K) in ((256, 256, 256), (1024, 512, 512), (8192, 1536, 512), (131072, 512, 2048)):
    A = ops.split_bf16(torch.randn(M, K, device='cuda')); W = ops.split_weight3(torch.randn(N, K, device='cuda') / K ** 0.5); b = torch.randn(N, device='cuda')
    r = torch.randn(M, N, device='cuda')
    kw = dict(residual=r, out_dtype=torch.float32, a_wrap=2 * K)
    ops.force_gemm_kernel(5); y5 = ops.gemm(A, W, b, **kw); ops.force_gemm_kernel(22); y22 = ops.gemm(A, W, b, **kw); ops.force_gemm_kernel(0)
    torch.cuda.synchronize()
    d = (y5 - y22).abs()
    print('probe x3 %dx%dx%d: max |d| %.3g (scale %.3g), elements above 1e-5: %d of %d' % (M, N, 3 * K, d.max().item(), y5.abs().max().item(), (d > 1e-5).sum().item(), d.numel()), flush=True)
    if d.max().item() > 2e-5:
        ok = False
        bad = (d > 1e-5).nonzero()
        print('  first bad', bad[:8].tolist(), 'rows', bad[:, 0].unique().numel(), 'cols', bad[:, 1].unique().numel(), 'row tiles', (bad[:, 0] // 256).unique()[:16].tolist(), 'col tiles', (bad[:, 1] // 256).unique()[:16].tolist())
print('PROBE_OK' if ok else 'PROBE_BAD')
P
echo "probe rc=$?"; cat $OUT/probe.txt
grep -q "PROBE_OK" $OUT/probe.txt || { echo "probe failed: stopping"; exit 0; }
KBENCH_GEMM_X3=1 KBENCH_GEMM_VARIANTS=9,10,20,22 KBENCH_GEMM_MSCALE=4 KBENCH_GEMM_ONLY=4,5,6,7,8,9,10,11,12,13,14,15 timeout 500 python tools/kbench.py gemm 2>&1 | grep "^gemm" | tee $OUT/kbench_gemm_x3_fused.txt
timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "gemm_4w" > $OUT/pytest_gemm_4w.txt 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gemm_4w.txt
