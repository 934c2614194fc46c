#!/bin/bash
# gemm_4w_p (selector 20: persistent, register-only epilogue): quick equality probe under a short timeout, the bit-exactness check, kbench
OUT=gpurun_out/r04s; mkdir -p $OUT; export TMPDIR=/tmp
timeout 60 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); torch.cuda.synchronize(); assert float(x.sum()) == float(1 << 20)" > $OUT/canary.log 2>&1 || { echo canary failed; exit 3; }
timeout 150 python - > $OUT/probe.txt 2>&1 <<'P'
import torch
from advancedliteratemachinery_amd import ops
torch.manual_seed(0)
ok = True
for (M, N, K, f32) in ((256, 256, 256, 0), (1024, 512, 512, 0), (8192, 4096, 512, 0), (131072, 1536, 512, 0), (8192, 512, 512, 1), (131072, 512, 2048, 1)):
    A = torch.randn(M, K, device='cuda').bfloat16(); W = (torch.randn(N, K, device='cuda') / K ** 0.5).bfloat16(); b = torch.randn(N, device='cuda')
    r = torch.randn(M, N, device='cuda') if f32 else None
    kw = dict(residual=r, out_dtype=torch.float32) if f32 else {}
    ops.force_gemm_kernel(5); y5 = ops.gemm(A, W, b, **kw); ops.force_gemm_kernel(20); y20 = ops.gemm(A, W, b, **kw); ops.force_gemm_kernel(0)
    torch.cuda.synchronize()
    vt = torch.int32 if f32 else torch.int16
    ne = (y5.view(vt) != y20.view(vt)).sum().item()
    print('probe %dx%dx%d f32res=%d: %d of %d elements differ; max |d| %.4g' % (M, N, K, f32, ne, y5.numel(), (y5.float() - y20.float()).abs().max().item()), flush=True)
    if ne:
        ok = False
        bad = ((y5.float() - y20.float()).abs() > 0).nonzero()
        print('  first bad', bad[:8].tolist(), 'rows with errors', bad[:, 0].unique().numel(), 'cols with errors', bad[:, 1].unique().numel(), 'row tiles', (bad[:, 0] // 256).unique()[:16].tolist(), 'col tiles', (bad[:, 1] // 256).unique()[:16].tolist())
print('PROBE_OK' if ok else 'PROBE_BAD')
P
echo "probe rc=$?"; cat $OUT/probe.txt
grep -q "PROBE_OK" $OUT/probe.txt || { echo "probe failed: stopping"; exit 0; }
KBENCH_GEMM_VARIANTS=9,16,20 KBENCH_GEMM_MSCALE=4 KBENCH_GEMM_ONLY=4,5,6,7,8,9,10,11,12,13,14,15,16 timeout 500 python tools/kbench.py gemm 2>&1 | grep "^gemm" | tee $OUT/kbench_gemm_bf16.txt
KBENCH_GEMM_X3=1 KBENCH_GEMM_VARIANTS=9,16,20 KBENCH_GEMM_MSCALE=4 KBENCH_GEMM_ONLY=4,5,6,7,8,9,10,11,12,13,14,15 timeout 500 python tools/kbench.py gemm 2>&1 | grep "^gemm" | tee $OUT/kbench_gemm_x3.txt
timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "gemm_4w" > $OUT/pytest_gemm_4w.txt 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gemm_4w.txt
