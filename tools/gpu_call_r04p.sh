#!/bin/bash
# gemm_4w_r (selector 16: weights streamed into registers): quick equality probe under a short timeout, the bit-exactness check, kbench
OUT=gpurun_out/r04p; mkdir -p $OUT; export TMPDIR=/tmp
timeout 60 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); torch.cuda.synchronize(); assert float(x.sum()) == float(1 << 20)" > $OUT/canary.log 2>&1 || { echo canary failed; exit 3; }
timeout 120 python - > $OUT/probe.txt 2>&1 <<'P'
import torch
from advancedliteratemachinery_amd import ops
torch.manual_seed(0)
for (M, N, K) in ((256, 256, 256), (1000, 512, 512), (4096, 1536, 2048)):
    A = torch.randn(M, K, device='cuda').bfloat16(); W = (torch.randn(N, K, device='cuda') / K ** 0.5).bfloat16(); b = torch.randn(N, device='cuda')
    ops.force_gemm_kernel(5); y5 = ops.gemm(A, W, b); ops.force_gemm_kernel(16); y16 = ops.gemm(A, W, b); ops.force_gemm_kernel(0)
    torch.cuda.synchronize()
    ne = (y5.view(torch.int16) != y16.view(torch.int16)).sum().item()
    print('probe %dx%dx%d: %d of %d elements differ; max |d| %.4g' % (M, N, K, ne, y5.numel(), (y5.float() - y16.float()).abs().max().item()), flush=True)
    if ne:
        d = (y5.float() - y16.float()).abs()
        bad = (d > 0).nonzero()
        print('  first bad', bad[:8].tolist(), 'rows with errors', bad[:, 0].unique().numel(), 'cols with errors', bad[:, 1].unique().numel())
P
echo "probe rc=$?"; cat $OUT/probe.txt
grep -q " 0 of" $OUT/probe.txt || { echo "probe failed: stopping"; exit 0; }
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "gemm_4w" > $OUT/pytest_gemm_4w.txt 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gemm_4w.txt
KBENCH_GEMM_VARIANTS=9,10,16,17 KBENCH_GEMM_MSCALE=4 KBENCH_GEMM_ONLY=4,5,6,7,8,9,10,11,12,13,14,15,16 timeout 500 python tools/kbench.py gemm > $OUT/kbench_gemm_bf16.txt 2>&1; cat $OUT/kbench_gemm_bf16.txt
KBENCH_GEMM_X3=1 KBENCH_GEMM_VARIANTS=9,10,16 KBENCH_GEMM_MSCALE=4 KBENCH_GEMM_ONLY=4,5,6,7,8,9,10,11,12,13,14,15 timeout 500 python tools/kbench.py gemm > $OUT/kbench_gemm_x3.txt 2>&1; cat $OUT/kbench_gemm_x3.txt
