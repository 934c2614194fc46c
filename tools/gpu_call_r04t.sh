#!/bin/bash
# would a bf16x3 GEMM that fetches each operand tile once (2/3 of the bytes) be faster?  selector 21 = gemm_4w_p with every third stage's requests thinned
OUT=gpurun_out/r04t; mkdir -p $OUT; export TMPDIR=/tmp
timeout 60 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); torch.cuda.synchronize(); assert float(x.sum()) == float(1 << 20)" > $OUT/canary.log 2>&1 || { echo canary failed; exit 3; }
KBENCH_GEMM_X3=1 KBENCH_GEMM_VARIANTS=9,20,21 KBENCH_GEMM_MSCALE=4 KBENCH_GEMM_ONLY=4,5,7,8,9,11,12,13,15 timeout 500 python tools/kbench.py gemm 2>&1 | grep "^gemm" | tee $OUT/kbench_gemm_x3_thin.txt
