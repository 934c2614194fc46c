#!/bin/bash
OUT=gpurun_out/r04f; mkdir -p $OUT; export TMPDIR=/tmp
timeout 60 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); torch.cuda.synchronize(); assert float(x.sum()) == float(1 << 20)" > $OUT/canary.log 2>&1 || { echo canary failed; exit 3; }
KBENCH_GEMM_VARIANTS=9,10 KBENCH_GEMM_MSCALE=4 KBENCH_GEMM_ONLY=4,5,6,7,8,9,10,11,12,13,14,15,16 timeout 400 python tools/kbench.py gemm > $OUT/kbench_gemm_bf16.txt 2>&1; cat $OUT/kbench_gemm_bf16.txt
KBENCH_GEMM_X3=1 KBENCH_GEMM_VARIANTS=9,10 KBENCH_GEMM_MSCALE=4 KBENCH_GEMM_ONLY=4,5,6,7,8,9,10,11,12,13,14,15 timeout 400 python tools/kbench.py gemm > $OUT/kbench_gemm_x3.txt 2>&1; cat $OUT/kbench_gemm_x3.txt
KBENCH_GEMM_VARIANTS=9,10 timeout 300 python tools/kbench.py kvproj > $OUT/kbench_kvproj.txt 2>&1; cat $OUT/kbench_kvproj.txt
