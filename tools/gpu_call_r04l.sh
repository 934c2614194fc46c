#!/bin/bash
OUT=gpurun_out/r04l; mkdir -p $OUT; export TMPDIR=/tmp
timeout 60 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); torch.cuda.synchronize(); assert float(x.sum()) == float(1 << 20)" > $OUT/canary.log 2>&1 || { echo canary failed; exit 3; }
KBENCH_GEMM_F32RES=0 KBENCH_GEMM_VARIANTS=9,10,12,13,14 KBENCH_GEMM_MSCALE=4 KBENCH_GEMM_ONLY=11,15 timeout 400 python tools/kbench.py gemm > $OUT/kbench_ablation_bigk.txt 2>&1; cat $OUT/kbench_ablation_bigk.txt
