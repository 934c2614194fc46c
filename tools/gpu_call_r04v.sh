#!/bin/bash
# A/B on one box: gemm_256 (k9) vs gemm_4w_p (k20) on the products the new dispatch rule moves (MGP-STR's ViT-B qkv / fc1, Swin stage-2/3 fc1, bf16x3 stage-1 qkv), 3 repetitions
OUT=gpurun_out/r04v; mkdir -p $OUT; export TMPDIR=/tmp
timeout 60 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); torch.cuda.synchronize(); assert float(x.sum()) == float(1 << 20)" > $OUT/canary.log 2>&1 || { echo canary failed; exit 3; }
for rep in 1 2 3; do
KBENCH_GEMM_F32RES=0 KBENCH_GEMM_SHAPES="131584,2304,768,0,0;131584,3072,768,1,0;131072,2048,512,1,0;32768,4096,1024,1,0;524288,768,256,0,0" KBENCH_GEMM_VARIANTS=9,20 timeout 300 python tools/kbench.py gemm 2>&1 | grep "^gemm"
KBENCH_GEMM_X3=1 KBENCH_GEMM_SHAPES="524288,768,256,0,0" KBENCH_GEMM_VARIANTS=9,10,20 timeout 300 python tools/kbench.py gemm 2>&1 | grep "^gemm"
done | tee $OUT/kbench_ab.txt
