#!/bin/bash
OUT=gpurun_out/r04e; mkdir -p $OUT; export TMPDIR=/tmp
timeout 60 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); torch.cuda.synchronize(); assert float(x.sum()) == float(1 << 20)" > $OUT/canary.log 2>&1 || { echo canary failed; exit 3; }
timeout 300 python -m pytest tests/test_gpu_ops.py -q -k "gemm_4w" > $OUT/test_4w.log 2>&1; echo "4w rc=$?" >> $OUT/rc.log; tail -5 $OUT/test_4w.log
KBENCH_GEMM_VARIANTS=9,10,12,13,14 KBENCH_GEMM_MSCALE=4 KBENCH_GEMM_ONLY=8,10,12,16 timeout 400 python tools/kbench.py gemm > $OUT/kbench_ablation.txt 2>&1; cat $OUT/kbench_ablation.txt
KBENCH_GEMM_X3=1 KBENCH_GEMM_VARIANTS=9,10 KBENCH_GEMM_MSCALE=4 KBENCH_GEMM_ONLY=8,10,11,15 timeout 400 python tools/kbench.py gemm > $OUT/kbench_x3.txt 2>&1; cat $OUT/kbench_x3.txt
cat $OUT/rc.log
