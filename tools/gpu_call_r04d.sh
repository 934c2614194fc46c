#!/bin/bash
# round 4, call d: gemm_4w (256x256 on four waves) -- bit-exactness vs gemm_dma, then the Swin-B shapes at 32-image chunks, bf16 and bf16x3
OUT=gpurun_out/r04d; mkdir -p $OUT; export TMPDIR=/tmp
timeout 60 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); torch.cuda.synchronize(); assert float(x.sum()) == float(1 << 20)" > $OUT/canary.log 2>&1 || { echo canary failed; exit 3; }
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "gemm_4w" > $OUT/test_4w.log 2>&1; echo "4w rc=$?" >> $OUT/rc.log; tail -30 $OUT/test_4w.log
KBENCH_GEMM_VARIANTS=9,10,11 KBENCH_GEMM_MSCALE=4 KBENCH_GEMM_ONLY=4,5,6,7,8,9,10,11,12,13,14,15,16 timeout 400 python tools/kbench.py gemm > $OUT/kbench_gemm_bf16.txt 2>&1; echo "kbench rc=$?" >> $OUT/rc.log; cat $OUT/kbench_gemm_bf16.txt
KBENCH_GEMM_X3=1 KBENCH_GEMM_VARIANTS=9,10,11 KBENCH_GEMM_MSCALE=4 KBENCH_GEMM_ONLY=8,9,10,11,12,13,14,15 timeout 400 python tools/kbench.py gemm > $OUT/kbench_gemm_x3.txt 2>&1; echo "kbench x3 rc=$?" >> $OUT/rc.log; cat $OUT/kbench_gemm_x3.txt
cat $OUT/rc.log
