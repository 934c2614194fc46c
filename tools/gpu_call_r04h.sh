#!/bin/bash
OUT=gpurun_out/r04h; mkdir -p $OUT; export TMPDIR=/tmp
timeout 60 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); torch.cuda.synchronize(); assert float(x.sum()) == float(1 << 20)" > $OUT/canary.log 2>&1 || { echo canary failed; exit 3; }
timeout 300 python tools/kbench.py cross_split > $OUT/kbench_cross_split.txt 2>&1; grep "rows/img=64" $OUT/kbench_cross_split.txt
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "gemm_x3 or check_gemm or gemm_4w" > $OUT/test_gemm.log 2>&1; echo "gemm rc=$?" >> $OUT/rc.log; tail -5 $OUT/test_gemm.log
timeout 900 python -m pytest tests/test_gpu_e2e.py -q -k "parity_engine" > $OUT/test_parity.log 2>&1; echo "parity rc=$?" >> $OUT/rc.log; tail -5 $OUT/test_parity.log
KBENCH_GEMM_X3=1 KBENCH_GEMM_VARIANTS=9 KBENCH_GEMM_MSCALE=4 KBENCH_GEMM_ONLY=6,10,14 timeout 300 python tools/kbench.py gemm > $OUT/kbench_x3_gelu.txt 2>&1; cat $OUT/kbench_x3_gelu.txt
cat $OUT/rc.log
