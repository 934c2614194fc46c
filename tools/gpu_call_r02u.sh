#!/bin/bash
# r02u: GELU consistency fix (batch == single), gemm_256 slab epilogues, CU-mask probe
OUT=gpurun_out/r02u; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_ops.py -m gpu -q -x -k "batch_equals_single or gemm or cross or decoder" > $OUT/tests_sel.log 2>&1; echo "tests_sel rc=$?" >> $OUT/rc.log; tail -3 $OUT/tests_sel.log
KBENCH_GEMM_VARIANTS=5,0 timeout 200 python tools/kbench.py kvproj > $OUT/kbench_kvproj.txt 2>&1; echo "kvproj rc=$?" >> $OUT/rc.log
timeout 300 python tools/cu_mask_probe.py > $OUT/cu_mask_probe.txt 2>&1; echo "probe rc=$?" >> $OUT/rc.log
cat $OUT/rc.log $OUT/kbench_kvproj.txt $OUT/cu_mask_probe.txt
