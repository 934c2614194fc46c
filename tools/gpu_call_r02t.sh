#!/bin/bash
# r02t: ViT attention kernel parity + MGP-STR batch-512 throughput, then the whole GPU suite on the fast-GELU tree
OUT=gpurun_out/r02t; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_mgp.py -m gpu -q -x > $OUT/tests_mgp.log 2>&1; echo "tests_mgp rc=$?" >> $OUT/rc.log; tail -3 $OUT/tests_mgp.log
timeout 300 python tools/mgp_bench.py 512 5 > $OUT/mgp_bench.txt 2>&1; echo "mgp rc=$?" >> $OUT/rc.log
OMP355_PARITY_REPORT=$OUT/parity_report.json timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_mgp.py > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/rc.log; tail -3 $OUT/tests.log
cat $OUT/rc.log $OUT/mgp_bench.txt
