#!/bin/bash
OUT=gpurun_out/r01g; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/rc.log
KBENCH_GEMM_VARIANTS=5,7,8 timeout 300 python tools/kbench.py cross dec_gemm selfattn gemm > $OUT/kbench.txt 2>&1; echo "kbench rc=$?" >> $OUT/rc.log
timeout 200 python tools/lane_sweep.py --lanes 1,2,4 --prio 0,1 --batches 8 --steps 12 > $OUT/sweep_b8.log 2>&1; echo "sweep8 rc=$?" >> $OUT/rc.log
timeout 200 python tools/lane_sweep.py --lanes 1,2 --prio 0 --batches 16,32 --steps 16 > $OUT/sweep_b16_32.log 2>&1; echo "sweep32 rc=$?" >> $OUT/rc.log
cat $OUT/rc.log; tail -3 $OUT/tests.log
