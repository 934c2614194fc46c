#!/bin/bash
OUT=gpurun_out/r01f; mkdir -p $OUT
for sc in test nograph sameshape onelane; do
  DIAG_TIMEOUT=40 timeout 80 python tools/lane_diag.py $sc > $OUT/diag_$sc.log 2>&1; echo "diag $sc rc=$?" >> $OUT/rc.log
done
timeout 150 python tools/lane_sweep.py --lanes 1,2,3,4,6,8 > $OUT/sweep_q4.log 2>&1; echo "sweep rc=$?" >> $OUT/rc.log
GPU_MAX_HW_QUEUES=8 timeout 150 python tools/lane_sweep.py --lanes 3,4,6,8 > $OUT/sweep_q8.log 2>&1; echo "sweep8 rc=$?" >> $OUT/rc.log
cat $OUT/rc.log
