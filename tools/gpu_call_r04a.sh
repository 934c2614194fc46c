#!/bin/bash
# round 4, call a: the r4-prep kernels (fused attention half at C = 256, leaner stage-0 kernel) for the first time on hardware,
# and the parity engine's per-kernel time table.
OUT=gpurun_out/r04a; mkdir -p $OUT; R=$(pwd); export TMPDIR=/tmp
if ! timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); torch.cuda.synchronize(); assert float(x.sum()) == float(1 << 20)" > $OUT/canary.log 2>&1; then
  echo "canary failed" | tee $OUT/rc.log; tail -3 $OUT/canary.log; exit 3
fi
timeout 300 python -m pytest tests/test_gpu_ops.py -q -k "swin_block" > $OUT/test_swin_block.log 2>&1; echo "swin_block rc=$?" >> $OUT/rc.log; tail -15 $OUT/test_swin_block.log
timeout 200 python tools/kbench.py swin_block > $OUT/kbench_swin_block.txt 2>&1; echo "kbench rc=$?" >> $OUT/rc.log; cat $OUT/kbench_swin_block.txt | tail -12
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o ks -- python $R/bench.py --dtype bf16x3 --steps 20 --warmup 5 --min-seconds 0 --no-parity-leg --no-config-legs --no-batch8 --no-eos-run --no-cpu-baseline --no-roofline --graph 0 --phase-times > $R/$OUT/prof_bench_x3.json 2> $R/$OUT/prof_x3.err); echo "prof x3 rc=$?" >> $OUT/rc.log
db=$(find $OUT/prof -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocpd_stats.py $db > $OUT/kernel_stats_x3.txt 2>> $OUT/prof_x3.err && python tools/rocpd_shapes.py $db 2.0 > $OUT/kernel_shapes_x3.txt 2>> $OUT/prof_x3.err
rm -rf $OUT/prof
tail -3 $OUT/prof_x3.err; head -30 $OUT/kernel_stats_x3.txt
cat $OUT/rc.log
