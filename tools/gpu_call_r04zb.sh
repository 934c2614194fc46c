#!/bin/bash
# kernel tables of the closing tree: rocprofv3 --kernel-trace --stats over the driver's bench command, eager launches, bf16 engine alone / parity engine alone
OUT=gpurun_out/r04zb; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
timeout 60 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); torch.cuda.synchronize(); assert float(x.sum()) == float(1 << 20)" > $OUT/canary.log 2>&1 || { echo canary failed; exit 3; }
for eng in bf16 bf16x3; do
  (cd /tmp && timeout 420 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_$eng -o ks -- python $R/bench.py --dtype $eng --steps 20 --warmup 5 --min-seconds 0 --no-parity-leg --no-config-legs --no-batch8 --no-eos-run --no-cpu-baseline --no-roofline --graph 0 > $R/$OUT/prof_bench_$eng.json 2> $R/$OUT/prof_$eng.err); echo "prof $eng rc=$?" >> $OUT/rc.log
  db=$(find $OUT/prof_$eng -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocpd_stats.py $db > $OUT/kernel_stats_$eng.txt 2>> $OUT/rc.log && python tools/rocpd_shapes.py $db 2.0 > $OUT/kernel_shapes_$eng.txt 2>> $OUT/rc.log
  rm -rf $OUT/prof_$eng
done
cat $OUT/rc.log; head -16 $OUT/kernel_stats_bf16.txt | cut -c1-190; head -14 $OUT/kernel_stats_bf16x3.txt | cut -c1-190
