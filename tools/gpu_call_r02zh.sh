#!/bin/bash
# r02zh: bench.py with step counts a driver might pass (fewer steps than the coalescing width, odd remainders) and the side workloads
OUT=gpurun_out/r02zh; mkdir -p $OUT; export TMPDIR=/tmp
for args in "--steps 20 --warmup 5" "--steps 5 --warmup 2" "--steps 70 --warmup 3 --lanes 2" ; do
  timeout 300 python bench.py $args --min-seconds 1 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err; rc=$?
  python -c "
import json; d=json.load(open('$OUT/b.json')); print('bench.py $args : rc=$rc  %.1f img/s  images/call %d  batch8 %s  eos %s  roofline %s' % (d['value'], d['config']['images_per_engine_call'], d.get('batch8',{}).get('images_per_sec', d.get('batch8')), d.get('eos_run',{}).get('images_per_sec', d.get('eos_run')), d['roofline']['kernel'][:24] if d.get('roofline') else None))" >> $OUT/summary.txt 2>&1 || { echo "bench.py $args : rc=$rc FAILED" >> $OUT/summary.txt; tail -5 $OUT/b.err >> $OUT/summary.txt; }
done
timeout 300 python bench.py --workload kie --steps 4 --warmup 1 --min-seconds 1 --no-roofline > $OUT/kie.json 2> $OUT/kie.err; echo "kie rc=$?" >> $OUT/summary.txt
timeout 300 python bench.py --workload mgp_str --steps 10 --warmup 2 --min-seconds 1 --no-roofline --no-cpu-baseline > $OUT/mgp.json 2> $OUT/mgp.err; echo "mgp rc=$?" >> $OUT/summary.txt
cat $OUT/summary.txt
