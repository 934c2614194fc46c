#!/bin/bash
# r02zk: the driver's step count (--steps 20 --warmup 5) with one lane (one 160-image call) vs two lanes (two 80-image calls in flight)
OUT=gpurun_out/r02zk; mkdir -p $OUT; export TMPDIR=/tmp
for l in 1 2 1 2; do
  timeout 200 python bench.py --steps 20 --warmup 5 --lanes $l --min-seconds 3 --no-cpu-baseline --no-batch8 --no-eos-run --no-roofline > $OUT/b.json 2> $OUT/b.err
  python -c "
import json; d=json.load(open('$OUT/b.json')); print('--steps 20 --warmup 5 --lanes $l : %.1f img/s  %.2f ms/step  (%d reps, p10 %.2f p90 %.2f)  images/call %d' % (d['value'], d['ms_per_step'], d['timing']['repeats'], d['timing']['ms_per_step_p10'], d['timing']['ms_per_step_p90'], d['config']['images_per_engine_call']))" >> $OUT/summary.txt
done
cat $OUT/summary.txt
