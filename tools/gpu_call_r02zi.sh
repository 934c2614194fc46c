#!/bin/bash
# NOTE: the --split-points flag these runs used was removed with the (not kept) split point decoder; kept as the record of the commands
# r02zi: point decoder as two halves on two streams -- end-to-end parity tests, then A/B at the driver's step count and at the default
OUT=gpurun_out/r02zi; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x > $OUT/tests_e2e.log 2>&1; echo "tests_e2e rc=$?" >> $OUT/rc.log; tail -3 $OUT/tests_e2e.log
run() { timeout 300 python bench.py --no-cpu-baseline --no-eos-run --no-roofline --min-seconds 3 "$@" > $OUT/b.json 2> $OUT/b.err; python -c "
import json,sys; d=json.load(open('$OUT/b.json')); print('%-52s : %.1f img/s  %.2f ms/step  (%d reps, p10 %.2f p90 %.2f)  batch8 %s' % (' '.join(sys.argv[1:]), d['value'], d['ms_per_step'], d['timing']['repeats'], d['timing']['ms_per_step_p10'], d['timing']['ms_per_step_p90'], d.get('batch8',{}).get('images_per_sec')))" "$@" >> $OUT/summary.txt 2>&1 || { echo "FAILED $@" >> $OUT/summary.txt; tail -4 $OUT/b.err >> $OUT/summary.txt; }; }
run --steps 20 --warmup 5 --split-points 0
run --steps 20 --warmup 5 --split-points 1
run --steps 192 --warmup 32 --split-points 0 --no-batch8
run --steps 192 --warmup 32 --split-points 1 --no-batch8
cat $OUT/rc.log $OUT/summary.txt
