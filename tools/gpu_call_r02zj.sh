#!/bin/bash
# NOTE: the --split-points flag these runs used was removed with the (not kept) split point decoder; kept as the record of the commands
# r02zj: split point decoder only for small calls (<= 16 images): small-call legs A/B, small e2e tests
OUT=gpurun_out/r02zj; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x -k "spot_odd or kie_sroie or batch_equals or graph or lanes" > $OUT/tests_sel.log 2>&1; echo "tests_sel rc=$?" >> $OUT/rc.log; tail -2 $OUT/tests_sel.log
for sp in 0 16 0 16; do
  timeout 300 python bench.py --steps 32 --warmup 8 --min-seconds 1 --no-cpu-baseline --no-roofline --split-points $sp > $OUT/b.json 2> $OUT/b.err
  python -c "
import json; d=json.load(open('$OUT/b.json')); print('split-points $sp : batch8 %.1f img/s (p10 %.1f p90 %.1f ms/step)  eos %.1f img/s' % (d['batch8']['images_per_sec'], d['batch8']['ms_per_step_p10'], d['batch8']['ms_per_step_p90'], d['eos_run']['images_per_sec']))" >> $OUT/summary.txt
done
cat $OUT/rc.log $OUT/summary.txt
