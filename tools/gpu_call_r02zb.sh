#!/bin/bash
# r02zb: the corrected context test; true batch-8 engine calls (coalesce 1) with the chunked vs block-per-step q4 kernel
OUT=gpurun_out/r02zb; mkdir -p $OUT; export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "contexts or masked" > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/rc.log; tail -3 $OUT/tests.log
timeout 200 python tools/kbench.py cross > $OUT/kbench_cross8.txt 2>&1; echo "kbench rc=$?" >> $OUT/rc.log
for m in 1 2 1 2; do
  timeout 200 python bench.py --coalesce 1 --steps 128 --warmup 16 --min-seconds 3 --q4-mode $m --no-cpu-baseline --no-batch8 --no-eos-run --no-roofline > $OUT/b8_q4_$m.json 2> $OUT/b8.err; echo "b8 q4=$m rc=$?" >> $OUT/rc.log
  python -c "
import json; d=json.load(open('$OUT/b8_q4_$m.json')); print('coalesce 1, q4 mode $m: %.1f img/s %.2f ms/step (p10 %.2f p90 %.2f)' % (d['value'], d['ms_per_step'], d['timing']['ms_per_step_p10'], d['timing']['ms_per_step_p90']))" >> $OUT/summary.txt
done
cat $OUT/rc.log $OUT/summary.txt; grep "rows/img=64" $OUT/kbench_cross8.txt
