#!/bin/bash
# r02y: whole GPU suite on the omp_ctx tree; q4 cross-attention variants (non-temporal DMA, ring depth); bench
OUT=gpurun_out/r02y; mkdir -p $OUT; export TMPDIR=/tmp
OMP355_PARITY_REPORT=$OUT/parity_report.json timeout 900 python -m pytest tests -m gpu -q > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/rc.log; tail -4 $OUT/tests.log
KBENCH_CROSS_IMAGES=256 timeout 200 python tools/kbench.py cross128 > $OUT/kbench_cross256.txt 2>&1; echo "kbench rc=$?" >> $OUT/rc.log
timeout 300 python bench.py --steps 192 --warmup 64 --min-seconds 3 --no-cpu-baseline --no-batch8 --no-eos-run --no-roofline --phase-times > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" >> $OUT/rc.log
python -c "
import json; d=json.load(open('$OUT/bench.json')); print('bench: %.1f img/s %.2f ms/step' % (d['value'], d['ms_per_step']))" >> $OUT/summary.txt
grep "phase ms" $OUT/bench.err >> $OUT/summary.txt
cat $OUT/rc.log $OUT/summary.txt; grep -v amdgpu $OUT/kbench_cross256.txt
