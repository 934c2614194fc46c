#!/bin/bash
# r02x: whole GPU suite on the register-swap reductions / chunked q4 tree, kernel A/B, images-per-call sweep
OUT=gpurun_out/r02x; mkdir -p $OUT; export TMPDIR=/tmp
OMP355_PARITY_REPORT=$OUT/parity_report.json timeout 900 python -m pytest tests -m gpu -q > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/rc.log; tail -4 $OUT/tests.log
KBENCH_CROSS_IMAGES=256 timeout 200 python tools/kbench.py cross128 misc > $OUT/kbench_cross_misc.txt 2>&1; echo "kbench rc=$?" >> $OUT/rc.log
for co in 32 64; do
  timeout 300 python bench.py --steps 192 --warmup 64 --min-seconds 3 --coalesce $co --no-cpu-baseline --no-batch8 --no-eos-run --no-roofline > $OUT/bench_co$co.json 2> $OUT/bench_co$co.err; echo "bench coalesce=$co rc=$?" >> $OUT/rc.log
  python -c "
import json; d=json.load(open('$OUT/bench_co$co.json')); print('coalesce $co: %.1f img/s %.2f ms/step' % (d['value'], d['ms_per_step']))" >> $OUT/summary.txt
done
cat $OUT/rc.log $OUT/summary.txt; grep -v amdgpu $OUT/kbench_cross_misc.txt
