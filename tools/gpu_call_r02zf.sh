#!/bin/bash
# r02zf: the committed tree at the end of round 2 -- smoke, the whole GPU suite, `python bench.py` exactly as the driver runs it
OUT=gpurun_out/r02zf; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke rc=$?" >> $OUT/rc.log; tail -1 $OUT/smoke.txt
OMP355_PARITY_REPORT=$OUT/parity_report.json timeout 900 python -m pytest tests -m gpu -q > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/rc.log; tail -2 $OUT/tests.log
( time timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench_time.txt; echo "bench rc=$?" >> $OUT/rc.log
cat $OUT/rc.log $OUT/bench_time.txt; head -c 700 $OUT/bench.json; echo; tail -3 $OUT/bench.err
