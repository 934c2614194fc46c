#!/bin/bash
# One GPU-box session: parity tests, bench line, rocprofv3 kernel stats, PMC passes (HBM bytes).
# usage: tools/gpu_round.sh <tag> [legs]   legs: subset of "tests bench prof pmc kbench" (default: all but kbench)
TAG=${1:-r01x}; LEGS=${2:-"tests bench prof pmc"}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for leg in $LEGS; do case $leg in
tests) timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/rc.log;;
bench) timeout 600 python bench.py --phase-times > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" >> $OUT/rc.log;;
prof)  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o ks -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/prof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/prof.err); echo "prof rc=$?" >> $OUT/rc.log
       db=$(find $OUT/prof -name "*.db" 2>/dev/null | head -1); [ -n "$db" ] && python tools/rocpd_stats.py $db > $OUT/kernel_stats.txt 2>> $OUT/prof.err;;
pmc)   for c in FETCH_SIZE WRITE_SIZE; do
         (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_$c -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --lanes 1 --graph 0 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/$OUT/pmc_$c.json 2> $GRAFT_REPO_ROOT/$OUT/pmc_$c.err); echo "pmc $c rc=$?" >> $OUT/rc.log
         f=$(find $OUT/pmc_$c -name "*counter_collection.csv" 2>/dev/null | head -1); [ -n "$f" ] && python tools/pmc_summary.py $f $c > $OUT/pmc_$c.txt 2>> $OUT/pmc_$c.err; rm -rf $OUT/pmc_$c
       done;;
kbench) timeout 600 python tools/kbench.py all > $OUT/kbench.txt 2>&1; echo "kbench rc=$?" >> $OUT/rc.log;;
esac; done
find $OUT/prof -name "*.db" -size +20M -delete 2>/dev/null
cat $OUT/rc.log
