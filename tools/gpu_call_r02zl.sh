#!/bin/bash
# r02zl: rocprofv3 kernel stats of bench.py at its committed defaults (one lane, 512 images per engine call)
OUT=gpurun_out/r02zl; mkdir -p $OUT; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); export TMPDIR=/tmp
(cd /tmp && timeout 170 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o ks -- python $R/bench.py --steps 64 --warmup 0 --min-seconds 0 --no-cpu-baseline --no-batch8 --no-eos-run > $R/$OUT/prof_bench.json 2> $R/$OUT/prof.err); echo "prof rc=$?" >> $OUT/rc.log
db=$(find $OUT/prof -name "*.db" 2>/dev/null | head -1)
[ -n "$db" ] && python tools/rocpd_stats.py $db > $OUT/kernel_stats.txt 2>> $OUT/prof.err
[ -n "$db" ] && python tools/rocpd_shapes.py $db 3.0 > $OUT/kernel_shapes.txt 2>> $OUT/prof.err
rm -rf $OUT/prof
cat $OUT/rc.log; head -8 $OUT/kernel_stats.txt | cut -c1-70,108-160
