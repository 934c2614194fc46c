#!/bin/bash
# r02r: fast-GELU effect on the GEMM / fused-MLP micro-benchmarks + a single-lane per-(kernel, grid) profile of 256-image engine calls
OUT=gpurun_out/r02r; mkdir -p $OUT; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); export TMPDIR=/tmp
KBENCH_GEMM_VARIANTS=0 timeout 300 python tools/kbench.py gemm mlp > $OUT/kbench.txt 2>&1; echo "kbench rc=$?" >> $OUT/rc.log
(cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o ks -- python $R/bench.py --steps 64 --warmup 0 --min-seconds 0 --lanes 1 --no-cpu-baseline --no-batch8 --no-eos-run --no-roofline > $R/$OUT/prof_bench.json 2> $R/$OUT/prof.err); echo "prof rc=$?" >> $OUT/rc.log
db=$(find $OUT/prof -name "*.db" 2>/dev/null | head -1)
[ -n "$db" ] && python tools/rocpd_stats.py $db > $OUT/kernel_stats.txt 2>> $OUT/prof.err
[ -n "$db" ] && python tools/rocpd_shapes.py $db 2.0 > $OUT/kernel_shapes.txt 2>> $OUT/prof.err
rm -rf $OUT/prof
cat $OUT/rc.log
