#!/bin/bash
# One GPU-box session (round 2): usage tools/gpu_round2.sh <tag> "<legs>"   legs: tests bench prof kbench mlp pmc
TAG=${1:-r02x}; LEGS=${2:-"tests bench prof"}
OUT=gpurun_out/$TAG; mkdir -p $OUT
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
export TMPDIR=/tmp
for leg in $LEGS; do case $leg in
tests) OMP355_PARITY_REPORT=$OUT/parity_report.json timeout 900 python -m pytest tests -m gpu -q ${PYTEST_ARGS} > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/rc.log; tail -3 $OUT/tests.log;;
bench) timeout 900 python bench.py --phase-times ${BENCH_ARGS} > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" >> $OUT/rc.log;;
bench20) timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err; echo "bench20 rc=$?" >> $OUT/rc.log;;
prof)  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o ks -- python $R/bench.py --steps 32 --warmup 0 --min-seconds 0 --no-cpu-baseline --no-batch8 --no-eos-run > $R/$OUT/prof_bench.json 2> $R/$OUT/prof.err); prc=$?; echo "prof rc=$prc" >> $OUT/rc.log
       if [ $prc -ne 0 ]; then
         rm -rf $OUT/prof; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o ks -- python $R/bench.py --steps 32 --warmup 0 --min-seconds 0 --no-cpu-baseline --no-batch8 --no-eos-run --graph 0 > $R/$OUT/prof_bench.json 2> $R/$OUT/prof_graph0.err); echo "prof(graph 0) rc=$?" >> $OUT/rc.log
       fi
       db=$(find $OUT/prof -name "*.db" 2>/dev/null | head -1); [ -n "$db" ] && python tools/rocpd_stats.py $db > $OUT/kernel_stats.txt 2>> $OUT/prof.err
       find $OUT/prof -name "*.db" -size +20M -delete 2>/dev/null;;
kbench) timeout 600 python tools/kbench.py ${KBENCH_WHAT:-gemm} > $OUT/kbench.txt 2>&1; echo "kbench rc=$?" >> $OUT/rc.log;;
pmc)   for I in ${PMC_IMAGES:-80 128}; do for c in FETCH_SIZE WRITE_SIZE; do
         (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$OUT/pmc_${c}_$I -o pmc -- python $R/tools/cross_pmc.py $I > $R/$OUT/pmc_${c}_$I.log 2>&1); echo "pmc $c $I rc=$?" >> $OUT/rc.log
         f=$(find $OUT/pmc_${c}_$I -name "*counter_collection.csv" 2>/dev/null | head -1); [ -n "$f" ] && python tools/pmc_summary.py $f $c > $OUT/pmc_${c}_$I.txt 2>> $OUT/rc.log; rm -rf $OUT/pmc_${c}_$I
         done
         python tools/pmc_cross_json.py $OUT/pmc_FETCH_SIZE_$I.txt $OUT/pmc_WRITE_SIZE_$I.txt $I "rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> --kernel-trace --output-format csv -- python tools/cross_pmc.py $I" $OUT/pmc_cross_attn.json > $OUT/pmc_cross_attn.json.new 2>> $OUT/rc.log && mv $OUT/pmc_cross_attn.json.new $OUT/pmc_cross_attn.json
       done;;
mfma)  (cd /tmp && KBENCH_GEMM_VARIANTS=0 timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/$OUT/pmc_mfma -o pmc -- python $R/tools/kbench.py gemm mlp > $R/$OUT/pmc_mfma_kbench.log 2>&1); echo "mfma rc=$?" >> $OUT/rc.log
       f=$(find $OUT/pmc_mfma -name "*counter_collection.csv" 2>/dev/null | head -1); [ -n "$f" ] && python tools/pmc_mfma.py $f > $OUT/pmc_mfma_gemm_shapes.txt 2>> $OUT/rc.log; rm -rf $OUT/pmc_mfma;;
esac; done
cat $OUT/rc.log
