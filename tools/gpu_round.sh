#!/bin/bash
# One GPU-box session: parity tests, PMC passes (HBM bytes of the cross-attention kernels), bench line,
# rocprofv3 kernel stats.   usage: tools/gpu_round.sh <tag> [legs]
#   legs: subset of "tests pmc bench prof kbench sweep" (default: tests pmc bench prof)
TAG=${1:-r01x}; LEGS=${2:-"tests bench prof"}
OUT=gpurun_out/$TAG; mkdir -p $OUT
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
export TMPDIR=/tmp
PMC_STEPS=${PMC_STEPS:-16}   # images per launch = 8 x PMC_STEPS (PMC_IMAGES must say the same): 20 = the driver's --steps 20 -> 160
PMC_ARGS="--steps $PMC_STEPS --warmup 0 --lanes 1 --graph 0 --no-cpu-baseline --no-roofline --no-parity-leg --no-config-legs --no-batch8 --no-eos-run --min-seconds 0.5"
PMC_CMD="python bench.py $PMC_ARGS"
# canary: a box whose GPU faults at the first device touch (round 3's last call: profiles/r03q_last_call_box_fault.txt) must not be
# allowed to burn the budget in every leg's timeout
if ! timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); torch.cuda.synchronize(); assert float(x.sum()) == float(1 << 20)" > $OUT/canary.log 2>&1; then
  echo "canary failed: this box's GPU does not work, nothing was run" | tee -a $OUT/rc.log; tail -3 $OUT/canary.log; exit 3
fi
for leg in $LEGS; do case $leg in
tests) timeout 900 python -m pytest tests -m gpu -q > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/rc.log; tail -3 $OUT/tests.log;;
pmc)   for c in FETCH_SIZE WRITE_SIZE; do
         (cd /tmp && timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$OUT/pmc_$c -o pmc -- python $R/bench.py $PMC_ARGS > $R/$OUT/pmc_$c.json 2> $R/$OUT/pmc_$c.err); echo "pmc $c rc=$?" >> $OUT/rc.log
         f=$(find $OUT/pmc_$c -name "*counter_collection.csv" 2>/dev/null | head -1); [ -n "$f" ] && python tools/pmc_summary.py $f $c > $OUT/pmc_$c.txt 2>> $OUT/pmc_$c.err; rm -rf $OUT/pmc_$c
       done
       python tools/pmc_cross_json.py $OUT/pmc_FETCH_SIZE.txt $OUT/pmc_WRITE_SIZE.txt ${PMC_IMAGES:-128} "rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> --kernel-trace --output-format csv -- $PMC_CMD" profiles/pmc_cross_attn.json > $OUT/pmc_cross_attn.json 2>> $OUT/rc.log;;
bench) [ -s $R/$OUT/pmc_cross_attn.json ] && export OMP355_PMC_JSON=$R/$OUT/pmc_cross_attn.json; timeout 900 python bench.py ${BENCH_ARGS:---steps 20 --warmup 5} --phase-times > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" >> $OUT/rc.log;;
prof)  (cd /tmp && timeout 420 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o ks -- python $R/bench.py ${PROF_ARGS:---steps 20 --warmup 5} --no-cpu-baseline --no-config-legs > $R/$OUT/prof_bench.json 2> $R/$OUT/prof.err); prc=$?; echo "prof rc=$prc" >> $OUT/rc.log
       if [ $prc -ne 0 ]; then   # rocprofv3 has crashed inside hipGraphLaunch tracing once: same command with eager launches
         rm -rf $OUT/prof; (cd /tmp && timeout 420 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o ks -- python $R/bench.py ${PROF_ARGS:---steps 20 --warmup 5} --no-cpu-baseline --no-config-legs --graph 0 > $R/$OUT/prof_bench.json 2> $R/$OUT/prof_graph0.err); echo "prof(graph 0) rc=$?" >> $OUT/rc.log
       fi
       db=$(find $OUT/prof -name "*.db" 2>/dev/null | head -1); [ -n "$db" ] && python tools/rocpd_stats.py $db > $OUT/kernel_stats.txt 2>> $OUT/prof.err
       find $OUT/prof -name "*.db" -size +20M -delete 2>/dev/null;;
kbench) timeout 600 python tools/kbench.py ${KBENCH_WHAT:-all} > $OUT/kbench.txt 2>&1; echo "kbench rc=$?" >> $OUT/rc.log;;
mgp)   timeout 300 python tools/mgp_bench.py 512 3 > $OUT/mgp_bench.txt 2>&1; echo "mgp rc=$?" >> $OUT/rc.log;;
mfma)  (cd /tmp && KBENCH_GEMM_VARIANTS=5 timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/$OUT/pmc_mfma -o pmc -- python $R/tools/kbench.py gemm > $R/$OUT/pmc_mfma_kbench.log 2>&1); echo "mfma rc=$?" >> $OUT/rc.log
       f=$(find $OUT/pmc_mfma -name "*counter_collection.csv" 2>/dev/null | head -1); [ -n "$f" ] && python tools/pmc_mfma.py $f > $OUT/pmc_mfma_gemm_shapes.txt 2>> $OUT/rc.log; rm -rf $OUT/pmc_mfma
       (cd /tmp && timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/$OUT/pmc_mfma2 -o pmc -- python $R/bench.py --steps 16 --warmup 0 --lanes 1 --graph 0 --no-cpu-baseline --no-roofline > $R/$OUT/pmc_mfma_bench.json 2> $R/$OUT/pmc_mfma_bench.err); echo "mfma2 rc=$?" >> $OUT/rc.log
       f=$(find $OUT/pmc_mfma2 -name "*counter_collection.csv" 2>/dev/null | head -1); [ -n "$f" ] && python tools/pmc_mfma.py $f 5 > $OUT/pmc_mfma_bench.txt 2>> $OUT/rc.log; rm -rf $OUT/pmc_mfma2;;
sweep) timeout 300 python tools/lane_sweep.py ${SWEEP_ARGS:---lanes 2,3 --batches 32 --steps 32} > $OUT/sweep.log 2>&1; echo "sweep rc=$?" >> $OUT/rc.log;;
esac; done
cat $OUT/rc.log
