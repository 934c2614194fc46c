#!/bin/bash
# End-of-round evidence (r02z): smoke, the whole GPU suite, the bench line as the driver runs it, rocprofv3 kernel stats of
# the same command, PMC passes (HBM bytes of the cross-attention kernels at 256 images per launch; MFMA-busy of the GEMMs),
# the config 3 / 5 lines and the kernel micro-benchmarks.
OUT=gpurun_out/r02z; mkdir -p $OUT; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke rc=$?" >> $OUT/rc.log; tail -1 $OUT/smoke.txt
OMP355_PARITY_REPORT=$OUT/parity_report.json timeout 900 python -m pytest tests -m gpu -q > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/rc.log; tail -3 $OUT/tests.log
for I in 256; do for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$OUT/pmc_${c}_$I -o pmc -- python $R/tools/cross_pmc.py $I > $R/$OUT/pmc_${c}_$I.log 2>&1); echo "pmc $c $I rc=$?" >> $OUT/rc.log
  f=$(find $OUT/pmc_${c}_$I -name "*counter_collection.csv" 2>/dev/null | head -1); [ -n "$f" ] && python tools/pmc_summary.py $f $c > $OUT/pmc_${c}_$I.txt 2>> $OUT/rc.log; rm -rf $OUT/pmc_${c}_$I
done
cp profiles/pmc_cross_attn.json $OUT/pmc_cross_attn.json
python tools/pmc_cross_json.py $OUT/pmc_FETCH_SIZE_$I.txt $OUT/pmc_WRITE_SIZE_$I.txt $I "rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> --kernel-trace --output-format csv -- python tools/cross_pmc.py $I" $OUT/pmc_cross_attn.json > $OUT/pmc_cross_attn.json.new 2>> $OUT/rc.log && mv $OUT/pmc_cross_attn.json.new $OUT/pmc_cross_attn.json
done
OMP355_PMC_JSON=$R/$OUT/pmc_cross_attn.json timeout 900 python bench.py --phase-times > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" >> $OUT/rc.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o ks -- python $R/bench.py --steps 64 --warmup 0 --min-seconds 0 --no-cpu-baseline --no-batch8 --no-eos-run > $R/$OUT/prof_bench.json 2> $R/$OUT/prof.err); echo "prof rc=$?" >> $OUT/rc.log
db=$(find $OUT/prof -name "*.db" 2>/dev/null | head -1)
[ -n "$db" ] && python tools/rocpd_stats.py $db > $OUT/kernel_stats.txt 2>> $OUT/prof.err
[ -n "$db" ] && python tools/rocpd_shapes.py $db 2.0 > $OUT/kernel_shapes.txt 2>> $OUT/prof.err
rm -rf $OUT/prof
(cd /tmp && KBENCH_GEMM_VARIANTS=0 timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/$OUT/pmc_mfma -o pmc -- python $R/tools/kbench.py gemm mlp kvproj > $R/$OUT/pmc_mfma_kbench.log 2>&1); echo "mfma rc=$?" >> $OUT/rc.log
f=$(find $OUT/pmc_mfma -name "*counter_collection.csv" 2>/dev/null | head -1); [ -n "$f" ] && python tools/pmc_mfma.py $f > $OUT/pmc_mfma_gemm_shapes.txt 2>> $OUT/rc.log; rm -rf $OUT/pmc_mfma
timeout 300 python bench.py --workload mgp_str --steps 40 --warmup 5 --min-seconds 2 > $OUT/bench_mgp_str.json 2> $OUT/bench_mgp_str.err; echo "mgp rc=$?" >> $OUT/rc.log
timeout 400 python bench.py --workload kie --steps 8 --warmup 2 --min-seconds 2 > $OUT/bench_kie.json 2> $OUT/bench_kie.err; echo "kie rc=$?" >> $OUT/rc.log
KBENCH_GEMM_VARIANTS=0 KBENCH_CROSS_IMAGES=256 timeout 300 python tools/kbench.py gemm mlp kvproj cross128 selfattn > $OUT/kbench_final.txt 2>&1; echo "kbench rc=$?" >> $OUT/rc.log
cat $OUT/rc.log; head -c 600 $OUT/bench.json; echo
