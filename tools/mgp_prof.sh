# kernel table of the MGP-STR leg (BASELINE config 5): rocprofv3 --kernel-trace --stats of bench.py --workload mgp_str, summarised by grid (profiles/r06j_*)
OUT=gpurun_out/r06j; mkdir -p $OUT; R=$PWD; export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_mgp -o ks -- python $R/bench.py --workload mgp_str --steps 4 --warmup 2 --min-seconds 0 --no-cpu-baseline --no-roofline > $R/$OUT/mgp_bench.json 2> $R/$OUT/mgp.err); echo rc=$?
db=$(find $OUT/prof_mgp -name "*.db" | head -1); python tools/rocpd_shapes.py $db 0.5 > $OUT/mgp_kernel_shapes.txt; rm -rf $OUT/prof_mgp; head -40 $OUT/mgp_kernel_shapes.txt | cut -c1-190
