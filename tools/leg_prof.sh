# kernel table of one bench workload / leg: rocprofv3 --kernel-trace --stats of `python bench.py <args>`, summarised by grid (tools/rocpd_shapes.py)
#     bash tools/leg_prof.sh <tag> <name> <bench.py arguments...>      -> gpurun_out/<tag>/<name>_kernel_shapes.txt
TAG=$1; NAME=$2; shift 2
OUT=gpurun_out/$TAG; mkdir -p $OUT; R=$PWD; export TMPDIR=/tmp
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_$NAME -o ks -- python $R/bench.py "$@" --min-seconds 0 --no-cpu-baseline --no-roofline > $R/$OUT/${NAME}_bench.json 2> $R/$OUT/${NAME}.err); echo "$NAME rc=$?"
db=$(find $OUT/prof_$NAME -name "*.db" | head -1); python tools/rocpd_shapes.py $db 1.0 > $OUT/${NAME}_kernel_shapes.txt; rm -rf $OUT/prof_$NAME; head -45 $OUT/${NAME}_kernel_shapes.txt | cut -c1-175
