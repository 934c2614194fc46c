#!/bin/bash
# r02s: polygon || recognition overlap and lane count at 256 images per engine call; decoder GEMMs at R = 16384 in isolation
OUT=gpurun_out/r02s; mkdir -p $OUT; export TMPDIR=/tmp
timeout 200 python tools/kbench.py dec_gemm selfattn > $OUT/kbench_dec.txt 2>&1; echo "kbench rc=$?" >> $OUT/rc.log
for cfg in "1 1" "1 0" "2 0" "2 1"; do set -- $cfg
  timeout 200 python bench.py --steps 64 --warmup 32 --min-seconds 3 --lanes $1 --overlap $2 --no-cpu-baseline --no-batch8 --no-eos-run --no-roofline --phase-times > $OUT/bench_l$1_o$2.json 2> $OUT/bench_l$1_o$2.err; echo "bench lanes=$1 overlap=$2 rc=$?" >> $OUT/rc.log
  python - <<P >> $OUT/summary.txt
import json
d=json.load(open('$OUT/bench_l$1_o$2.json'))
print('lanes=$1 overlap=$2 : %.1f img/s  %.2f ms/step' % (d['value'], d['ms_per_step']))
P
  grep "phase ms" $OUT/bench_l$1_o$2.err >> $OUT/summary.txt
done
cat $OUT/rc.log $OUT/summary.txt
