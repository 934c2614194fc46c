#!/bin/bash
# r02ze: lanes 1 vs 2 on the final tree; 512 images per engine call with a balanced step count
OUT=gpurun_out/r02ze; mkdir -p $OUT; export TMPDIR=/tmp
run() { timeout 300 python bench.py --no-cpu-baseline --no-batch8 --no-eos-run --no-roofline --min-seconds 4 "$@" > $OUT/b.json 2> $OUT/b.err; python -c "
import json,sys; d=json.load(open('$OUT/b.json')); print('%-44s : %.1f img/s  %.2f ms/step  (p10 %.2f p90 %.2f, %d reps), call median %.0f ms' % (' '.join(sys.argv[1:]), d['value'], d['ms_per_step'], d['timing']['ms_per_step_p10'], d['timing']['ms_per_step_p90'], d['timing']['repeats'], d['engine_call_ms']['median']))" "$@" >> $OUT/summary.txt; }
run --lanes 1 --steps 192 --warmup 32
run --lanes 2 --steps 192 --warmup 32
run --lanes 2 --steps 256 --warmup 64 --coalesce 64
run --lanes 1 --steps 256 --warmup 64 --coalesce 64
cat $OUT/summary.txt
