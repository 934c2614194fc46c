#!/bin/bash
# r02zd: the bench line with per-launch roofline fractions and GEMM-class traffic (ABI 12), after a quick GPU regression
OUT=gpurun_out/r02zd; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py -m gpu -q -x -k "gemm or mlp_fused or contexts or spot_odd or batch_equals" > $OUT/tests_sel.log 2>&1; echo "tests_sel rc=$?" >> $OUT/rc.log; tail -2 $OUT/tests_sel.log
timeout 900 python bench.py --phase-times > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" >> $OUT/rc.log
cat $OUT/rc.log; python - <<P
import json
d=json.load(open('$OUT/bench.json'))
print(d['value'], d['ms_per_step'], d['batch8'], d['eos_run']['images_per_sec'])
print(json.dumps(d['roofline'])[:900])
for r in d['roofline_other']: print(json.dumps(r)[:500])
P
