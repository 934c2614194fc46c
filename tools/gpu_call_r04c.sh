#!/bin/bash
# round 4, call c: split-plane slabs (parity engine) -- op checks, e2e parity incl. the new bench-shape fixtures, kernel bench, engine bench
OUT=gpurun_out/r04c; mkdir -p $OUT; export TMPDIR=/tmp
timeout 60 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); torch.cuda.synchronize(); assert float(x.sum()) == float(1 << 20)" > $OUT/canary.log 2>&1 || { echo canary failed; exit 3; }
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "cross_attn or check_gemm or decoder_x3" > $OUT/test_ops.log 2>&1; echo "ops rc=$?" >> $OUT/rc.log; tail -25 $OUT/test_ops.log
OMP355_PARITY_REPORT=$OUT/parity_report_x3.json timeout 900 python -m pytest tests/test_gpu_e2e.py -q -k "parity_engine or batch_equals or n40 or n64" > $OUT/test_e2e.log 2>&1; echo "e2e rc=$?" >> $OUT/rc.log; tail -25 $OUT/test_e2e.log
timeout 300 python tools/kbench.py cross_split > $OUT/kbench_cross_split.txt 2>&1; echo "kbench rc=$?" >> $OUT/rc.log; cat $OUT/kbench_cross_split.txt
timeout 600 python bench.py --dtype bf16x3 --steps 20 --warmup 5 --min-seconds 2 --no-parity-leg --no-config-legs --no-batch8 --no-eos-run --no-cpu-baseline --phase-times > $OUT/bench_x3.json 2> $OUT/bench_x3.err; echo "bench rc=$?" >> $OUT/rc.log
tail -3 $OUT/bench_x3.err; python -c "
import json;d=json.loads(open('$OUT/bench_x3.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step']);print(json.dumps(d.get('roofline'))[:600]);print(json.dumps(d.get('roofline_other'))[:900])"
cat $OUT/rc.log
