#!/bin/bash
OUT=gpurun_out/r04j; mkdir -p $OUT; export TMPDIR=/tmp
timeout 60 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); torch.cuda.synchronize(); assert float(x.sum()) == float(1 << 20)" > $OUT/canary.log 2>&1 || { echo canary failed; exit 3; }
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "window_attn or cross_attn" > $OUT/test_ops.log 2>&1; echo "ops rc=$?" >> $OUT/rc.log; tail -12 $OUT/test_ops.log
OMP355_PARITY_REPORT=$OUT/parity_report_x3.json timeout 900 python -m pytest tests/test_gpu_e2e.py -q -k "parity_engine or swin_t or batch_equals" > $OUT/test_e2e.log 2>&1; echo "e2e rc=$?" >> $OUT/rc.log; tail -12 $OUT/test_e2e.log
timeout 600 python bench.py --dtype bf16x3 --steps 20 --warmup 5 --min-seconds 2 --no-parity-leg --no-config-legs --no-batch8 --no-eos-run --no-cpu-baseline --phase-times > $OUT/bench_x3.json 2> $OUT/bench_x3.err; echo "bench rc=$?" >> $OUT/rc.log
tail -2 $OUT/bench_x3.err; python -c "
import json;d=json.loads(open('$OUT/bench_x3.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step']);print(json.dumps({k:d['roofline'][k] for k in ('kernel','achieved','frac','avg_us')}));print([ {k:r.get(k) for k in ('kernel','achieved','frac','avg_us')} for r in d.get('roofline_other',[])])"
cat $OUT/rc.log
