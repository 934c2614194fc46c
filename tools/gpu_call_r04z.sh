#!/bin/bash
# fused three-product kernel dispatched in the parity engine: check_gemm_4w, the parity-engine end-to-end tests (+ the replicated fixture), parity bench leg
OUT=gpurun_out/r04z; mkdir -p $OUT; export TMPDIR=/tmp
timeout 60 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); torch.cuda.synchronize(); assert float(x.sum()) == float(1 << 20)" > $OUT/canary.log 2>&1 || { echo canary failed; exit 3; }
timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "gemm_4w" > $OUT/pytest_gemm_4w.txt 2>&1; echo "pytest ops rc=$?"; tail -3 $OUT/pytest_gemm_4w.txt
timeout 900 python -m pytest tests/test_gpu_e2e.py -q -k "bf16x3 or replicated" > $OUT/pytest_e2e_x3.txt 2>&1; echo "pytest e2e rc=$?"; tail -6 $OUT/pytest_e2e_x3.txt
timeout 600 python bench.py --steps 20 --warmup 5 --min-seconds 3 --no-batch8 --no-config-legs --no-eos-run --no-cpu-baseline --phase-times > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'P'
import json
d=json.loads(open('gpurun_out/r04z/bench.json').read().strip().splitlines()[-1]); pe=d.get('parity_engine') or {}
print('headline %.1f img/s; parity engine %s img/s' % (d['value'], pe.get('images_per_sec')))
for r in [pe.get('roofline')] + (pe.get('roofline_other') or []):
    if r: print('  parity', {k: r.get(k) for k in ('kernel','achieved','frac','avg_us')})
P
