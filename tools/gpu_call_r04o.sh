#!/bin/bash
# images per encoder pass (OMP355_ENC_CHUNK): does a residual stream that fits the 256 MB MALL pay?  Headline + parity engine.
OUT=gpurun_out/r04o; mkdir -p $OUT; export TMPDIR=/tmp
timeout 60 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); torch.cuda.synchronize(); assert float(x.sum()) == float(1 << 20)" > $OUT/canary.log 2>&1 || { echo canary failed; exit 3; }
for c in 32 8 16 20 40 80; do
  OMP355_ENC_CHUNK=$c timeout 300 python bench.py --steps 20 --warmup 5 --min-seconds 3 --no-batch8 --no-config-legs --no-eos-run --no-cpu-baseline --no-roofline --phase-times > $OUT/b_c${c}.json 2> $OUT/b_c${c}.err
  python - <<P
import json
d=json.loads(open('$OUT/b_c${c}.json').read().strip().splitlines()[-1]); pe=d.get('parity_engine') or {}
print('enc_chunk $c: headline %.1f img/s  parity engine %s' % (d['value'], pe.get('images_per_sec')))
P
  grep "phase ms (one" $OUT/b_c${c}.err | cut -c1-400
done | tee $OUT/summary.txt
