#!/bin/bash
# hardware queues x lanes for the 8-image-call legs (GPU_MAX_HW_QUEUES is read by the HIP runtime at initialisation)
OUT=gpurun_out/r04n; mkdir -p $OUT; export TMPDIR=/tmp
timeout 60 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); torch.cuda.synchronize(); assert float(x.sum()) == float(1 << 20)" > $OUT/canary.log 2>&1 || { echo canary failed; exit 3; }
for q in 4 8 16; do for l in 4 6 8 12; do
  GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --steps 20 --warmup 5 --min-seconds 1 --no-parity-leg --no-config-legs --no-eos-run --no-cpu-baseline --no-roofline --batch8-lanes $l > $OUT/b_q${q}_l${l}.json 2> $OUT/b_q${q}_l${l}.err
  python - <<P
import json
d=json.loads(open('$OUT/b_q${q}_l${l}.json').read().strip().splitlines()[-1]); b=d['batch8']
print('queues $q lanes $l: headline %.1f  batch8 %.1f img/s (p10-p90 %.1f-%.1f ms per step)' % (d['value'], b['images_per_sec'], b['ms_per_step_p10'], b['ms_per_step_p90']))
P
done; done | tee $OUT/summary.txt
