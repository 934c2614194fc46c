#!/bin/bash
# launch-latency knobs of the HIP runtime on the small-call legs: HIP_FORCE_DEV_KERNARG (kernel arguments staged in device memory), graph kernarg pool
OUT=gpurun_out/r04za; mkdir -p $OUT; export TMPDIR=/tmp
timeout 60 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); torch.cuda.synchronize(); assert float(x.sum()) == float(1 << 20)" > $OUT/canary.log 2>&1 || { echo canary failed; exit 3; }
run() {  # tag env...
  tag=$1; shift
  env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --min-seconds 2 --no-parity-leg --no-config-legs --no-cpu-baseline --no-roofline > $OUT/b_$tag.json 2> $OUT/b_$tag.err
  python - <<P
import json
d=json.loads(open('$OUT/b_$tag.json').read().strip().splitlines()[-1]); b=d['batch8']; e=d.get('eos_run') or {}
print('$tag: headline %.1f  batch8 %.1f  eos_run %s img/s' % (d['value'], b['images_per_sec'], e.get('images_per_sec')))
P
}
for rep in 1 2; do
run default X=1
run dev_kernarg HIP_FORCE_DEV_KERNARG=1
run dev_kernarg0 HIP_FORCE_DEV_KERNARG=0
done | tee $OUT/summary.txt
