#!/bin/bash
# round 4, call g: full GPU suite on the current tree, patch-embed kernel bench, the driver's bench command
OUT=gpurun_out/r04g; mkdir -p $OUT; export TMPDIR=/tmp
timeout 60 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); torch.cuda.synchronize(); assert float(x.sum()) == float(1 << 20)" > $OUT/canary.log 2>&1 || { echo canary failed; exit 3; }
#timeout 200 python tools/kbench.py patch_embed > $OUT/kbench_patch_embed.txt 2>&1; cat $OUT/kbench_patch_embed.txt
OMP355_PARITY_REPORT=$OUT/parity_report.json timeout 1500 python -m pytest tests -m gpu -q > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/rc.log; tail -30 $OUT/tests.log
#timeout 900 python bench.py --steps 20 --warmup 5 --phase-times > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" >> $OUT/rc.log
cat $OUT/rc.log
