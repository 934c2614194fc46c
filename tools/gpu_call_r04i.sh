#!/bin/bash
OUT=gpurun_out/r04i; mkdir -p $OUT; export TMPDIR=/tmp
timeout 60 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); torch.cuda.synchronize(); assert float(x.sum()) == float(1 << 20)" > $OUT/canary.log 2>&1 || { echo canary failed; exit 3; }
KBENCH_CROSS_IMAGES=160 timeout 300 python tools/kbench.py cross128 > $OUT/kbench_cross160.txt 2>&1; grep "rows/img=64" $OUT/kbench_cross160.txt
timeout 300 python tools/kbench.py cross_split > $OUT/kbench_cross_split.txt 2>&1; grep "rows/img=64" $OUT/kbench_cross_split.txt | grep split
