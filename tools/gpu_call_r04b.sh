#!/bin/bash
OUT=gpurun_out/r04b; mkdir -p $OUT; export TMPDIR=/tmp
timeout 60 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); torch.cuda.synchronize(); assert float(x.sum()) == float(1 << 20)" > $OUT/canary.log 2>&1 || { echo canary failed; exit 3; }
timeout 300 python tools/kbench.py dec_rows > $OUT/kbench_dec_rows.txt 2>&1; echo "rc=$?"; cat $OUT/kbench_dec_rows.txt
KBENCH_DEC_ROWS=32768 timeout 300 python tools/kbench.py dec_rows > $OUT/kbench_dec_rows_32768.txt 2>&1; echo "rc=$?"; cat $OUT/kbench_dec_rows_32768.txt
