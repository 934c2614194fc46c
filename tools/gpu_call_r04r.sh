#!/bin/bash
# where a gemm_4w_r tile spends its time (selector 18: s_memtime stamps per workgroup)
OUT=gpurun_out/r04r; mkdir -p $OUT; export TMPDIR=/tmp
timeout 60 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); torch.cuda.synchronize(); assert float(x.sum()) == float(1 << 20)" > $OUT/canary.log 2>&1 || { echo canary failed; exit 3; }
timeout 300 python tools/gemm4wr_trace.py 2>&1 | grep -v amdgpu.ids | tee $OUT/gemm4wr_trace.txt
