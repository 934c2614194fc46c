#!/bin/bash
# does the row stride of A / W (a power of two at K = 512 .. 4096) pace the operand path?  same products with rows padded by 64 / 32 elements
OUT=gpurun_out/r04q; mkdir -p $OUT; export TMPDIR=/tmp
timeout 60 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); torch.cuda.synchronize(); assert float(x.sum()) == float(1 << 20)" > $OUT/canary.log 2>&1 || { echo canary failed; exit 3; }
cat gpurun_out/r04p/probe.txt 2>/dev/null
for pad in 0 64 32; do
  echo "== row strides padded by $pad elements"
  KBENCH_GEMM_PAD=$pad KBENCH_GEMM_VARIANTS=9,16,17 KBENCH_GEMM_MSCALE=4 KBENCH_GEMM_ONLY=8,10,12,14,16 timeout 300 python tools/kbench.py gemm 2>&1 | grep "^gemm"
done | tee $OUT/kbench_gemm_pad.txt
