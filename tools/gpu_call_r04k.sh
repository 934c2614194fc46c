#!/bin/bash
# round 4, call k: counter passes (HBM bytes of the cross-attention kernels bf16 / split planes, of the fused attention blocks, of the GEMM
# class over an encoder chunk; matrix-core busy cycles of the GEMM kernels) and rocprofv3 kernel stats of both engines at the driver's command
OUT=gpurun_out/r04k; mkdir -p $OUT; R=$(pwd); export TMPDIR=/tmp
timeout 60 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); torch.cuda.synchronize(); assert float(x.sum()) == float(1 << 20)" > $OUT/canary.log 2>&1 || { echo canary failed; exit 3; }
pmc() {  # name counter command...
  local name=$1 ctr=$2; shift 2
  (cd /tmp && timeout 400 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $R/$OUT/pmc_$name -o pmc -- "$@" > $R/$OUT/pmc_$name.log 2>&1); echo "pmc $name rc=$?" >> $OUT/rc.log
  f=$(find $OUT/pmc_$name -name "*counter_collection.csv" 2>/dev/null | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py $f ${ctr%% *} > $OUT/pmc_$name.txt 2>> $OUT/rc.log
}
for c in FETCH_SIZE WRITE_SIZE; do
  pmc cross160_$c $c python $R/tools/cross_pmc.py 160; rm -rf $OUT/pmc_cross160_$c
  pmc crossx3_160_$c $c python $R/tools/cross_pmc.py 160 split; rm -rf $OUT/pmc_crossx3_160_$c
  KBENCH_SWIN_B=8 pmc swinblock_$c $c python $R/tools/kbench.py swin_block; rm -rf $OUT/pmc_swinblock_$c
  pmc encode_$c $c python $R/tools/encode_pmc.py 32; cp $OUT/pmc_encode_$c.log $OUT/encode_alg_$c.json; rm -rf $OUT/pmc_encode_$c
done
python tools/pmc_cross_json.py $OUT/pmc_cross160_FETCH_SIZE.txt $OUT/pmc_cross160_WRITE_SIZE.txt 160 "rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> --kernel-trace --output-format csv -- python tools/cross_pmc.py 160" profiles/pmc_cross_attn.json > $OUT/pmc_cross_attn_a.json 2>> $OUT/rc.log
python tools/pmc_cross_json.py $OUT/pmc_crossx3_160_FETCH_SIZE.txt $OUT/pmc_crossx3_160_WRITE_SIZE.txt x3_160 "rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> --kernel-trace --output-format csv -- python tools/cross_pmc.py 160 split" $OUT/pmc_cross_attn_a.json > $OUT/pmc_cross_attn.json 2>> $OUT/rc.log
# matrix-core busy cycles of the GEMM kernels on the stage-2 / stage-3 shapes (32-image chunks), bf16 and bf16x3
(cd /tmp && KBENCH_GEMM_VARIANTS=9,10 KBENCH_GEMM_MSCALE=4 KBENCH_GEMM_ONLY=8,9,10,11,12,14,15 timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/$OUT/pmc_mfma -o pmc -- python $R/tools/kbench.py gemm > $R/$OUT/pmc_mfma_kbench.log 2>&1); echo "mfma rc=$?" >> $OUT/rc.log
f=$(find $OUT/pmc_mfma -name "*counter_collection.csv" 2>/dev/null | head -1); [ -n "$f" ] && python tools/pmc_mfma.py $f > $OUT/pmc_mfma_gemm_shapes.txt 2>> $OUT/rc.log; rm -rf $OUT/pmc_mfma
(cd /tmp && KBENCH_GEMM_X3=1 KBENCH_GEMM_VARIANTS=9,10 KBENCH_GEMM_MSCALE=4 KBENCH_GEMM_ONLY=8,10,11,15 timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/$OUT/pmc_mfma3 -o pmc -- python $R/tools/kbench.py gemm > $R/$OUT/pmc_mfma_kbench_x3.log 2>&1); echo "mfma x3 rc=$?" >> $OUT/rc.log
f=$(find $OUT/pmc_mfma3 -name "*counter_collection.csv" 2>/dev/null | head -1); [ -n "$f" ] && python tools/pmc_mfma.py $f > $OUT/pmc_mfma_gemm_shapes_x3.txt 2>> $OUT/rc.log; rm -rf $OUT/pmc_mfma3
# kernel stats, eager launches (graph 0), headline alone / parity engine alone
for eng in bf16 bf16x3; do
  (cd /tmp && timeout 420 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_$eng -o ks -- python $R/bench.py --dtype $eng --steps 20 --warmup 5 --min-seconds 0 --no-parity-leg --no-config-legs --no-batch8 --no-eos-run --no-cpu-baseline --no-roofline --graph 0 > $R/$OUT/prof_bench_$eng.json 2> $R/$OUT/prof_$eng.err); echo "prof $eng rc=$?" >> $OUT/rc.log
  db=$(find $OUT/prof_$eng -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocpd_stats.py $db > $OUT/kernel_stats_$eng.txt 2>> $OUT/rc.log && python tools/rocpd_shapes.py $db 2.0 > $OUT/kernel_shapes_$eng.txt 2>> $OUT/rc.log
  rm -rf $OUT/prof_$eng
done
cat $OUT/rc.log; head -12 $OUT/pmc_crossx3_160_FETCH_SIZE.txt; head -8 $OUT/pmc_swinblock_FETCH_SIZE.txt; head -16 $OUT/pmc_mfma_gemm_shapes.txt
