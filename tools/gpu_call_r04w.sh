#!/bin/bash
# which unit is busy while a 256x256 tile waits for operands?  SQ / TCC / TCP / TA counters of gemm_256 (k9) and gemm_4w_p (k20) on two products
OUT=gpurun_out/r04w; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
timeout 60 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); torch.cuda.synchronize(); assert float(x.sum()) == float(1 << 20)" > $OUT/canary.log 2>&1 || { echo canary failed; exit 3; }
(cd /tmp && timeout 120 rocprofv3 -L > $R/$OUT/counters_avail.txt 2>&1)
grep -o -E "\b(TCP|TCC|TA|TD|SQ|GRBM|TCA)_[A-Za-z0-9_]+" $OUT/counters_avail.txt | sort -u > $OUT/counter_names.txt; wc -l $OUT/counter_names.txt
have() { for c in "$@"; do grep -qx "$c" $OUT/counter_names.txt && printf "%s " "$c"; done; }
pass() {  # name counters...
  name=$1; shift; ctrs=$(have "$@"); [ -z "$ctrs" ] && { echo "pass $name: none of [$*] exist"; return; }
  (cd /tmp && KBENCH_GEMM_F32RES=0 KBENCH_GEMM_SHAPES="131072,1536,512,0,0;32768,1024,4096,0,0" KBENCH_GEMM_VARIANTS=9,20 timeout 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $R/$OUT/p_$name -o pmc -- python $R/tools/kbench.py gemm > $R/$OUT/p_$name.log 2>&1); echo "pass $name [$ctrs] rc=$?"
  f=$(find $OUT/p_$name -name "*counter_collection.csv" 2>/dev/null | head -1)
  [ -n "$f" ] && python tools/pmc_multi.py $f gemm_256 gemm_4w_p > $OUT/pmc_$name.txt; rm -rf $OUT/p_$name
}
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
pass sq2 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT
pass tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
pass tcc2 TCC_BUSY_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
pass tcp1 TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
pass tcp2 TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
pass tcp3 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_ACCESSES_sum TCP_UTCL1_TRANSLATION_MISS_sum
pass ta1 TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum
for f in $OUT/pmc_*.txt; do echo "== $f"; cat $f; done
grep -E "^(TCP|TA)_" $OUT/counter_names.txt | tr '\n' ' ' | fold -w 220 | head -30
