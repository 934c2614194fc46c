#!/bin/bash
# ONE parametrised GPU-box session (replaces the one-shot tools/gpu_call_r04*.sh scripts of round 4):
#     gpurun --timeout 900 -- 'bash tools/gpu_call.sh <tag> <leg> [<leg> ...]'
# Every leg writes under gpurun_out/<tag>/ and appends "<leg> rc=<n>" to rc.log; the summaries worth judging are copied into profiles/ by hand.
# Legs (environment knobs in brackets):
#   smoke            __graft_entry__.smoke()
#   tests            the whole GPU suite                 [TESTS_K: a pytest -k expression; TESTS_TIMEOUT]
#   bench            the driver's bench command          [BENCH_ARGS, default "--steps 20 --warmup 5 --phase-times"]
#   prof:<engine>    rocprofv3 --kernel-trace --stats of the bench command, eager launches, one engine (bf16 | bf16x3) -> kernel_stats / kernel_shapes
#   probe            tools/build/probe_stream (weight-stream rate of one compute unit, csrc/dec_rows.hip's inner loop alone)
#   libgemm          hipBLASLt next to gemm_256 / gemm_4w_p on the K >= 1024 products: kernel trace (name, grid, LDS, registers) + SQ / TCP counters
#   q4tail           per-launch durations of the 64-row cross-attention kernel by what ran beside it (tools/rocpd_overlap.py)
#   kbench:<what>    tools/kbench.py <what>              [KBENCH_* knobs of that tool]
#   pmcdec           FETCH_SIZE / WRITE_SIZE passes over the decoders' many-row kernels (tools/dec_rows_pmc.py) -> profiles/pmc_dec_rows.json
#   pmcenc           FETCH_SIZE / WRITE_SIZE passes over one 80-image encoder chunk + its K / V^T projection (tools/encode_pmc.py) -> pmc_gemm.json   [PMC_ENC_ARGS: "<images> <engine>"]
#   pmccross         FETCH_SIZE / WRITE_SIZE passes over the cross-attention kernels at PMC_IMAGES images per launch -> pmc_cross_attn.json
#   ab:<VAR>         the headline alone (no side legs, no CPU baseline) with VAR=0 and VAR=1 in the environment, phase times on stderr   [AB_ARGS, AB_VALUES]
#   timeline         kernel trace of the headline WITH graph replay, then tools/rocpd_timeline.py at TL_ANCHOR (kernel substring), occurrence TL_OCC, TL_COUNT dispatches   [TL_ARGS: bench arguments]
#   rccl1            bench.py --rccl-selftest under torch.distributed.run with one rank: RCCL rendezvous, device check, the packed all-gather per engine call, barriers   [RCCL1_ARGS]
#   lanes            tests/test_gpu_e2e.py::test_pipelined_lanes_match_direct in LANES_REPS fresh processes (graph capture under another thread's event waits)
#   py:<script>      python <script> (stdout -> <script basename>.txt)
TAG=${1:?tag}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
if ! timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); torch.cuda.synchronize(); assert float(x.sum()) == float(1 << 20)" > $OUT/canary.log 2>&1; then
  echo "canary failed: this box's GPU does not work, nothing was run" | tee -a $OUT/rc.log; tail -3 $OUT/canary.log; exit 3
fi
counters() {   # once per call: the counter names this rocprofv3 knows
  [ -s $OUT/counter_names.txt ] && return
  (cd /tmp && timeout 120 rocprofv3 -L > $R/$OUT/counters_avail.txt 2>&1)
  grep -o -E "\b(TCP|TCC|TA|TD|SQ|GRBM|TCA)_[A-Za-z0-9_]+" $OUT/counters_avail.txt | sort -u > $OUT/counter_names.txt
}
have() { for c in "$@"; do grep -qx "$c" $OUT/counter_names.txt && printf "%s " "$c"; done; }
pmc_pass() {   # <name> <kernel substrings, comma separated> <command...> -- counters...   (one counter group per pass; only --kernel-trace beside --pmc)
  name=$1; subs=$2; shift 2; cmd=(); while [ "$1" != "--" ]; do cmd+=("$1"); shift; done; shift
  ctrs=$(have "$@"); [ -z "$ctrs" ] && { echo "pass $name: none of [$*] exist" >> $OUT/rc.log; return; }
  (cd /tmp && timeout 400 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $R/$OUT/p_$name -o pmc -- "${cmd[@]}" > $R/$OUT/p_$name.log 2>&1); echo "pass $name [$ctrs] rc=$?" >> $OUT/rc.log
  f=$(find $OUT/p_$name -name "*counter_collection.csv" 2>/dev/null | head -1)
  [ -n "$f" ] && python tools/pmc_multi.py $f ${subs//,/ } > $OUT/pmc_$name.txt; rm -rf $OUT/p_$name
}
for leg in "$@"; do case $leg in
smoke)  timeout 300 python __graft_entry__.py smoke > $OUT/smoke.txt 2>&1; echo "smoke rc=$?" >> $OUT/rc.log; tail -2 $OUT/smoke.txt;;
tests)  OMP355_PARITY_REPORT=$OUT/parity_report.json timeout ${TESTS_TIMEOUT:-1500} python -m pytest tests -m gpu -q ${TESTS_K:+-k "$TESTS_K"} > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/rc.log; tail -8 $OUT/tests.log;;
bench)  timeout 900 python bench.py ${BENCH_ARGS:---steps 20 --warmup 5 --phase-times} > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" >> $OUT/rc.log
        tail -3 $OUT/bench.err; python tools/bench_summary.py $OUT/bench.json;;
prof:*) eng=${leg#prof:}
        (cd /tmp && timeout 420 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_$eng -o ks -- python $R/bench.py --dtype $eng ${PROF_ARGS:---steps 20 --warmup 5} --min-seconds 0 --no-parity-leg --no-config-legs --no-batch8 --no-eos-run --no-cpu-baseline --no-roofline --graph 0 > $R/$OUT/prof_bench_$eng.json 2> $R/$OUT/prof_$eng.err); echo "prof $eng rc=$?" >> $OUT/rc.log
        db=$(find $OUT/prof_$eng -name "*.db" | head -1)
        [ -n "$db" ] && python tools/rocpd_stats.py $db > $OUT/kernel_stats_$eng.txt 2>> $OUT/rc.log && python tools/rocpd_shapes.py $db 2.0 > $OUT/kernel_shapes_$eng.txt 2>> $OUT/rc.log
        [ -n "$db" ] && python tools/rocpd_overlap.py $db dec_cross_attn_q4 > $OUT/q4_overlap_$eng.txt 2>> $OUT/rc.log
        rm -rf $OUT/prof_$eng; head -24 $OUT/kernel_stats_$eng.txt | cut -c1-200;;
probe)  timeout 300 tools/build/probe_stream > $OUT/probe_stream.txt 2>&1; echo "probe rc=$?" >> $OUT/rc.log; cat $OUT/probe_stream.txt;;
libgemm) counters
        SH="131072,512,2048,0,0;32768,4096,1024,0,0;32768,1024,4096,0,0;131072,1536,512,0,0"
        (cd /tmp && KBENCH_GEMM_F32RES=0 KBENCH_GEMM_LIB=1 KBENCH_GEMM_SHAPES="$SH" KBENCH_GEMM_VARIANTS=9,20 timeout 400 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/p_libtrace -o kt -- python $R/tools/kbench.py gemm > $R/$OUT/libgemm_kbench.txt 2>&1); echo "libgemm trace rc=$?" >> $OUT/rc.log
        f=$(find $OUT/p_libtrace -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/trace_csv_summary.py $f > $OUT/libgemm_kernels.txt; rm -rf $OUT/p_libtrace
        for grp in "sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
                   "tcp1 TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
                   "tcp3 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_ACCESSES_sum TCP_UTCL1_TRANSLATION_MISS_sum" \
                   "sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU"; do
          set -- $grp; n=$1; shift
          KBENCH_GEMM_F32RES=0 KBENCH_GEMM_LIB=1 KBENCH_GEMM_SHAPES="$SH" KBENCH_GEMM_VARIANTS=9,20 pmc_pass lib_$n "Cijk,gemm_256,gemm_4w_p" python $R/tools/kbench.py gemm -- "$@"
        done
        cat $OUT/libgemm_kbench.txt | grep gemm; cat $OUT/libgemm_kernels.txt | cut -c1-260;;
q4tail) (cd /tmp && timeout 420 rocprofv3 --kernel-trace -d $R/$OUT/p_q4 -o kt -- python $R/bench.py --steps 20 --warmup 0 --min-seconds 0 --no-parity-leg --no-config-legs --no-batch8 --no-eos-run --no-cpu-baseline --no-roofline --graph 0 > $R/$OUT/q4_bench.json 2> $R/$OUT/q4.err); echo "q4tail rc=$?" >> $OUT/rc.log
        db=$(find $OUT/p_q4 -name "*.db" | head -1); [ -n "$db" ] && python tools/rocpd_overlap.py $db dec_cross_attn_q4 > $OUT/q4_overlap.txt 2>> $OUT/rc.log; rm -rf $OUT/p_q4; cat $OUT/q4_overlap.txt;;
kbench:*) what=${leg#kbench:}; timeout 600 python tools/kbench.py $what > $OUT/kbench_$what.txt 2>&1; echo "kbench $what rc=$?" >> $OUT/rc.log; grep -v amdgpu.ids $OUT/kbench_$what.txt | tail -60;;
pmcdec) for c in FETCH_SIZE WRITE_SIZE; do
          (cd /tmp && timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$OUT/pd_$c -o pmc -- python $R/tools/dec_rows_pmc.py run > $R/$OUT/pmcdec_$c.log 2>&1); echo "pmcdec $c rc=$?" >> $OUT/rc.log
          f=$(find $OUT/pd_$c -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $OUT/pmcdec_$c.csv; rm -rf $OUT/pd_$c
        done
        python tools/dec_rows_pmc.py summarise $OUT/pmcdec_FETCH_SIZE.csv $OUT/pmcdec_WRITE_SIZE.csv > $OUT/pmc_dec_rows.json 2>> $OUT/rc.log; cat $OUT/pmc_dec_rows.json; rm -f $OUT/pmcdec_*.csv;;
pmcenc) for c in FETCH_SIZE WRITE_SIZE; do
          (cd /tmp && timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$OUT/pe_$c -o pmc -- python $R/tools/encode_pmc.py run ${PMC_ENC_ARGS:-80 bf16} > $R/$OUT/pmcenc_$c.json 2> $R/$OUT/pmcenc_$c.err); echo "pmcenc $c rc=$?" >> $OUT/rc.log
          f=$(find $OUT/pe_$c -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $OUT/pmcenc_$c.csv; rm -rf $OUT/pe_$c
        done
        tail -1 $OUT/pmcenc_FETCH_SIZE.json > $OUT/encode_alg.json
        python tools/encode_pmc.py summarise $OUT/pmcenc_FETCH_SIZE.csv $OUT/pmcenc_WRITE_SIZE.csv $OUT/encode_alg.json > $OUT/pmc_gemm.json 2>> $OUT/rc.log; python -c "import json; print(json.dumps(json.load(open('$OUT/pmc_gemm.json'))['summary'], indent=1))"; rm -f $OUT/pmcenc_*.csv;;
pmccross) PMC_STEPS=${PMC_STEPS:-20}
        PMC_ARGS="--steps $PMC_STEPS --warmup 0 --lanes 1 --graph 0 --no-cpu-baseline --no-roofline --no-parity-leg --no-config-legs --no-batch8 --no-eos-run --min-seconds 0.5"
        for c in FETCH_SIZE WRITE_SIZE; do
          (cd /tmp && timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$OUT/pmc_$c -o pmc -- python $R/bench.py $PMC_ARGS > $R/$OUT/pmc_$c.json 2> $R/$OUT/pmc_$c.err); echo "pmc $c rc=$?" >> $OUT/rc.log
          f=$(find $OUT/pmc_$c -name "*counter_collection.csv" 2>/dev/null | head -1); [ -n "$f" ] && python tools/pmc_summary.py $f $c > $OUT/pmc_$c.txt 2>> $OUT/pmc_$c.err; rm -rf $OUT/pmc_$c
        done
        python tools/pmc_cross_json.py $OUT/pmc_FETCH_SIZE.txt $OUT/pmc_WRITE_SIZE.txt $((8 * PMC_STEPS)) "rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> --kernel-trace --output-format csv -- python bench.py $PMC_ARGS" profiles/pmc_cross_attn.json > $OUT/pmc_cross_attn.json 2>> $OUT/rc.log;;
ab:*)   var=${leg#ab:}
        for v in ${AB_VALUES:-0 1}; do
          env $var=$v timeout 600 python bench.py --steps 20 --warmup 5 --phase-times --no-parity-leg --no-config-legs --no-batch8 --no-eos-run --no-cpu-baseline ${AB_ARGS} > $OUT/ab_${var}_$v.json 2> $OUT/ab_${var}_$v.err; echo "ab $var=$v rc=$?" >> $OUT/rc.log
          echo "== $var=$v"; grep "phase ms" $OUT/ab_${var}_$v.err | cut -c1-400; python tools/bench_summary.py $OUT/ab_${var}_$v.json | head -12
        done;;
timeline) (cd /tmp && timeout 420 rocprofv3 --kernel-trace -d $R/$OUT/p_tl -o kt -- python $R/bench.py --steps 20 --warmup 0 --min-seconds 0 --no-parity-leg --no-config-legs --no-batch8 --no-eos-run --no-cpu-baseline --no-roofline ${TL_ARGS} > $R/$OUT/tl_bench.json 2> $R/$OUT/tl.err); echo "timeline rc=$?" >> $OUT/rc.log
        db=$(find $OUT/p_tl -name "*.db" | head -1)
        [ -n "$db" ] && python tools/rocpd_timeline.py $db "${TL_ANCHOR:-dec_embed_ln_kernel}" ${TL_OCC:-300} ${TL_COUNT:-60} > $OUT/timeline.txt 2>> $OUT/rc.log
        [ -n "$db" ] && python tools/rocpd_timeline.py $db "dec_rows_ffn_kernel<5, 1, 0, 0>" ${TL_OCC2:-40} 44 > $OUT/timeline_polyrec.txt 2>> $OUT/rc.log
        rm -rf $OUT/p_tl; cat $OUT/timeline.txt | cut -c1-150;;
rccl1)  # the N > 1 protocol over RCCL with the ONE rank a 1-GPU box allows, launched exactly as the driver launches N > 1
        timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 5 --rccl-selftest \
          --no-parity-leg --no-config-legs --no-batch8 --no-eos-run --no-cpu-baseline ${RCCL1_ARGS} > $OUT/rccl1_bench.json 2> $OUT/rccl1.err; echo "rccl1 rc=$?" >> $OUT/rc.log
        tail -3 $OUT/rccl1.err | cut -c1-300; python -c "import json; d=json.loads(open('$OUT/rccl1_bench.json').read().strip().splitlines()[-1]); print({k: d.get(k) for k in ('value','ms_per_step','n_gpus','backend','ranks','per_rank_ms_per_step','all_gather_ms')})";;
lanes)  # the cross-thread capture race shows once in ~4-9 processes: LANES_REPS fresh processes of the pipelined-lane tests
        for i in $(seq ${LANES_REPS:-10}); do timeout 300 python -m pytest tests -m gpu -q -k "pipelined_lanes_match_direct" 2>&1 | tail -1; done > $OUT/lanes_loop.txt
        echo "lanes rc=$(grep -c failed $OUT/lanes_loop.txt) (processes with a failure, of ${LANES_REPS:-10})" >> $OUT/rc.log; cat $OUT/lanes_loop.txt;;
py:*)   sc=${leg#py:}; timeout 600 python $sc > $OUT/$(basename $sc .py).txt 2>&1; echo "py $sc rc=$?" >> $OUT/rc.log; tail -40 $OUT/$(basename $sc .py).txt;;
*)      echo "unknown leg $leg" | tee -a $OUT/rc.log;;
esac; done
cat $OUT/rc.log
