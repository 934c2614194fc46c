#!/bin/bash
# NOTE: selector 10 (persistent 256x256 GEMM) measured here was removed afterwards; kept as the record of the commands
# r02w: chunked q4 cross-attention + persistent 256x256 GEMM (parity + A/B), side workloads of bench.py
OUT=gpurun_out/r02w; mkdir -p $OUT; export TMPDIR=/tmp
timeout 700 python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py -m gpu -q -x -k "gemm or cross or decoder or spot_odd or kie_sroie or batch_equals" > $OUT/tests_sel.log 2>&1; echo "tests_sel rc=$?" >> $OUT/rc.log; tail -3 $OUT/tests_sel.log
KBENCH_CROSS_IMAGES=256 timeout 200 python tools/kbench.py cross128 > $OUT/kbench_cross256.txt 2>&1; echo "cross rc=$?" >> $OUT/rc.log
KBENCH_GEMM_VARIANTS=9,10 KBENCH_GEMM_MSCALE=4 timeout 300 python tools/kbench.py gemm > $OUT/kbench_gemm_persistent.txt 2>&1; echo "gemm rc=$?" >> $OUT/rc.log
timeout 300 python bench.py --workload mgp_str --steps 40 --warmup 5 --min-seconds 2 > $OUT/bench_mgp_str.json 2> $OUT/bench_mgp_str.err; echo "mgp rc=$?" >> $OUT/rc.log
timeout 400 python bench.py --workload kie --steps 8 --warmup 2 --min-seconds 2 > $OUT/bench_kie.json 2> $OUT/bench_kie.err; echo "kie rc=$?" >> $OUT/rc.log
cat $OUT/rc.log; grep -v amdgpu $OUT/kbench_cross256.txt; grep -v amdgpu $OUT/kbench_gemm_persistent.txt; head -c 1500 $OUT/bench_mgp_str.json; echo; tail -3 $OUT/bench_mgp_str.err; head -c 1500 $OUT/bench_kie.json; echo; tail -3 $OUT/bench_kie.err
