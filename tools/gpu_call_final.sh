#!/bin/bash
OUT=gpurun_out/r01o; mkdir -p $OUT; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 150 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$OUT/pmc_$c -o pmc -- python $R/tools/cross_pmc.py 128 > $R/$OUT/cross_pmc_$c.log 2>&1); echo "pmc $c rc=$?" >> $OUT/rc.log
  f=$(find $OUT/pmc_$c -name "*counter_collection.csv" 2>/dev/null | head -1); [ -n "$f" ] && python tools/pmc_summary.py $f $c > $OUT/pmc_$c.txt 2>> $OUT/rc.log; rm -rf $OUT/pmc_$c
done
python tools/pmc_cross_json.py $OUT/pmc_FETCH_SIZE.txt $OUT/pmc_WRITE_SIZE.txt 128 "rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> --kernel-trace --output-format csv -- python tools/cross_pmc.py 128  (the two cross-attention kernels in isolation at the bench shapes, launch mix 22:10 as in bench.py)" > $OUT/pmc_cross_attn.json 2>> $OUT/rc.log
timeout 200 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/rc.log
cat $OUT/rc.log; tail -2 $OUT/smoke.log
