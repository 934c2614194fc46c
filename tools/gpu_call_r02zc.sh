#!/bin/bash
# r02zc: HBM traffic of the large GEMMs (FETCH_SIZE / WRITE_SIZE PMC passes over one encoder chunk + K / V^T projection)
OUT=gpurun_out/r02zc; mkdir -p $OUT; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$OUT/pmc_$c -o pmc -- python $R/tools/encode_pmc.py 32 > $R/$OUT/encode_alg_$c.json 2> $R/$OUT/encode_$c.err); echo "pmc $c rc=$?" >> $OUT/rc.log
  f=$(find $OUT/pmc_$c -name "*counter_collection.csv" 2>/dev/null | head -1); [ -n "$f" ] && python tools/pmc_summary.py $f $c > $OUT/pmc_encode_$c.txt 2>> $OUT/rc.log; rm -rf $OUT/pmc_$c
done
tail -1 $OUT/encode_alg_FETCH_SIZE.json > $OUT/encode_alg.json
python tools/pmc_gemm_json.py $OUT/pmc_encode_FETCH_SIZE.txt $OUT/pmc_encode_WRITE_SIZE.txt $OUT/encode_alg.json > $OUT/pmc_gemm.json 2>> $OUT/rc.log
cat $OUT/rc.log; cat $OUT/pmc_gemm.json; head -12 $OUT/pmc_encode_FETCH_SIZE.txt
