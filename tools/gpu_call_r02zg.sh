#!/bin/bash
# r02zg: HBM traffic of the cross-attention kernels at 512 images per launch (the bench default), FETCH_SIZE / WRITE_SIZE passes
OUT=gpurun_out/r02zg; mkdir -p $OUT; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); export TMPDIR=/tmp
I=512
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$OUT/pmc_${c}_$I -o pmc -- python $R/tools/cross_pmc.py $I > $R/$OUT/pmc_${c}_$I.log 2>&1); echo "pmc $c $I rc=$?" >> $OUT/rc.log
  f=$(find $OUT/pmc_${c}_$I -name "*counter_collection.csv" 2>/dev/null | head -1); [ -n "$f" ] && python tools/pmc_summary.py $f $c > $OUT/pmc_${c}_$I.txt 2>> $OUT/rc.log; rm -rf $OUT/pmc_${c}_$I
done
cp profiles/pmc_cross_attn.json $OUT/pmc_cross_attn.json
python tools/pmc_cross_json.py $OUT/pmc_FETCH_SIZE_$I.txt $OUT/pmc_WRITE_SIZE_$I.txt $I "rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> --kernel-trace --output-format csv -- python tools/cross_pmc.py $I" $OUT/pmc_cross_attn.json > $OUT/pmc_cross_attn.json.new 2>> $OUT/rc.log && mv $OUT/pmc_cross_attn.json.new $OUT/pmc_cross_attn.json
cat $OUT/rc.log; head -5 $OUT/pmc_FETCH_SIZE_$I.txt; python -c "
import json; d=json.load(open('$OUT/pmc_cross_attn.json')); print({k:(v['fetch_kib_mean'], v['write_kib_mean']) for k,v in d.items()})"
