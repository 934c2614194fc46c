#!/bin/bash
# round 4, fifth closing call (dispatch table exposed as host logic): smoke, the full GPU suite, the driver's bench command
OUT=gpurun_out/r04zc; mkdir -p $OUT; export TMPDIR=/tmp
timeout 60 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); torch.cuda.synchronize(); assert float(x.sum()) == float(1 << 20)" > $OUT/canary.log 2>&1 || { echo canary failed; exit 3; }
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.txt 2>&1; echo "smoke rc=$?" >> $OUT/rc.log; tail -2 $OUT/smoke.txt
OMP355_PARITY_REPORT=$OUT/parity_report.json timeout 1500 python -m pytest tests -m gpu -q > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/rc.log; tail -8 $OUT/tests.log
timeout 900 python bench.py --steps 20 --warmup 5 --phase-times > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" >> $OUT/rc.log
tail -3 $OUT/bench.err; python - <<'P'
import json
d=json.loads(open('gpurun_out/r04zc/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step'], 'batch8', d.get('images_per_sec_batch8'))
for k in ('eos_run','parity_engine','kie','mgp_str','long_pt'):
    v=d.get(k); print(k, {kk:vv for kk,vv in v.items() if kk in ('images_per_sec','words_per_sec','ms_per_step','tokens_per_sec')} if isinstance(v,dict) else v)
print('roofline', {k:d['roofline'].get(k) for k in ('kernel','achieved','frac','traffic','avg_us','frac_of_launch_rooflines')})
for r in d.get('roofline_other',[]): print('other', {k:r.get(k) for k in ('kernel','achieved','frac','avg_us','traffic')})
pe=d.get('parity_engine',{})
if isinstance(pe,dict) and pe.get('roofline'):
    print('parity roofline', {k:pe['roofline'].get(k) for k in ('kernel','achieved','frac','avg_us','traffic')})
    for r in pe.get('roofline_other',[]): print('  other', {k:r.get(k) for k in ('kernel','achieved','frac','avg_us','traffic')})
P
cat $OUT/rc.log
