"""Headline figures of a bench.py JSON line (the last line of the file): python tools/bench_summary.py gpurun_out/<tag>/bench.json"""
import json
import sys


def main():
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('value', d.get('value'), 'ms/step', d.get('ms_per_step'), 'batch8', d.get('images_per_sec_batch8'))
    for k in ('eos_run', 'parity_engine', 'kie', 'kie_parity', 'mgp_str', 'mgp_str_parity', 'long_pt'):
        v = d.get(k)
        if v is None:
            continue
        print(k, {kk: vv for kk, vv in v.items() if kk in ('images_per_sec', 'words_per_sec', 'ms_per_step', 'tokens_per_sec', 'error')} if isinstance(v, dict) else v)
    keys = ('kernel', 'achieved', 'frac', 'traffic', 'avg_us', 'frac_of_launch_rooflines', 'gpu_ms_per_image', 'launches')
    r = d.get('roofline') or {}
    print('roofline', {k: r.get(k) for k in keys})
    for c in r.get('classes', []) or []:
        print('  class', {k: c.get(k) for k in keys})
    for r in d.get('roofline_other', []) or []:
        print('other', {k: r.get(k) for k in keys})
        for c in r.get('classes', []) or []:
            print('  class', {k: c.get(k) for k in keys})
    pe = d.get('parity_engine', {})
    if isinstance(pe, dict) and pe.get('roofline'):
        print('parity roofline', {k: pe['roofline'].get(k) for k in keys})
        for r in pe.get('roofline_other', []) or []:
            print('  other', {k: r.get(k) for k in keys})
    cb = d.get('cpu_baseline')
    if isinstance(cb, dict):
        print('cpu_baseline', {k: cb.get(k) for k in ('value', 'cores', 'kind', 'estimated')})


if __name__ == '__main__':
    main()
