"""Per-kernel table from a rocprofv3 --kernel-trace --output-format csv file: calls, mean / min duration, grid, workgroup, LDS, registers.
    python tools/trace_csv_summary.py <x_kernel_trace.csv> [substring ...]
The kernel NAME of a hipBLASLt (Tensile) product encodes its macro-tile (MT..x..x..), matrix instruction (MI..) and LDS / prefetch
options: this is how `tools/gpu_call.sh libgemm` records what the library runs for a product next to libomp355's kernels."""
import collections
import csv
import sys


def main():
    path, subs = sys.argv[1], sys.argv[2:]
    agg = collections.OrderedDict()
    with open(path) as f:
        rd = csv.DictReader(f)
        for r in rd:
            name = r.get('Kernel_Name', '?')
            if subs and not any(s in name for s in subs):
                continue
            dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
            key = (name, r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('Grid_Size_Y', ''), r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?')))
            a = agg.setdefault(key, dict(n=0, t=0.0, mn=1e30, lds=r.get('LDS_Block_Size', '?'), vgpr=r.get('VGPR_Count', '?'),
                                          agpr=r.get('Accum_VGPR_Count', '?'), sgpr=r.get('SGPR_Count', '?'), scratch=r.get('Scratch_Size', r.get('Private_Segment_Size', '?'))))
            a['n'] += 1
            a['t'] += dur
            a['mn'] = min(a['mn'], dur)
    print('%7s %10s %10s %12s %6s %7s %6s %6s %6s %8s  %s' % ('calls', 'avg_us', 'min_us', 'grid_x', 'wg', 'lds', 'vgpr', 'agpr', 'sgpr', 'scratch', 'kernel'))
    for (name, gx, gy, wg), a in sorted(agg.items(), key=lambda kv: -kv[1]['t']):
        if a['t'] < 50.0 and not subs:
            continue
        print('%7d %10.1f %10.1f %12s %6s %7s %6s %6s %6s %8s  %s' % (a['n'], a['t'] / a['n'], a['mn'], gx + ('x' + gy if gy not in ('', '1') else ''), wg, a['lds'], a['vgpr'], a['agpr'], a['sgpr'], a['scratch'], name[:400]))


if __name__ == '__main__':
    main()
