"""What ran BESIDE every launch of one kernel (rocprofv3 --kernel-trace rocpd database)?
    python tools/rocpd_overlap.py x_results.db <kernel substring> [min workgroups]
For each dispatch of the anchor kernel: its duration and the fraction of it during which (a) another dispatch of the SAME kernel, (b) any
other kernel was executing.  Printed as a table by overlap class -- the question behind VERDICT r4 item 6 (the 64-row cross-attention
kernel: average 250 us, maximum 461 us at the same grid): a launch that shares the chip with the other decoder stream's launch of the
same HBM-bound kernel takes twice as long without being any slower per byte."""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    anchor = sys.argv[2]
    min_wgs = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    cols = [r[1] for r in db.execute('pragma table_info(rocpd_kernel_dispatch)').fetchall()]
    qcol = next((c for c in ('stream_id', 'queue_id') if c in cols), None)
    rows = db.execute('select k.start, k.end, s.display_name, k.grid_size_x * k.grid_size_y * k.grid_size_z / max(k.workgroup_size_x * k.workgroup_size_y * k.workgroup_size_z, 1)%s '
                      'from rocpd_kernel_dispatch k join rocpd_info_kernel_symbol s on k.kernel_id = s.id order by k.start' % ((', k.' + qcol) if qcol else ', 0')).fetchall()
    n = len(rows)
    anchors = [i for i, r in enumerate(rows) if anchor in r[2] and r[3] >= min_wgs]
    if not anchors:
        print('no dispatch of %r' % anchor)
        return
    # grids of the anchor: report the most frequent big one
    by_grid = {}
    for i in anchors:
        by_grid.setdefault(rows[i][3], []).append(i)
    print('# %s: %d dispatches of %r; queue column: %s' % (sys.argv[1].split('/')[-1], len(anchors), anchor, qcol))
    for grid, idx in sorted(by_grid.items(), key=lambda kv: -len(kv[1]) * kv[0]):
        if len(idx) < 8:
            continue
        recs = []
        for i in idx:
            s0, e0 = rows[i][0], rows[i][1]
            same = other = 0
            beside = set()
            j = i - 1
            while j >= 0 and rows[j][0] > s0 - 5_000_000:   # dispatches that started up to 5 ms earlier may still run
                if rows[j][1] > s0:
                    ov = min(e0, rows[j][1]) - s0
                    if anchor in rows[j][2]:
                        same += ov
                    else:
                        other += ov
                        beside.add(re.sub(r'\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d+', '', rows[j][2])[:28])
                j -= 1
            j = i + 1
            while j < n and rows[j][0] < e0:
                ov = min(e0, rows[j][1]) - rows[j][0]
                if anchor in rows[j][2]:
                    same += ov
                else:
                    other += ov
                    beside.add(re.sub(r'\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d+', '', rows[j][2])[:28])
                j += 1
            d = e0 - s0
            recs.append((d / 1e3, same / d, other / d, rows[i][4], beside))
        print('\n## grid %d workgroups: %d launches, avg %.1f us, min %.1f, max %.1f' % (grid, len(recs), sum(r[0] for r in recs) / len(recs), min(r[0] for r in recs), max(r[0] for r in recs)))
        classes = [('alone (no other kernel for > 90 %% of the launch)', lambda r: r[1] < 0.1 and r[2] < 0.1),
                   ('beside OTHER kernels only', lambda r: r[1] < 0.1 and r[2] >= 0.1),
                   ('beside another launch of the SAME kernel for 10-60 %', lambda r: 0.1 <= r[1] < 0.6),
                   ('beside another launch of the SAME kernel for > 60 %', lambda r: r[1] >= 0.6)]
        print('%-58s %7s %9s %9s %9s %9s' % ('class', 'count', 'avg_us', 'p50_us', 'p99_us', 'max_us'))
        for name, pred in classes:
            ds = sorted(r[0] for r in recs if pred(r))
            if not ds:
                continue
            print('%-58s %7d %9.1f %9.1f %9.1f %9.1f' % (name, len(ds), sum(ds) / len(ds), ds[len(ds) // 2], ds[min(len(ds) - 1, int(0.99 * len(ds)))], ds[-1]))
        worst = sorted(recs, key=lambda r: -r[0])[:6]
        print('slowest launches: ' + '; '.join('%.0f us (same %.0f %%, other %.0f %%: %s)' % (r[0], 100 * r[1], 100 * r[2], ','.join(sorted(r[4]))[:80]) for r in worst))
        qs = sorted(set(r[3] for r in recs))
        print('queues / streams seen: %s' % qs)


if __name__ == '__main__':
    main()
