"""Summarise a rocprofv3 rocpd database (--kernel-trace) into a per-kernel table:
    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [steps] > profiles/rNN_kernel_stats.txt
`steps` (optional) divides the totals so the table reads per bench step."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    return name[:110]


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    cur = db.cursor()
    rows = cur.execute('select s.display_name, count(*), sum(k.end - k.start), min(k.end - k.start), max(k.end - k.start), '
                       'max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(s.sgpr_count), max(k.group_segment_size) '
                       'from rocpd_kernel_dispatch k join rocpd_info_kernel_symbol s on k.kernel_id = s.id '
                       'group by s.display_name order by 3 desc').fetchall()
    tot = sum(r[2] for r in rows)
    span = cur.execute('select min(start), max(end) from rocpd_kernel_dispatch').fetchone()
    print('# %s : %d kernels, %d dispatches, GPU busy %.3f ms, first->last dispatch span %.3f ms%s'
          % (sys.argv[1].split('/')[-1], len(rows), sum(r[1] for r in rows), tot / 1e6, (span[1] - span[0]) / 1e6,
             (' ; per-step columns divide by %g steps' % steps) if steps != 1 else ''))
    print('%-110s %8s %11s %9s %9s %9s %6s %5s %5s %6s' % ('kernel', 'calls', 'total_ms', 'avg_us', 'min_us', 'max_us', '%', 'vgpr', 'agpr', 'lds'))
    for n, c, t, mn, mx, vg, ag, sg, lds in rows:
        print('%-110s %8.1f %11.3f %9.2f %9.2f %9.2f %6.2f %5d %5d %6d'
              % (short(n), c / steps, t / 1e6 / steps, t / c / 1e3, mn / 1e3, mx / 1e3, 100.0 * t / tot, vg or 0, ag or 0, lds or 0))


if __name__ == '__main__':
    main()
