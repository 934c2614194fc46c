"""Per-(kernel, grid) table from a rocprofv3 rocpd database (--kernel-trace): the same kernel launched at different
problem sizes (encoder chunk vs decoder phase) is listed per launch geometry.
    python tools/rocpd_shapes.py gpurun_out/.../ks_results.db [min_total_ms] > profiles/rNN_kernel_shapes.txt"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'_ZN12_GLOBAL__N_1\d+', '', name)
    name = re.sub(r'^void ', '', name)
    return name[:72]


def main():
    db = sqlite3.connect(sys.argv[1])
    floor = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    cur = db.cursor()
    cols = [r[1] for r in cur.execute('pragma table_info(rocpd_kernel_dispatch)').fetchall()]
    gx = next((c for c in ('grid_size_x', 'grid_x', 'grid_size') if c in cols), None)
    gy = next((c for c in ('grid_size_y', 'grid_y') if c in cols), None)
    gz = next((c for c in ('grid_size_z', 'grid_z') if c in cols), None)
    wx = next((c for c in ('workgroup_size_x', 'workgroup_x', 'workgroup_size') if c in cols), None)
    if gx is None:
        print('# no grid column in rocpd_kernel_dispatch; columns:', cols)
        return
    g = 'k.%s' % gx + (' * k.%s' % gy if gy else '') + (' * k.%s' % gz if gz else '')
    w = ('k.%s' % wx) if wx else '1'
    rows = cur.execute('select s.display_name, %s as threads, %s as wg, count(*), sum(k.end - k.start), min(k.end - k.start), '
                       'max(k.end - k.start) from rocpd_kernel_dispatch k join rocpd_info_kernel_symbol s '
                       'on k.kernel_id = s.id group by s.display_name, threads, wg order by 5 desc' % (g, w)).fetchall()
    tot = sum(r[4] for r in rows)
    print('# %s : GPU busy %.3f ms; rows below %.1f ms are folded into "(rest)"' % (sys.argv[1].split('/')[-1], tot / 1e6, floor))
    print('%-72s %9s %5s %7s %10s %9s %9s %9s %6s' % ('kernel', 'wgs', 'wg', 'calls', 'total_ms', 'avg_us', 'min_us', 'max_us', '%'))
    rest = 0
    for n, threads, wg, c, t, mn, mx in rows:
        if t / 1e6 < floor:
            rest += t
            continue
        print('%-72s %9d %5d %7d %10.3f %9.2f %9.2f %9.2f %6.2f'
              % (short(n), (threads or 0) // max(wg or 1, 1), wg or 0, c, t / 1e6, t / c / 1e3, mn / 1e3, mx / 1e3, 100.0 * t / tot))
    print('%-72s %9s %5s %7s %10.3f %9s %9s %9s %6.2f' % ('(rest)', '', '', '', rest / 1e6, '', '', '', 100.0 * rest / tot))


if __name__ == '__main__':
    main()
