"""Timeline of consecutive kernel dispatches from a rocprofv3 rocpd database (--kernel-trace):
    python tools/rocpd_timeline.py x_results.db <anchor kernel substring> [occurrence] [count]
prints `count` dispatches starting at the `occurrence`-th dispatch whose name contains the anchor: start offset, duration,
gap to the previous dispatch's end, grid, name -- how a captured decoder step is actually spent (kernel bodies vs gaps)."""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    anchor = sys.argv[2]
    occ = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    count = int(sys.argv[4]) if len(sys.argv) > 4 else 60
    rows = db.execute('select k.start, k.end, s.display_name, k.grid_size_x, k.grid_size_y, k.grid_size_z, k.workgroup_size_x '
                      'from rocpd_kernel_dispatch k join rocpd_info_kernel_symbol s on k.kernel_id = s.id order by k.start').fetchall()
    idx = [i for i, r in enumerate(rows) if anchor in r[2]]
    if len(idx) < occ:
        print('anchor %r found %d times' % (anchor, len(idx)))
        return
    i0 = idx[occ - 1]
    t0 = rows[i0][0]
    prev_end = rows[i0 - 1][1] if i0 > 0 else t0
    tot_k = tot_g = 0.0
    print('%9s %8s %8s  %-14s %s' % ('t_us', 'dur_us', 'gap_us', 'grid/wg', 'kernel'))
    for r in rows[i0:i0 + count]:
        name = re.sub(r'\(anonymous namespace\)::', '', r[2])
        name = re.sub(r'^void ', '', name)[:90]
        gap = (r[0] - prev_end) / 1e3
        dur = (r[1] - r[0]) / 1e3
        tot_k += dur
        tot_g += max(gap, 0.0)
        print('%9.2f %8.2f %8.2f  %-14s %s' % ((r[0] - t0) / 1e3, dur, gap, '%dx%dx%d/%d' % (r[3] // max(r[6], 1), r[4], r[5], r[6]), name))
        prev_end = max(prev_end, r[1])
    print('# %d dispatches: kernel time %.1f us, gaps %.1f us, span %.1f us' % (count, tot_k, tot_g, (prev_end - t0) / 1e3))


if __name__ == '__main__':
    main()
