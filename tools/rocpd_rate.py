"""Dispatch rate and concurrency over time from a rocprofv3 rocpd database (--kernel-trace):
    python tools/rocpd_rate.py x_results.db [window_ms] [marker kernel substring]
Per window of window_ms: dispatches started, kernels per millisecond, the sum of kernel durations / window (= how many kernels ran
side by side on average), queues seen, and how many of the dispatches were the marker kernel (default: the few-row fused self-attention
kernel, i.e. the point steps of 8-image calls).  What bounds pipelined 8-image engine calls: the chip, or the rate at which kernels start?"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 20e6
    marker = sys.argv[3] if len(sys.argv) > 3 else 'dec_fused_self_attn_kernel'
    cols = [r[1] for r in db.execute('pragma table_info(rocpd_kernel_dispatch)').fetchall()]
    q = 'k.queue_id' if 'queue_id' in cols else '0'
    rows = db.execute('select k.start, k.end, s.display_name, %s from rocpd_kernel_dispatch k join rocpd_info_kernel_symbol s '
                      'on k.kernel_id = s.id order by k.start' % q).fetchall()
    if not rows:
        print('no dispatches')
        return
    t0 = rows[0][0]
    bins = {}
    for st, en, name, qu in rows:
        b = int((st - t0) // win)
        e = bins.setdefault(b, [0, 0.0, set(), 0])
        e[0] += 1
        e[1] += en - st
        e[2].add(qu)
        e[3] += 1 if marker in name else 0
    print('%9s %10s %12s %12s %7s %8s' % ('t_ms', 'dispatches', 'kernels/ms', 'concurrency', 'queues', 'marker'))
    for b in sorted(bins):
        n, busy, qs, mk = bins[b]
        print('%9.0f %10d %12.1f %12.2f %7d %8d' % (b * win / 1e6, n, n / (win / 1e6), busy / win, len(qs), mk))


if __name__ == '__main__':
    main()
