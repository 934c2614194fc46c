"""Turn the FETCH_SIZE / WRITE_SIZE summaries of tools/pmc_summary.py into the small record bench.py reads for
roofline.traffic:   python tools/pmc_cross_json.py <fetch.txt> <write.txt> <images_per_launch> "<command>" > profiles/pmc_cross_attn.json
Launch-weighted mean over every kernel whose name contains dec_cross_attn (both query-tile variants)."""
import json
import sys


def mean_of(path):
    n, tot = 0, 0.0
    for line in open(path):
        if 'dec_cross_attn' not in line:
            continue
        f = line.split()
        calls, total = int(f[-4]), float(f[-1])
        n += calls
        tot += total
    return (tot / n if n else 0.0), n


def main():
    fm, fn = mean_of(sys.argv[1])
    wm, wn = mean_of(sys.argv[2])
    print(json.dumps(dict(kernel='dec_cross_attn*', fetch_kib_mean=fm, write_kib_mean=wm, launches_fetch=fn, launches_write=wn,
                          images_per_launch=int(sys.argv[3]), command=sys.argv[4] if len(sys.argv) > 4 else '',
                          units='KiB as reported by rocprofv3; bench.py applies the gfx950 x2 correction to FETCH_SIZE'), indent=1))


if __name__ == '__main__':
    main()
