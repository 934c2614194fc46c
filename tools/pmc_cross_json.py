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
    """python tools/pmc_cross_json.py <fetch.txt> <write.txt> <images_per_launch> "<command>" [existing.json]
    prints the JSON file bench.py reads: one record per images-per-launch (an existing file is merged into)."""
    fm, fn = mean_of(sys.argv[1])
    wm, wn = mean_of(sys.argv[2])
    out = {}
    if len(sys.argv) > 5:
        try:
            old = json.load(open(sys.argv[5]))
            out = {str(old['images_per_launch']): old} if 'images_per_launch' in old else dict(old)
        except (OSError, ValueError):
            out = {}
    key = sys.argv[3]   # images per launch; 'x3_<images>' = the split-plane kernels of the parity engine
    out[key] = dict(kernel='dec_cross_attn*', fetch_kib_mean=fm, write_kib_mean=wm, launches_fetch=fn, launches_write=wn,
                                      images_per_launch=int(key.split('_')[-1]), command=sys.argv[4] if len(sys.argv) > 4 else '',
                                      units='KiB as reported by rocprofv3; bench.py applies the gfx950 x2 correction to FETCH_SIZE')
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
