"""HBM traffic of the row-owner chain kernels (csrc/dec_rows.hip) for bench.py's roofline class, from rocprofv3 --pmc passes:
    (cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -o pmc -- python tools/dec_rows_pmc.py run)   (and WRITE_SIZE)
    python tools/dec_rows_pmc.py summarise <FETCH csv> <WRITE csv>  > profiles/pmc_dec_rows.json
`run`: a few eager steps of the polygon and recognition decoders over 160 images x 64 instances (R = 10 240 rows, M = 4096 memory tokens), the
shape of the bench's engine call; every kernel of the step runs, the summary keeps the chain kernels.  FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE
counts a wide streaming read at half its size on gfx950 (MI355X_MICROARCH.md) -> x2, as for the other classes.  Algorithmic bytes per launch = rows
in and out as laid out (att bf16, x fp32 read + written, q / q k v bf16 or logits fp32 written) + the launch's weight stream ONCE (every
workgroup re-reads it from L2: those re-reads are not HBM traffic and not algorithmic)."""
import collections
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
I, N, M, D, V = 160, 64, 4096, 512, 1104


def run():
    import torch
    import bench as B_
    model, args, _ = B_.build_model('bf16', 0, torch.device('cuda'))
    _, dec = model.engine()
    gen = torch.Generator(device='cpu').manual_seed(0)
    mem = torch.randn(I * M, D, generator=gen).to('cuda', torch.bfloat16)
    kv = dec.project_memory(mem, mem, I, M, None)
    counts = [N] * I
    pts = torch.randint(0, args.num_bins, (N * I, 2), generator=gen, dtype=torch.int32).to('cuda')
    dec.use_graph = False
    dec.rows_min = 1
    for kind, sos in (('poly', args.poly_sos_index), ('rec', args.rec_sos_index)):
        ph = dec.begin_instances(kind, kv, pts, counts, sos, 4)
        dec._run(ph, 0, 6)
    torch.cuda.synchronize()
    print('ran 2 x 6 steps of %d rows' % (N * I))


def alg_bytes(name, R):
    stream = {'mid': 128, 'qkv': 64 + 512 + 192, 'head': 64 + 512 + 128 + 2 * 64 + 16, 'embed': 192}
    rows = {'mid': R * D * (2 + 4 + 4 + 2), 'qkv': R * D * (2 + 4 + 4) + R * 3 * D * 2, 'head': R * D * (2 + 4 + 4) + R * V * 4, 'embed': R * D * 4 + R * 3 * D * 2 + R * D * 4}
    return rows[name] + stream[name] * 8 * 1024


def classify(kname):
    if 'dec_rows_mid_kernel' in kname:
        return 'mid'
    m = re.search(r'dec_rows_ffn_kernel<\d+, *(\d), *(\d)(?:, *\d)?>|dec_rows_ffn_kernelILi\d+ELi(\d)ELi(\d)E', kname)
    if m:
        pro, tail = (m.group(1), m.group(2)) if m.group(1) is not None else (m.group(3), m.group(4))
        return 'embed' if pro == '1' else ('qkv' if tail == '0' else 'head')
    return None


def read(path, ctr):
    agg = collections.defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r.get('Counter_Name') != ctr:
                continue
            c = classify(r['Kernel_Name'])
            if c:
                agg[c][0] += 1
                agg[c][1] += float(r['Counter_Value'])
    return agg


def summarise(fetch_csv, write_csv):
    R = I * N
    f, w = read(fetch_csv, 'FETCH_SIZE'), read(write_csv, 'WRITE_SIZE')
    per = {}
    tot_meas = tot_alg = launches = 0
    for c in ('embed', 'mid', 'qkv', 'head'):
        if f[c][0] == 0 or w[c][0] == 0:
            continue
        meas = f[c][1] / f[c][0] * 1024.0 * 2.0 + w[c][1] / w[c][0] * 1024.0
        alg = alg_bytes(c, R)
        per[c] = dict(launches=f[c][0], fetch_kib_mean=f[c][1] / f[c][0], write_kib_mean=w[c][1] / w[c][0], measured_bytes_per_launch=meas,
                      algorithmic_bytes_per_launch=alg, measured_over_alg=meas / alg)
        tot_meas += meas * f[c][0]
        tot_alg += alg * f[c][0]
        launches += f[c][0]
    out = dict(command='rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> --kernel-trace --output-format csv -- python tools/dec_rows_pmc.py run',
               shape=dict(images=I, rows=R, memory_tokens=M), kernels=per,
               summary=dict(measured_bytes_per_launch=tot_meas / max(1, launches), algorithmic_bytes_per_launch=tot_alg / max(1, launches),
                            measured_over_alg=tot_meas / max(1.0, tot_alg), launches=launches,
                            scope='launch-weighted over the chain kernels of 2 x 6 eager decoder steps at 10 240 rows (1 embedding, 4 mid, 3 q-k-v and 1 head launch per step); '
                                  'FETCH_SIZE x2 (gfx950) + WRITE_SIZE'))
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    if sys.argv[1] == 'run':
        run()
    else:
        summarise(sys.argv[2], sys.argv[3])
