#!/usr/bin/env python
"""OmniParser text-spotting throughput on MI355X (BASELINE.json config 2).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU over RCCL)

Workload ("step" = one batch through the whole hot path): Swin-B -> FPN -> input_proj ->
memory K/V projection -> point decoder -> polygon decoder -> recognition decoder on a batch of
`--batch` synthetic 1024x1024 images per rank, bf16 engine, seeded procedural weights in the
reference's state-dict layout.  Random weights never emit a sensible EOS, so decoding is FORCED to
`--instances` text instances per image: 2*64+6 point steps, 32+2 polygon steps and
25+2 recognition steps, every step a full 4-layer decoder pass for every row.  Nothing is skipped
or cached across steps; images are resident in HBM before the timed region.

Engine scheduling (all inside the timed region): `--coalesce` consecutive steps are merged into one engine
call (dynamic batching: the latency-bound decoder steps then advance coalesce*batch images per launch)
and `--lanes` such groups are in flight at once on separate HIP streams (engine/pipeline.py).

Scaling is weak: every rank processes its own `--batch` images per step; ranks exchange one all-gather
of the decoded (padded) sequences per engine call, as a real image-sharded deployment would.

One JSON line on rank 0: images/s (whole job), chars/s, ms per step, the roofline record of the
dominant HBM-bound kernel (decoder cross-attention, timed with HIP events on its launch stream)
and the CPU baseline (the oracle = reference algorithm restated on CPU, timed on a bounded sample
of the same workload and scaled as described in its `sample` field).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s peak


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=64)
    p.add_argument('--warmup', type=int, default=16)
    p.add_argument('--batch', type=int, default=8, help='images per GPU per step')
    p.add_argument('--size', type=int, default=1024)
    p.add_argument('--instances', type=int, default=64, help='forced text instances per image')
    p.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32'])
    p.add_argument('--graph', type=int, default=int(os.environ.get('OMP355_GRAPH', '1')))
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--no-roofline', action='store_true')
    p.add_argument('--phase-times', action='store_true', help='also print a per-phase time breakdown (stderr)')
    p.add_argument('--overlap', type=int, default=1, help='polygon || recognition decoders on two streams')
    p.add_argument('--lanes', type=int, default=int(os.environ.get('OMP355_LANES', '2')),
                   help='step groups in flight per GPU (engine/pipeline.py): they overlap on separate HIP streams')
    p.add_argument('--coalesce', type=int, default=int(os.environ.get('OMP355_COALESCE', '16')),
                   help='consecutive steps (batches of --batch images) merged into one engine call: the decoders then '
                        'advance coalesce*batch images per launch (dynamic batching across steps); 1 = every step alone; '
                        'capped at ceil(steps / lanes) so that every lane gets work')
    return p.parse_args()


def build_model(dtype, graph, device):
    from advancedliteratemachinery_amd.model import OmniParser
    from advancedliteratemachinery_amd.utils.parser import make_args
    from advancedliteratemachinery_amd.utils import synthetic as weights   # seeded procedural checkpoint (data only)
    args = make_args(tfm_pre_norm=True, use_fpn=True, use_char_window_prompt=True)
    sd = weights.make_state_dict(args, seed=0)
    model = OmniParser(args, engine_dtype=dtype)
    model.load_state_dict(sd)
    model = model.to(device)
    model.use_graph = bool(graph)
    return model, args, sd


def phase_breakdown(model, one_step, stream):
    model.phase_events = []
    with torch.cuda.stream(stream):
        one_step()
        torch.cuda.synchronize()
    ev = model.phase_events
    model.phase_events = None
    return {ev[i][0]: ev[i - 1][1].elapsed_time(ev[i][1]) for i in range(1, len(ev))}


def prompts(args):
    nb = args.num_bins
    pt = torch.tensor([[0, 0, nb - 1, nb - 1, nb, nb + len(args.chars), args.pt_sos_index]], dtype=torch.long)
    return [pt, torch.full((1, 1), args.poly_sos_index, dtype=torch.long), torch.full((1, 1), args.rec_sos_index, dtype=torch.long)]


def gather_results(results, B, N, rec_len, world, device):
    """Image-sharded deployment: one all-gather of padded token tensors per batch."""
    ids = torch.zeros(B, N, 2 + 32 + rec_len, dtype=torch.int32, device=device)
    probs = torch.zeros(B, N, rec_len, dtype=torch.float32, device=device)
    for b, r in enumerate(results):
        if r is None:
            continue
        (pt, poly, rec), (pr,) = r
        n = min(N, pt.numel() // 2)
        ids[b, :n, 0:2] = pt.reshape(-1, 2)[:n].int()
        ids[b, :n, 2:34] = poly.reshape(-1, 32)[:n].int()
        ids[b, :n, 34:] = rec[0][:n].int()
        probs[b, :n] = pr[:n]
    if world > 1:
        all_ids = torch.empty(world * B, N, ids.shape[2], dtype=torch.int32, device=device)
        all_pr = torch.empty(world * B, N, rec_len, dtype=torch.float32, device=device)
        dist.all_gather_into_tensor(all_ids, ids)
        dist.all_gather_into_tensor(all_pr, probs)
        return all_ids, all_pr
    return ids, probs


def pmc_traffic(images_per_launch):
    """HBM bytes per cross-attention launch from the rocprofv3 PMC passes (separate runs; tools/pmc_cross_json.py
    turns their summaries into profiles/pmc_cross_attn.json).  FETCH_SIZE is in KiB and counts 16-byte/lane
    streaming reads at half their size on gfx950 (MI355X_MICROARCH.md, HBM) -> x2; WRITE_SIZE is in KiB."""
    path = os.environ.get('OMP355_PMC_JSON', os.path.join(ROOT, 'profiles', 'pmc_cross_attn.json'))
    try:
        with open(path) as f:
            rec = json.load(f)
        if int(rec['images_per_launch']) != int(images_per_launch):
            return None
        return float(rec['fetch_kib_mean']) * 1024.0 * 2.0 + float(rec['write_kib_mean']) * 1024.0
    except (OSError, ValueError, KeyError):
        return None


def host_cores():
    """Cores this process may really use: affinity mask, capped by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            txt = open(path).read().split()
            if path.endswith('cpu.max'):
                if txt[0] != 'max':
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
                    n = min(n, max(1, int(q / per + 0.5)))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, n)


def cpu_baseline(args, sd, size, instances, pt_steps, budget_s=40.0):
    """Reference algorithm (oracle restatement: no KV cache, full prefix re-decoded every step, memory
    broadcast per instance) on the host cores -- a BOUNDED sample scaled to the GPU workload:
    backbone+FPN+projection of one image at (size/2)^2 (x4: Swin cost is linear in pixels), then single
    decoder calls against a full-size (size/16)^2 memory: point decoder at a short and a mid prefix,
    polygon / recognition at N=2 instances (reference decode cost is linear in instances and ~affine in
    prefix length).  Every leg is skipped (and the number marked partial) once `budget_s` is spent."""
    from oracle import omniparser_ref as O
    cores = host_cores()
    torch.set_num_threads(min(cores, 64))
    t_start = time.time()
    left = lambda: budget_s - (time.time() - t_start)   # noqa: E731
    log = lambda m: print('[cpu_baseline] ' + m, file=sys.stderr, flush=True)  # noqa: E731
    g = torch.Generator().manual_seed(1234)
    sd = {k: (v.float() if v.is_floating_point() else v) for k, v in sd.items()}
    half = max(64, size // 2)
    d = args.tfm_hidden_dim
    M = (size // 16) ** 2
    notes = []
    with torch.no_grad():
        img = torch.randn(1, 3, half, half, generator=g)
        t0 = time.time()
        O.encode(sd, args, img, torch.zeros(1, half, half, dtype=torch.bool))
        t_enc = (time.time() - t0) * (size / float(half)) ** 2
        log('encode %dx%d: %.2fs -> %.2fs at %dx%d (cores=%d)' % (half, half, time.time() - t0, t_enc, size, size, cores))
        mem = torch.randn(M, 1, d, generator=g)
        pos = torch.randn(M, 1, d, generator=g)
        m = torch.zeros(1, M, dtype=torch.bool)

        def step_time(kind, n, L):
            if left() <= 0:
                return None
            seq = torch.randint(0, args.num_bins, (n, L), generator=g)
            ts = []
            for _ in range(3):   # median of 3 while the budget lasts
                t = time.time()
                O.decode(sd, args, seq, mem, m, pos, kind)
                ts.append(time.time() - t)
                if left() <= 0:
                    break
            dt_ = sorted(ts)[len(ts) // 2]
            log('%s decode call n=%d L=%d: %.3fs' % (kind, n, L, dt_))
            return dt_

        n_s = 2
        t_pt_a = step_time('pt', 1, 7)
        t_poly = step_time('poly', n_s, 3 + 16)
        t_rec = step_time('rec', n_s, 3 + 12)
        t_pt_b = step_time('pt', 1, 7 + pt_steps // 2)
    if t_pt_a is None:
        raise RuntimeError('cpu_baseline: budget too small for a single decoder call')
    if t_pt_b is None:
        t_pt_b = t_pt_a
        notes.append('mid-prefix point step not measured (budget)')
    if t_poly is None or t_rec is None:
        t_poly = t_rec = t_pt_a * n_s
        notes.append('polygon/recognition steps estimated from the point step (budget)')
    t_pt = pt_steps * 0.5 * (t_pt_a + t_pt_b)
    t_total = t_enc + t_pt + (32 * t_poly + args.rec_length * t_rec) * (instances / n_s)
    return dict(value=1.0 / t_total, unit='images/s', cores=cores, kind='port',
                sample=('oracle (CPU restatement of the reference path, fp32, %d threads): encode of one %dx%d image '
                        'measured and scaled x%.0f to %dx%d = %.2fs; point-decoder call %.3fs (L=7) / %.3fs (L=%d) '
                        'measured against a %d-token memory, x%d steps; polygon / recognition call %.3fs / %.3fs '
                        'measured at N=%d instances, scaled linearly to N=%d (x32 / x%d steps); estimated '
                        'full-workload time %.1fs per image%s'
                        % (min(cores, 64), half, half, (size / float(half)) ** 2, size, size, t_enc, t_pt_a, t_pt_b,
                           7 + pt_steps // 2, M, pt_steps, t_poly, t_rec, n_s, instances, args.rec_length, t_total,
                           ('; ' + '; '.join(notes)) if notes else '')),
                chars_per_sec=instances * args.rec_length / t_total)


def main():
    a = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if a.gpus != world and world > 1:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (a.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU fallback for the hot path)')
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', init_method='env://', device_id=device)

    from advancedliteratemachinery_amd import _lib
    model, args, sd = build_model(a.dtype, a.graph, device)
    model.overlap_decoders = bool(a.overlap)
    model.engine()   # pack the weights once, before any lane thread asks for them
    B, N = a.batch, a.instances
    g = torch.Generator(device='cpu').manual_seed(1234 + rank)
    img = torch.randn(B, 3, a.size, a.size, generator=g).to(device)   # resident in HBM before timing
    mask = torch.zeros(B, a.size, a.size, dtype=torch.bool, device=device)
    seqs = prompts(args)
    stream = torch.cuda.Stream(device=device)

    from advancedliteratemachinery_amd.engine.pipeline import LanePool
    lanes = max(1, a.lanes)
    pool = LanePool(device, lanes) if lanes > 1 else None

    # steps per engine call: `coalesce`, but never so many that a lane would stay idle in a short run
    G = max(1, min(a.coalesce, -(-a.steps // lanes)))

    def group_input(g):
        """g consecutive steps as one engine call: the g batches are concatenated inside the timed region (a serving
        engine receives them as separate tensors)."""
        if g == 1:
            return img, mask
        return torch.cat([img] * g, 0), torch.cat([mask] * g, 0)

    def one_step():
        """synchronous form (phase breakdown, roofline leg): ONE batch on the current stream"""
        res = model.infer(img, mask, seqs, forced_instances=N, has_padding=False)
        return gather_results(res, B, N, args.rec_length, world, device)

    def run_group(g, lane=None):
        gi, gm = group_input(g)
        res = model.infer(gi, gm, seqs, forced_instances=N, has_padding=False, lane=lane)
        return gather_results(res, B * g, N, args.rec_length, 1, device)

    def run_steps(k):
        """k steps = k batches through the whole hot path, in groups of `coalesce` consecutive steps per engine call.
        With lanes > 1 consecutive groups are in flight on different HIP streams (lane threads enqueue them); the
        all-gather of the decoded sequences (one per group) is issued from THIS thread in step order, after the
        lane's completion event, so every rank calls the collectives in the same order."""
        sizes = [G] * (k // G) + ([k % G] if k % G else [])
        out = None
        if pool is None:
            for g in sizes:
                ids, probs = run_group(g)
                out = exchange(ids, probs, g)
            return out
        futs = [pool.submit(lambda lane, g=g: run_group(g, lane)) for g in sizes]
        for f, g in zip(futs, sizes):
            (ids, probs), ev = f.result()
            torch.cuda.current_stream().wait_event(ev)
            out = exchange(ids, probs, g)
        return out

    def exchange(ids, probs, g):
        if world == 1:
            return ids, probs
        all_ids = torch.empty(world * B * g, N, ids.shape[2], dtype=torch.int32, device=device)
        all_pr = torch.empty(world * B * g, N, args.rec_length, dtype=torch.float32, device=device)
        dist.all_gather_into_tensor(all_ids, ids)
        dist.all_gather_into_tensor(all_pr, probs)
        return all_ids, all_pr

    def barrier():
        if world > 1:
            dist.barrier()

    with torch.cuda.stream(stream):
        # untimed set-up: every lane (or the model itself) allocates its buffers and captures its graphs for
        # both group sizes the timed region will use (full groups and the remainder group)
        for g in sorted({G, a.steps % G, a.warmup % G} - {0}):
            for _ in range(lanes):
                run_steps(g)
        run_steps(a.warmup)
        torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
        out = run_steps(a.steps)
        torch.cuda.synchronize()
        barrier()
        elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    total_images = world * B * a.steps
    ips = total_images / elapsed
    # sanity: the forced workload really produced N instances x rec_length chars per image
    ids, _ = out
    last_g = a.steps % G or G
    assert ids.shape[0] == world * B * last_g and int((ids[:, :, 34:] >= args.num_bins).all()), 'decode output malformed'

    if a.phase_times and rank == 0:
        print('phase ms: %s' % json.dumps(phase_breakdown(model, one_step, stream)), file=sys.stderr, flush=True)

    roof = None
    if rank == 0 and not a.no_roofline:
        # Dominant HBM-bound kernel: decoder cross-attention (streams K and V^T of every image once per
        # launch, shared by all query rows).  Timed with HIP events on the launch stream over the same
        # K steps, eager launches (events cannot bracket kernels inside a graph replay).
        enc, dec = model.engine()
        was = model.use_graph
        model.use_graph = False
        h = _lib.lib()
        n_groups = (a.steps + G - 1) // G
        with torch.cuda.stream(stream):
            run_group(G)
            torch.cuda.synchronize()
            h.omp_prof_enable(1)
            for _ in range(n_groups):   # the same engine calls as the timed region (groups of G steps), eagerly
                run_group(G)
            torch.cuda.synchronize()
        tot, cnt = ctypes.c_double(0), ctypes.c_int64(0)
        h.omp_prof_read(ctypes.byref(tot), ctypes.byref(cnt))
        h.omp_prof_enable(0)
        model.use_graph = was
        M = (a.size // 16) ** 2
        esz = 2 if a.dtype == 'bf16' else 4
        BI = B * G   # images per launch
        # algorithmic bytes per launch: K + V^T of the images in the call (d = 512) + q in / o out of
        # the rows (launch-weighted: 1 row/image in the point phase, N rows/image in polygon / recognition)
        rows_avg = BI * (1 * (2 * N + 6) + N * (34 + 27)) / float((2 * N + 6) + 34 + 27)
        alg = BI * 2 * M * 512 * esz + rows_avg * 2 * 512 * esz
        avg_s = (tot.value / 1e3) / max(1, cnt.value)
        ach = alg / avg_s / 1e9
        roof = dict(bound='hbm', kernel='dec_cross_attn_kernel / dec_cross_attn_q4_kernel', achieved=ach, peak=HBM_PEAK_GBS,
                    unit='GB/s', frac=ach / HBM_PEAK_GBS, traffic=pmc_traffic(BI), launches=int(cnt.value),
                    avg_us=avg_s * 1e6, alg_bytes_per_launch=alg, images_per_launch=BI,
                    note='hipEvent-bracketed eager launches of the same %d engine calls (graph replay cannot be bracketed); '
                         'traffic = rocprofv3 FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE per launch from the committed PMC passes '
                         '(profiles/), null when they were taken at another images-per-launch' % n_groups)

    if rank == 0:
        rec = dict(metric='images/sec (1024x1024) + chars/sec decoded, OmniParser text-spotting', value=ips, unit='images/s',
                   n_gpus=world, steps=a.steps, warmup=a.warmup, ms_per_step=elapsed / a.steps * 1e3,
                   higher_is_better=True, scaling='weak', vs_baseline=None, dtype=a.dtype, data='synthetic',
                   chars_per_sec=ips * N * args.rec_length,
                   config=dict(workload='OmniParser text-spotting, Swin-B, batch %d/GPU @ %dx%d, forced %d instances/image '
                                        '(%d pt + 34 poly + 27 rec decoder steps), %s' % (B, a.size, a.size, N, 2 * N + 6, a.dtype),
                               global_batch=world * B, image_size=a.size, instances_per_image=N, parallelism='image-sharded dp%d' % world,
                               hip_graph=bool(a.graph), lanes=lanes, coalesce=G,
                               images_per_engine_call=B * G))
        if roof is not None:
            rec['roofline'] = roof
        if not a.no_cpu_baseline and world == 1:
            rec['cpu_baseline'] = cpu_baseline(args, sd, a.size, N, 2 * N + 1)
        print(json.dumps(rec), flush=True)
    if pool is not None:
        pool.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
