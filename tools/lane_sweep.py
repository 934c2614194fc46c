"""Throughput vs lanes / batch on one GPU, one process (development aid; bench.py is the contract).

  python tools/lane_sweep.py [--lanes 1,2,3,4,6,8] [--batches 8] [--steps 8]
Also prints, for one synchronous batch, the host time spent inside each phase's enqueue calls next to the
GPU time of the phase (is the lane thread or the GPU the limiter?)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--lanes', default='1,2,3,4,6,8')
    p.add_argument('--batches', default='8')
    p.add_argument('--steps', type=int, default=8)
    p.add_argument('--prio', default='0', help='comma list of 0/1: decoder phases on high-priority streams')
    p.add_argument('--side', type=int, default=1, help='0 = one HIP stream per lane (polygon then recognition on the lane stream)')
    a = p.parse_args()
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    model, args, sd = bench.build_model('bf16', 1, dev)
    model.overlap_decoders = True
    model.engine()
    seqs = bench.prompts(args)
    from advancedliteratemachinery_amd.engine.pipeline import LanePool
    print('GPU_MAX_HW_QUEUES=%s' % os.environ.get('GPU_MAX_HW_QUEUES'), flush=True)
    for B in [int(x) for x in a.batches.split(',')]:
        g = torch.Generator(device='cpu').manual_seed(1234)
        img = torch.randn(B, 3, 1024, 1024, generator=g).to(dev)
        mask = torch.zeros(B, 1024, 1024, dtype=torch.bool, device=dev)
        st = torch.cuda.Stream(device=dev)
        # synchronous batch: host time per phase (enqueue incl. internal syncs) -- from model.phase_events + wall
        with torch.cuda.stream(st):
            model.infer(img, mask, seqs, forced_instances=64, has_padding=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            model.phase_events = []
            model.infer(img, mask, seqs, forced_instances=64, has_padding=False)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            ev = model.phase_events
            model.phase_events = None
        ph = {ev[i][0]: ev[i - 1][1].elapsed_time(ev[i][1]) for i in range(1, len(ev))}
        print('B=%d sync batch: host return %.1f ms, gpu done %.1f ms, phases %s' % (B, (t1 - t0) * 1e3, (t2 - t0) * 1e3, {k: round(v, 1) for k, v in ph.items()}), flush=True)
        for L, prio in [(int(x), int(y)) for y in a.prio.split(',') for x in a.lanes.split(',')]:
            pool = LanePool(dev, L, dec_priority=bool(prio), side_streams=bool(a.side))
            def run(k):
                futs = [pool.submit(lambda lane: model.infer(img, mask, seqs, forced_instances=64, has_padding=False, lane=lane)) for _ in range(k)]
                for f in futs:
                    _, e = f.result()
                    e.wait()
            with torch.cuda.stream(st):
                run(L)
                run(2)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                nst = max(a.steps // max(1, B // 8), 2 * L)
                run(nst)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            print('B=%d lanes=%d prio=%d side=%d : %.1f ms/step  %.1f img/s  (%d steps)' % (B, L, prio, a.side, dt / nst * 1e3, B * nst / dt, nst), flush=True)
            pool.close()
            del pool
            torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
