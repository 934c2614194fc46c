"""bench.py's StepLoop over world-size-2 gloo with a STUB engine: the part of the bench line that must be right on N GPUs
and cannot be run here on hardware -- one packed all-gather per engine call in step order, EXACTLY K steps between barrier pairs, the MAX over
ranks, the all-reduced stop decision of the repetition loop, every rank's own ms per step gathered.  No GPU, no model: the engine call is a
function that returns deterministic token tensors keyed by (rank, step)."""
import os
import sys
import time

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

B, N, T, L = 2, 3, 7, 4    # images per step, instances, token columns, probability columns


def _tokens(rank, first, g):
    ids = torch.zeros(B * g, N, T, dtype=torch.int32)
    probs = torch.zeros(B * g, N, L)
    for s in range(g):
        for b in range(B):
            ids[s * B + b] = 1000 * rank + 10 * (first + s) + b
            probs[s * B + b] = rank + (first + s) / 100.0 + b / 1000.0
    return ids, probs


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import torch.distributed as dist
    import bench
    dist.init_process_group('gloo', init_method='env://')
    calls = []

    def run_group(first, g, lane, forced):
        calls.append((first, g, forced))
        if rank == 1:
            time.sleep(0.01 * g)          # the slow rank: the max over ranks is ITS time
        return _tokens(rank, first, g)
    loop = bench.StepLoop(run_group, world, rank, torch.device('cpu'), group=2)
    K = 5
    loop.timed(K, forced='N')      # untimed warm-up: gloo opens its pair connections inside the first collective (0.1-0.3 s on a loaded host, which
    del loop.local_s[:]            # made ONE repetition satisfy min_seconds and the >= 2 repetitions below fail once in ~20 CPU runs)
    n_warm, n_warm_calls = loop.n_gathers, len(calls)
    reps, out = loop.repeat(K, min_seconds=0.12, max_reps=6, forced='N')
    ids, probs = out
    # the last engine call of a 5-step region in groups of 2 is the remainder group (1 step): every rank holds every rank's rows, rank order
    ok = ids.shape == (world * B, N, T) and probs.shape == (world * B, N, L)
    for r in range(world):
        ei, ep = _tokens(r, 4, 1)
        ok &= bool(torch.equal(ids[r * B:(r + 1) * B], ei)) and bool(torch.equal(probs[r * B:(r + 1) * B], ep))   # bit patterns survive the int32 payload
    per_rank = loop.per_rank_ms(K)
    q.put(dict(rank=rank, ok=bool(ok), reps=[round(x, 6) for x in reps], n_reps=len(reps), gathers=loop.n_gathers - n_warm,
               calls=calls[n_warm_calls:n_warm_calls + 3], n_calls=len(calls) - n_warm_calls, per_rank=per_rank, local=[round(x, 6) for x in loop.local_s]))
    dist.barrier()
    dist.destroy_process_group()


def test_step_loop_world2_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 35500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=240) for _ in procs), key=lambda d: d['rank'])
    for p in procs:
        p.join(timeout=60)
    r0, r1 = got
    assert r0['ok'] and r1['ok']
    # every rank took the same stop decision and reports the same (max-over-ranks) time of every repetition
    assert r0['n_reps'] == r1['n_reps'] >= 2 and r0['reps'] == r1['reps']
    # ... which is the slow rank's: rank 1 sleeps 10 ms per step, rank 0 does not
    assert all(t >= 0.045 for t in r0['reps'])          # (the all-gather of every engine call makes the fast rank wait: its own time is the same)
    # exactly K steps per repetition, in groups of 2 + the remainder, one all-gather per engine call
    assert r0['calls'] == [(0, 2, 'N'), (2, 2, 'N'), (4, 1, 'N')] and r0['n_calls'] == 3 * r0['n_reps'] == r0['gathers']
    # every rank's own ms per step, gathered on every rank (>= the slow rank's 10 ms per step: the collectives hold the fast rank back)
    assert r0['per_rank'] == r1['per_rank'] and len(r0['per_rank']) == 2 and all(v >= 9.0 for v in r0['per_rank'])


def test_step_loop_single_rank_needs_no_process_group():
    import bench
    loop = bench.StepLoop(lambda f, g, lane, forced: _tokens(0, f, g), 1, 0, torch.device('cpu'), group=3)
    el, (ids, probs) = loop.timed(7)
    assert ids.shape[0] == B * 1 and loop.n_gathers == 0 and loop.per_rank_ms(7) is None and el > 0


def test_step_loop_one_rank_with_collectives_on():
    """bench.py --rccl-selftest: the N > 1 protocol with ONE rank (the only RCCL run a 1-GPU box allows; here over gloo) -- every engine call
    still goes through the packed all-gather, the barriers and the all-reduced stop decision, and the payload comes back bit for bit."""
    import torch.distributed as dist
    import bench
    port = 37600 + (os.getpid() % 2000)
    os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', init_method='env://')
    try:
        loop = bench.StepLoop(lambda f, g, lane, forced: _tokens(0, f, g), 1, 0, torch.device('cpu'), group=3, collectives=True)
        reps, (ids, probs) = loop.repeat(7, min_seconds=0.0, max_reps=2)
        ei, ep = _tokens(0, 6, 1)
        assert len(reps) == 1 and loop.n_gathers == 3                  # groups of 3, 3, 1
        assert torch.equal(ids, ei) and torch.equal(probs, ep)
        assert len(loop.per_rank_ms(7)) == 1
    finally:
        dist.destroy_process_group()
        for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
            os.environ.pop(k, None)
