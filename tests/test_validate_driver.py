"""engine/inference.py::validate / predict end to end on the CPU with a STUB model (deterministic fake decodes keyed by
image content): the drop-in for the reference's engine.validate (OCR/OmniParser/engine/val.py:11-100) -- batching of
dataloader items, padding of mixed sizes, record formatting, the JSON files, and the image-sharded path: world-size-2
gloo, shard_range split, one all-gather, rank 0 writes.  The model arithmetic is out of scope here (tests/test_gpu_*)."""
import json
import os

import torch
import torch.multiprocessing as mp

from advancedliteratemachinery_amd.engine import inference as inf
from advancedliteratemachinery_amd.utils import dist as udist
from advancedliteratemachinery_amd.utils.parser import make_args

REC = 25


class RefStyleNested(object):
    """what the REFERENCE's collate hands over: its own NestedTensor class (utils/nested_tensor.py:7-35), not ours"""

    def __init__(self, tensors, mask):
        self.tensors, self.mask = tensors, mask

    def to(self, device):
        return RefStyleNested(self.tensors.to(device), self.mask.to(device))


def _key(img, mask):
    """content key of the UNPADDED image (so batching / padding cannot change it)"""
    h = int((~mask[:, 0]).sum())
    w = int((~mask[0, :]).sum())
    return int(img[:, :h, :w].abs().sum().item() * 10) % 997


def _fake(key, kie):
    n = key % 4     # 0 instances -> None, exercised on purpose
    if n == 0:
        return None
    if kie:
        return [('w%d' % key, 'total', 0.5, [[1.0, 2.0, 3.0, 4.0]])] * n
    g = torch.Generator().manual_seed(key)
    return ([torch.randint(0, 1000, (1, 2 * n), generator=g), torch.randint(0, 1000, (1, 32 * n), generator=g),
             torch.randint(1000, 1096, (1, n, REC), generator=g)], [torch.rand(n, REC, generator=g)])


class StubModel(torch.nn.Module):
    def __init__(self, kie=False):
        super().__init__()
        self.p = torch.nn.Parameter(torch.zeros(1))
        self.kie = kie
        self.calls = []

    def infer(self, img, mask, seqs, has_padding=None, **kw):
        self.calls.append((tuple(img.shape), bool(has_padding)))
        assert seqs[0].tolist() == [[0, 0, 999, 999, 1000, 1095, 1100]]
        return [_fake(_key(img[b], mask[b]), self.kie) for b in range(img.shape[0])]


def _loader(n, ref_style=True):
    """batch-1 items of different sizes, like the reference's val dataloader"""
    items = []
    for i in range(n):
        g = torch.Generator().manual_seed(50 + i)
        h, w = 32 + 8 * (i % 3), 40 + 8 * (i % 2)
        img = torch.randn(1, 3, h, w, generator=g)
        nt = RefStyleNested(img, torch.zeros(1, h, w, dtype=torch.bool))
        if not ref_style:
            from advancedliteratemachinery_amd.utils.nested_tensor import NestedTensor
            nt = NestedTensor(nt.tensors, nt.mask)
        items.append((nt, [{'file_name': 'img_%02d.jpg' % i, 'orig_size': torch.tensor([h * 2, w * 2]), 'dataset_name': 'unit_val'}]))
    return items


def _expected(n, args):
    out = []
    for samples, targets in _loader(n):
        r = _fake(_key(samples.tensors[0], samples.mask[0]), False)
        if r is None:
            continue
        out.extend(inf.decode_pred_seq([t[0] for t in r[0]], r[1][0], targets[0], args))
    return out


def _same(a, b):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert x['image_id'] == y['image_id'] and x['rec'] == y['rec'] and abs(x['score'] - y['score']) < 1e-6
        assert torch.allclose(torch.tensor(x['polys']), torch.tensor(y['polys'])) and torch.allclose(torch.tensor(x['pts']), torch.tensor(y['pts']))


def test_validate_single_process_batches_and_writes_json(tmp_path):
    args = make_args(use_char_window_prompt=True, output_folder=str(tmp_path))
    for bs, ref_style in ((1, True), (3, True), (4, False)):
        model = StubModel()
        got = inf.validate(model, _loader(7, ref_style), 3, args, batch_size=bs)
        _same(got, _expected(7, args))
        assert len(model.calls) == -(-7 // bs)
        if bs > 1:   # mixed sizes were padded and flagged
            assert any(pad for _, pad in model.calls)
        path = os.path.join(str(tmp_path), 'results', 'ep003', 'unit_val.json')
        _same(json.load(open(path)), _expected(7, args))


def test_predict_accepts_reference_nested_tensor():
    args = make_args(use_char_window_prompt=True)
    samples, targets = _loader(3)[2]
    recs = inf.predict(StubModel(), samples, args, targets=targets)
    assert len(recs) == 1
    r = _fake(_key(samples.tensors[0], samples.mask[0]), False)
    if r is not None:
        _same(recs[0], inf.decode_pred_seq([t[0] for t in r[0]], r[1][0], targets[0], args))


def test_validate_kie_writes_one_json_per_image(tmp_path):
    args = make_args(use_char_window_prompt=True, output_folder=str(tmp_path), vie_categories=4, infer_vie=True, val_dataset=['sroie_val'])
    got = inf.validate(StubModel(kie=True), _loader(6), 0, args, batch_size=2)
    folder = os.path.join(str(tmp_path), 'results', 'ep000')
    n_files = 0
    for samples, targets in _loader(6):
        r = _fake(_key(samples.tensors[0], samples.mask[0]), True)
        path = os.path.join(folder, targets[0]['file_name'] + '.json')
        if r is None:
            assert not os.path.exists(path)      # the reference skips empty outputs (val.py:36-37)
            continue
        n_files += 1
        assert json.load(open(path)) == json.loads(json.dumps(r))
    assert len(got) == n_files


def _worker(rank, world, port, folder, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    assert udist.init_distributed_mode(backend='gloo')
    args = make_args(use_char_window_prompt=True, output_folder=folder)
    model = StubModel()
    got = inf.validate(model, _loader(7), 1, args, batch_size=2)
    lo, hi = udist.shard_range(7, rank, world)
    q.put((rank, len(got), sum(b for (b, _, _, _), _ in model.calls), hi - lo))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_validate_world2_gloo_shards_gathers_and_rank0_writes(tmp_path):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    args = make_args(use_char_window_prompt=True)
    exp = _expected(7, args)
    (r0, n0, imgs0, shard0), (r1, n1, imgs1, shard1) = got
    assert (n0, n1) == (len(exp), 0)                       # only rank 0 returns / writes
    assert (imgs0, imgs1) == (shard0, shard1) and shard0 + shard1 == 7   # every rank decoded exactly its shard
    _same(json.load(open(os.path.join(str(tmp_path), 'results', 'ep001', 'unit_val.json'))), exp)


def test_rank_items_reads_only_the_ranks_own_shard():
    """a sharded validate must not decode the other ranks' images.  A torch DataLoader over a map-style dataset is
    re-pointed at Subset(dataset, [lo, hi)): __getitem__ runs for this rank's indices only; sequences are sliced; unsized
    iterables are refused for world > 1 instead of being materialised with list()."""
    import pytest

    class Counting(torch.utils.data.Dataset):
        def __init__(self, n):
            self.n, self.touched = n, []

        def __len__(self):
            return self.n

        def __getitem__(self, i):
            self.touched.append(i)
            return torch.full((3, 8, 8), float(i)), {'file_name': 'f%d' % i, 'orig_size': (8, 8)}

    ds = Counting(10)
    dl = torch.utils.data.DataLoader(ds, batch_size=1, shuffle=False, num_workers=0, collate_fn=lambda b: b[0])
    items, lo, hi = inf._rank_items(dl, 1, 3, False)
    got = [int(img[0, 0, 0]) for img, _ in items]
    assert (lo, hi) == udist.shard_range(10, 1, 3) and got == list(range(lo, hi)) and sorted(ds.touched) == list(range(lo, hi))
    seq = list(range(7))
    assert list(inf._rank_items(seq, 0, 2, False)[0]) == [0, 1, 2, 3] and list(inf._rank_items(seq, 1, 2, False)[0]) == [4, 5, 6]
    assert inf._rank_items(seq, 1, 2, True)[0] is seq and inf._rank_items(seq, 0, 1, False)[0] is seq   # already sharded / single rank
    with pytest.raises(TypeError):
        inf._rank_items(iter(seq), 0, 2, False)
    # a full-length NON-sequential sampler defines the item order itself -- the shard is cut from THAT order (islice),
    # the loader is not silently re-pointed at Subset(dataset, range(lo, hi))
    order = [9, 0, 8, 1, 7, 2, 6, 3, 5, 4]
    dl2 = torch.utils.data.DataLoader(Counting(10), batch_size=1, sampler=order, num_workers=0, collate_fn=lambda b: b[0])
    items, lo, hi = inf._rank_items(dl2, 1, 3, False)
    assert [int(img[0, 0, 0]) for img, _ in items] == order[lo:hi]
    # the re-pointed loader keeps the original's worker_init_fn
    mark = lambda _: None   # noqa: E731
    dl3 = torch.utils.data.DataLoader(Counting(10), batch_size=1, shuffle=False, num_workers=0, collate_fn=lambda b: b[0], worker_init_fn=mark)
    assert inf._rank_items(dl3, 0, 2, False)[0].worker_init_fn is mark


def test_plan_eos_balance_is_a_balanced_partition_with_homogeneous_batches():
    """dense pages sit together in a dataset -> contiguous shards make one rank the straggler; the plan spreads predicted cost over
    the ranks (LPT, shard sizes as shard_range) and batches images of similar predicted count."""
    g = torch.Generator().manual_seed(7)
    n, world, bs = 203, 8, 4
    pred = [int(v) for v in torch.randint(1, 8, (n,), generator=g)]
    for i in range(40):            # a run of dense images at the front of the dataset
        pred[i] = 40 + (i % 25)
    plan = inf.plan_eos_balance(pred, world, bs)
    flat = [i for batches in plan for b in batches for i in b]
    assert sorted(flat) == list(range(n))                                                      # a partition
    sizes = [sum(len(b) for b in batches) for batches in plan]
    assert sizes == [udist.shard_range(n, r, world)[1] - udist.shard_range(n, r, world)[0] for r in range(world)]
    assert all(len(b) <= bs for batches in plan for b in batches)
    contiguous = [[list(range(lo, hi))[o:o + bs] for o in range(0, hi - lo, bs)] for lo, hi in (udist.shard_range(n, r, world) for r in range(world))]
    c_new, c_old = inf.plan_cost(plan, pred), inf.plan_cost(contiguous, pred)
    assert max(c_new) < 0.5 * max(c_old)                                                       # the straggler is gone
    assert max(c_new) <= 1.25 * (sum(c_new) / world)                                           # ranks within 25 % of the mean
    for batches in plan:                                                                       # descending, homogeneous batches
        tops = [max(pred[i] for i in b) for b in batches]
        assert tops == sorted(tops, reverse=True)
    assert inf.plan_eos_balance(pred, world, bs) == plan                                       # deterministic: no communication needed


def _worker_balanced(rank, world, port, folder, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    assert udist.init_distributed_mode(backend='gloo')
    args = make_args(use_char_window_prompt=True, output_folder=folder)
    loader = _loader(9)
    args.eos_pred_counts = [_key(s.tensors[0], s.mask[0]) % 4 for s, _ in loader]   # the stub's true instance counts as the prediction
    model = StubModel()
    got = inf.validate(model, loader, 2, args, batch_size=2)
    plan = inf.plan_eos_balance(args.eos_pred_counts, world, 2)
    q.put((rank, len(got), [b for (b, _, _, _), _ in model.calls], [len(b) for b in plan[rank]]))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_validate_world2_eos_balanced_shards_give_the_same_json(tmp_path):
    """args.eos_pred_counts: balanced, non-contiguous shards and plan-cut engine calls -- the JSON is the contiguous run's, in dataset order"""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker_balanced, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    args = make_args(use_char_window_prompt=True)
    exp = _expected(9, args)
    assert got[0][1] == len(exp) and got[1][1] == 0
    for _, _, calls, planned in got:
        assert calls == planned                                  # every rank made exactly the engine calls of its plan
    _same(json.load(open(os.path.join(str(tmp_path), 'results', 'ep002', 'unit_val.json'))), exp)


def _loader_pairs(n_items):
    """dataloader items that hold TWO images each (validate's docstring allows it): same sizes inside an item, as a collate would give"""
    singles = _loader(2 * n_items)
    items = []
    for i in range(n_items):
        (a, ta), (b, tb) = singles[2 * i], singles[2 * i + 1]
        h = max(a.tensors.shape[2], b.tensors.shape[2])
        w = max(a.tensors.shape[3], b.tensors.shape[3])
        img = torch.zeros(2, 3, h, w)
        mask = torch.ones(2, h, w, dtype=torch.bool)
        for k, s in enumerate((a, b)):
            hh, ww = s.tensors.shape[2:]
            img[k, :, :hh, :ww] = s.tensors[0]
            mask[k, :hh, :ww] = False
        items.append((RefStyleNested(img, mask), ta + tb))
    return items


def _worker_pairs(rank, world, port, folder, q, balanced):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    assert udist.init_distributed_mode(backend='gloo')
    args = make_args(use_char_window_prompt=True, output_folder=folder)
    loader = _loader_pairs(5)
    if balanced:   # one predicted count per ITEM
        args.eos_pred_counts = [sum(_key(s.tensors[k], s.mask[k]) % 4 for k in range(2)) for s, _ in loader]
    model = StubModel()
    got = inf.validate(model, loader, 4 + int(balanced), args, batch_size=2)
    q.put((rank, len(got), [b for (b, _, _, _), _ in model.calls]))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_validate_world2_items_with_two_images_keep_dataset_order(tmp_path):
    """`lo`, `order` and the plan's cuts count dataloader ITEMS; with two images per item the per-image sort keys of the ranks used to
    overlap (contiguous shards) and the balanced plan was cut after the wrong image.  Both paths must write the single-image loader's JSON."""
    args = make_args(use_char_window_prompt=True)
    exp = _expected(10, args)
    for balanced in (False, True):
        ctx = mp.get_context('spawn')
        q = ctx.Queue()
        port = 35500 + (os.getpid() % 2000) + int(balanced)
        procs = [ctx.Process(target=_worker_pairs, args=(r, 2, port, str(tmp_path), q, balanced)) for r in range(2)]
        for p in procs:
            p.start()
        got = sorted(q.get(timeout=180) for _ in procs)
        for p in procs:
            p.join(timeout=60)
        assert got[0][1] == len(exp) and got[1][1] == 0
        assert sum(sum(c) for _, _, c in got) == 10                                  # every image decoded exactly once
        if balanced:
            assert all(b % 2 == 0 for _, _, c in got for b in c)                         # engine calls are cut between items
        _same(json.load(open(os.path.join(str(tmp_path), 'results', 'ep%03d' % (4 + int(balanced)), 'unit_val.json'))), exp)
