"""Batched inference driver: the MI355X replacement for the reference's batch-1 `validate` loop
(OCR/OmniParser/engine/val.py:11-100), plus a `predict()` convenience API.

  * prompts are built exactly like val.py:25-33;
  * B images per call (the reference asserts 1);
  * `validate` is image-sharded: under torch.distributed every rank decodes a contiguous shard of the dataloader's
    items (utils/dist.py::shard_range), the decoded token tensors meet in ONE all-gather per tensor at the end
    (pack_results / all_gather_results: RCCL over xGMI on MI355X, gloo in the CPU tests) and rank 0 alone writes the
    JSON -- the reference writes the same file from every rank (val.py:64-68);
  * results are formatted like decode_pred_seq (val.py:70-100): {image_id, pts, score, polys, rec}; KIE results
    (`vie_categories > 0`) go to one <file_name>.json per image like val.py:39-43.
"""
import json
import os

import torch
import torch.distributed as tdist

from ..utils import dist as udist
from ..utils.misc import decode_seq
from ..utils.nested_tensor import NestedTensor, nested_tensor_from_tensor_list


def build_prompts(args, device='cpu'):
    """val.py:25-33."""
    nb = args.num_bins
    if args.use_char_window_prompt:
        pt = [0, 0, nb - 1, nb - 1, nb, nb + len(args.chars), args.pt_sos_index]
    else:
        pt = [0, 0, nb - 1, nb - 1, args.pt_sos_index]
    mk = lambda v: torch.tensor([v], dtype=torch.long, device=device)  # noqa: E731
    return [mk(pt), mk([args.poly_sos_index]), mk([args.rec_sos_index])]


def decode_pred_seq(index_seqs, prob_seqs, target, args):
    """val.py:70-100: ids -> records in ORIGINAL image coordinates."""
    pt = index_seqs[0]
    if len(pt) % 2 != 0:
        pt = pt[:-1]
    pts = decode_seq(pt, args, 'pt')
    polys = decode_seq(index_seqs[1], args, 'poly')
    recs, confs = decode_seq(index_seqs[2], args, 'rec', prob_seqs)
    h, w = target['orig_size']
    h, w = float(h), float(w)
    scale = torch.tensor([w, h] * 16)
    out = []
    for p, poly, rec, conf in zip(pts, polys, recs, confs):
        out.append({'image_id': target['file_name'],
                    'pts': [[p['point'][0] * w, p['point'][1] * h]],
                    'score': conf,
                    'polys': (poly['polygon'] * scale).reshape(-1, 2).tolist(),
                    'rec': rec['rec']})
    return out


def _as_nested(images):
    """NestedTensor from whatever the caller has: anything with .tensors / .mask (ours OR the reference's class, as its
    dataloader collates), a (B,3,H,W) tensor, or a list of (3,H,W) tensors of possibly different sizes."""
    if hasattr(images, 'tensors'):
        mask = images.mask
        if mask is None:
            t = images.tensors
            mask = torch.zeros(t.shape[0], t.shape[2], t.shape[3], dtype=torch.bool, device=t.device)
        return NestedTensor(images.tensors, mask)
    if isinstance(images, torch.Tensor):
        return NestedTensor(images, torch.zeros(images.shape[0], images.shape[2], images.shape[3], dtype=torch.bool, device=images.device))
    return nested_tensor_from_tensor_list(list(images))


@torch.no_grad()
def predict_raw(model, images, args, orig_sizes=None):
    """-> (per-image raw model outputs exactly as the reference's forward returns them, NestedTensor on the device)."""
    nt = _as_nested(images)
    dev = next(model.parameters()).device
    B = nt.tensors.shape[0]
    has_padding = bool(nt.mask.any())
    nt = nt.to(dev)
    seqs = build_prompts(args)
    if args.infer_vie:
        if orig_sizes is None:
            orig_sizes = [(int(nt.tensors.shape[2]), int(nt.tensors.shape[3]))] * B
        seqs.append([torch.as_tensor(s) for s in orig_sizes])
    return model.infer(nt.tensors, nt.mask, seqs, has_padding=has_padding), nt


@torch.no_grad()
def predict(model, images, args, targets=None, orig_sizes=None):
    """images: list of (3,H,W) tensors, a (B,3,H,W) tensor or a NestedTensor (this package's or the reference's).
    Returns one entry per image: list of records (text spotting), list of tuples (KIE) or []."""
    if orig_sizes is None and targets is not None and args.infer_vie:
        orig_sizes = [t['orig_size'] for t in targets]
    raw, nt = predict_raw(model, images, args, orig_sizes)
    if args.infer_vie:
        return [r if r is not None else [] for r in raw]
    out = []
    for b, r in enumerate(raw):
        if r is None:
            out.append([])
            continue
        tgt = (targets[b] if targets is not None else
               {'file_name': str(b), 'orig_size': (nt.tensors.shape[2], nt.tensors.shape[3])})
        seq_cpu = [t[0].cpu() for t in r[0]]
        out.append(decode_pred_seq(seq_cpu, r[1][0].cpu(), tgt, args))
    return out


@torch.no_grad()
def predict_images(model, images_u8, args, file_names=None, preprocessor=None):
    """Raw images in, records out: uint8 RGB [H, W, 3] arrays / tensors -> the reference's val transform chain on the
    device (utils/preprocess.py: aspect-preserving Pillow-exact resize to test_min_size / test_max_size, ToTensor,
    Normalize, pad + mask) -> the hot path -> records in ORIGINAL image coordinates (val.py:70-100).
    Returns (results per image, preprocessor) so the coefficient tables can be reused by the next call."""
    from ..utils.preprocess import DevicePreprocessor
    dev = next(model.parameters()).device
    if preprocessor is None:
        preprocessor = DevicePreprocessor(args.test_min_size, args.test_max_size, dev)
    imgs = [torch.as_tensor(i).to(dev) for i in images_u8]
    nt, _ = preprocessor(imgs)
    targets = [{'file_name': (file_names[b] if file_names is not None else str(b)),
                'orig_size': (int(im.shape[0]), int(im.shape[1]))} for b, im in enumerate(imgs)]
    return predict(model, nt, args, targets=targets, orig_sizes=[t['orig_size'] for t in targets]), preprocessor


def _meta(t):
    """the (small, picklable) part of a dataloader target that result formatting needs"""
    h, w = t['orig_size']
    return {'file_name': t['file_name'], 'orig_size': (float(h), float(w)), 'dataset_name': t.get('dataset_name', 'results')}


def plan_eos_balance(pred, world, batch_size):
    """Straggler balance for EOS-honouring inference ("expected scaling loss comes from ragged decode length (straggler
    rank), not bandwidth -> balance by sorting / bucketing images by predicted N").  An image's decode time grows with its instance count
    (2 N point steps, then N polygon and N recognition rows); images of one engine call are decoded in lock-step, so a call costs its
    DENSEST image, and a rank costs the sum of its calls -- contiguous shards of a dataset whose dense pages sit together make one rank
    the straggler of every all-gather.
      pred: predicted instance count (any cost proxy) per image, dataset order -- annotation counts of a validation set, the previous
            epoch's decoded counts, a detector's proposals;
      -> plan[rank] = list of batches, each a list of GLOBAL image indices.
    (1) across ranks: longest-processing-time-first -- images by predicted cost, descending, each to the least-loaded rank that still has
        room; shard SIZES stay those of utils/dist.py::shard_range, so every rank makes the same number of engine calls (+-1) and the
        padded all-gather is unchanged;
    (2) inside a rank: by predicted cost, descending, cut into batches of `batch_size`: images that finish together are decoded together.
    Deterministic (ties by index): every rank computes the same plan without communicating."""
    n = len(pred)
    cap = [udist.shard_range(n, r, world)[1] - udist.shard_range(n, r, world)[0] for r in range(world)]
    load = [0.0] * world
    mine = [[] for _ in range(world)]
    for i in sorted(range(n), key=lambda i: (-float(pred[i]), i)):
        r = min((r for r in range(world) if len(mine[r]) < cap[r]), key=lambda r: (load[r], r))
        mine[r].append(i)
        load[r] += float(pred[i])
    bs = max(1, int(batch_size))
    plan = []
    for r in range(world):
        order = sorted(mine[r], key=lambda i: (-float(pred[i]), i))
        plan.append([order[o:o + bs] for o in range(0, len(order), bs)])
    return plan


def plan_cost(plan, pred):
    """lock-step cost model of a plan: per rank, the sum over its engine calls of the call's densest image -> list per rank"""
    return [sum(max(float(pred[i]) for i in b) for b in batches if b) for batches in plan]


def _items_by_index(dataloader, order):
    """the items of `dataloader` with the given GLOBAL indices, in that order, without touching the others (map-style DataLoader that walks
    its dataset in order: re-pointed at Subset(dataset, order); a sequence: indexed)."""
    ds = getattr(dataloader, 'dataset', None)
    if isinstance(dataloader, torch.utils.data.DataLoader) and ds is not None and hasattr(ds, '__getitem__') and \
            (dataloader.batch_size in (1, None)) and len(ds) == len(dataloader) and \
            isinstance(getattr(dataloader, 'sampler', None), torch.utils.data.SequentialSampler):
        kw = {}
        if dataloader.num_workers > 0:
            kw = dict(prefetch_factor=dataloader.prefetch_factor, persistent_workers=dataloader.persistent_workers)
        return torch.utils.data.DataLoader(torch.utils.data.Subset(ds, list(order)), batch_size=dataloader.batch_size, shuffle=False,
                                           num_workers=dataloader.num_workers, collate_fn=dataloader.collate_fn, pin_memory=dataloader.pin_memory,
                                           worker_init_fn=dataloader.worker_init_fn, **kw)
    if isinstance(dataloader, (list, tuple)):
        return [dataloader[i] for i in order]
    raise TypeError('EOS balancing (args.eos_pred_counts) needs a sized, index-addressable loader: a sequence or a DataLoader over a map-style dataset in order')


def _rank_items(dataloader, rank, ws, sharded):
    """This rank's part of the validation set WITHOUT decoding the other ranks' images (every rank used to iterate the
    whole loader and discard what was not its own).  A torch DataLoader with a map-style dataset is re-pointed at
    Subset(dataset, [lo, hi)) (same collate / workers / batch size 1 as the reference's val loader); a plain sequence is
    sliced; a generic iterable is consumed lazily with islice -- skipped items are still produced by the iterable, but never
    held: nothing is materialised with list().  -> (iterable of this rank's items, lo, hi)."""
    import itertools
    if ws == 1 or sharded:
        return dataloader, 0, None
    if not hasattr(dataloader, '__len__'):
        raise TypeError('validate over %d ranks needs a dataloader with a length (to cut contiguous shards); pass a sized loader or '
                        'one that is already rank-sharded (args.dataloader_is_sharded)' % ws)
    n_items = len(dataloader)
    lo, hi = udist.shard_range(n_items, rank, ws)
    ds = getattr(dataloader, 'dataset', None)
    # only a loader that walks its dataset IN ORDER may be re-pointed at a Subset: a full-length non-sequential sampler (a
    # permutation, a weighted sampler) defines the item order itself -- that falls through to islice below
    if isinstance(dataloader, torch.utils.data.DataLoader) and ds is not None and hasattr(ds, '__getitem__') and \
            (dataloader.batch_size in (1, None)) and len(ds) == n_items and \
            isinstance(getattr(dataloader, 'sampler', None), torch.utils.data.SequentialSampler):
        sub = torch.utils.data.Subset(ds, range(lo, hi))
        kw = {}
        if dataloader.num_workers > 0:   # these two are only legal with worker processes
            kw = dict(prefetch_factor=dataloader.prefetch_factor, persistent_workers=dataloader.persistent_workers)
        return torch.utils.data.DataLoader(sub, batch_size=dataloader.batch_size, shuffle=False, num_workers=dataloader.num_workers,
                                           collate_fn=dataloader.collate_fn, pin_memory=dataloader.pin_memory,
                                           worker_init_fn=dataloader.worker_init_fn, **kw), lo, hi
    if isinstance(dataloader, (list, tuple)):
        return dataloader[lo:hi], lo, hi
    return itertools.islice(iter(dataloader), lo, hi), lo, hi


@torch.no_grad()
def validate(model, dataloader, epoch, args, batch_size=1):
    """Drop-in for engine.validate (val.py:11-68), image-sharded over the ranks of torch.distributed.

    dataloader yields (samples, targets) as the reference's does (samples: NestedTensor of 1 or more images, targets:
    list of dicts with file_name / orig_size / dataset_name).  Rank r decodes items [lo, hi) = shard_range(len, r, W)
    -- a loader that is already rank-sharded (DistributedSampler: `args.dataloader_is_sharded`) is consumed whole --
    `batch_size` consecutive items are merged into one engine call (different sizes are padded and masked exactly as the
    reference's collate would), decoded token tensors are packed to fixed size and all-gathered ONCE at the end, and
    rank 0 formats and writes <output_folder>/results/epXXX/<dataset>.json (text spotting) or one <file_name>.json per
    image (KIE).  Returns the records on rank 0 (a list; per-image lists for KIE) and [] elsewhere.
    args.eos_pred_counts (optional: one predicted instance count per dataloader item) switches the contiguous shards for plan_eos_balance's:
    cost-balanced across ranks, homogeneous inside an engine call; the output is the same, in dataset order."""
    model.eval()
    rank, ws = udist.world()
    dev = next(model.parameters()).device
    sharded = bool(getattr(args, 'dataloader_is_sharded', False))
    pred = getattr(args, 'eos_pred_counts', None)
    kie = args.vie_categories > 0
    local_raw, local_meta = [], []
    pend_imgs, pend_tg, pend_key = [], [], []
    if pred is not None and not sharded:
        # balanced shards + homogeneous batches (plan_eos_balance): this rank's images in plan order, engine calls cut where the plan cuts;
        # every meta carries its dataset index so that rank 0 restores dataset order before formatting (the JSON is the unbalanced run's)
        if len(pred) != len(dataloader):
            raise ValueError('args.eos_pred_counts holds %d entries for %d dataloader items' % (len(pred), len(dataloader)))
        my = plan_eos_balance(pred, ws, batch_size)[rank]
        order = [i for b in my for i in b]
        cuts = set()
        acc = 0
        for b in my:
            acc += len(b)
            cuts.add(acc)
        items, lo = _items_by_index(dataloader, order), 0
    else:
        items, lo, hi = _rank_items(dataloader, rank, ws, sharded)
        order, cuts = None, None
    seen = [0]

    def flush():
        if not pend_imgs:
            return
        raw, _ = predict_raw(model, list(pend_imgs), args, [t['orig_size'] for t in pend_tg] if args.infer_vie else None)
        local_raw.extend(raw)
        for t, key in zip(pend_tg, pend_key):
            m = _meta(t)
            m['_idx'] = key
            local_meta.append(m)
        del pend_imgs[:], pend_tg[:], pend_key[:]

    # Sort key of an image = (shard, ITEM index, image inside the item): `lo`, `order` and `cuts` count dataloader ITEMS, and an item may hold
    # several images (counting images there let the ranks' keys overlap and cut the balanced plan in the wrong places).  A loader that
    # is already rank-sharded has no global item index: its ranks' parts stay in rank order (shard = rank), as the gather delivers them.
    for it, (samples, targets) in enumerate(items):
        nt = _as_nested(samples)
        item = order[it] if order is not None else lo + it
        for k, (img, t) in enumerate(zip(nt.unpad_tensors(), targets)):
            pend_imgs.append(img)
            pend_tg.append(t)
            pend_key.append((rank if sharded else 0, item, k))
            if cuts is None and len(pend_imgs) >= max(1, int(batch_size)):
                flush()
        seen[0] += 1
        if cuts is not None and seen[0] in cuts:   # the plan cuts between ITEMS
            flush()
    flush()

    folder = os.path.join(args.output_folder, 'results', 'ep%03d' % epoch) if getattr(args, 'output_folder', None) else None
    if kie:
        # results are host objects (strings, class names): small, gathered as objects
        pairs = [(m, r) for m, r in zip(local_meta, local_raw) if r]
        if ws > 1:
            allp = [None] * ws
            tdist.all_gather_object(allp, pairs)
            pairs = [p for part in allp for p in part]
        if rank != 0:
            return []
        pairs.sort(key=lambda mr: mr[0]['_idx'])
        if folder is not None:
            for m, r in pairs:
                path = os.path.join(folder, m['file_name'] + '.json')
                os.makedirs(os.path.dirname(path), exist_ok=True)
                with open(path, 'w') as f:
                    json.dump(r, f)
        return [r for _, r in pairs]

    # text spotting: fixed-size token tensors, one all-gather per tensor for the whole validation set
    n_loc = torch.tensor([len(local_raw), max([0] + [r[0][0].numel() // 2 for r in local_raw if r is not None])], dtype=torch.int64, device=dev)
    if ws > 1:
        tdist.all_reduce(n_loc, op=tdist.ReduceOp.MAX)
    n_max, inst_max = int(n_loc[0]), max(1, int(n_loc[1]))
    padded = list(local_raw) + [None] * (n_max - len(local_raw))
    ids, probs, n_inst = udist.pack_results(padded, inst_max, args.rec_length, dev)
    ids, probs, n_inst = udist.all_gather_results(ids, probs, n_inst)
    metas = local_meta
    if ws > 1:
        allm = [None] * ws
        tdist.all_gather_object(allm, local_meta)
        metas = allm
    if rank != 0:
        return []
    results, last = [], None
    per_rank = [metas] if ws == 1 else metas
    every = []
    for r_, ms in enumerate(per_rank):
        outs = udist.unpack_results(ids[r_ * n_max:r_ * n_max + len(ms)].cpu(), probs[r_ * n_max:r_ * n_max + len(ms)].cpu(),
                                    n_inst[r_ * n_max:r_ * n_max + len(ms)].cpu())
        every.extend(zip(ms, outs))
    every.sort(key=lambda mo: mo[0]['_idx'])     # dataset order, whatever the sharding was
    for m, o in every:
        last = m
        if o is None:       # the reference skips empty outputs (val.py:36-37)
            continue
        results.extend(decode_pred_seq([t[0] for t in o[0]], o[1][0], m, args))
    if folder is not None and last is not None:
        os.makedirs(folder, exist_ok=True)
        with open(os.path.join(folder, last['dataset_name'] + '.json'), 'w') as f:
            f.write(json.dumps(results, indent=4))
    return results
