"""Image-sharded multi-GPU inference: one process per GPU over torch.distributed.

The reference only knows DDP training (utils/dist.py:32-46: env:// NCCL rendezvous) and has no
distributed inference: `validate` would write the same JSON from every rank (engine/val.py:64-68).
Here images are split contiguously over ranks and ONE all-gather of fixed-size padded token
tensors per batch returns every rank the full result (RCCL over xGMI on MI355X -- backend 'nccl'
under PyTorch-ROCm; 'gloo' with CPU tensors in the tests).  Nothing is communicated per decode step.
"""
import os

import torch
import torch.distributed as dist


def init_distributed_mode(args=None, backend=None):
    """Rendezvous from RANK / WORLD_SIZE / LOCAL_RANK like the reference's init_distributed_mode."""
    if 'RANK' not in os.environ or 'WORLD_SIZE' not in os.environ:
        if args is not None:
            args.distributed = False
        return False
    rank, world, local = int(os.environ['RANK']), int(os.environ['WORLD_SIZE']), int(os.environ.get('LOCAL_RANK', 0))
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if backend is None:
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'
    if backend == 'nccl':
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, init_method='env://')
    if args is not None:
        args.distributed, args.rank, args.local_rank = True, rank, local
    return True


def world():
    return (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)


def shard_range(n_items, rank, world_size):
    """Contiguous shard [lo, hi) of n_items for `rank`; the first n_items % world ranks get one more."""
    base, extra = divmod(n_items, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def pack_results(results, max_instances, rec_length, device):
    """Text-spotting results of a local batch -> fixed-size tensors:
    ids int32 [b, Nmax, 2+32+rec_length] (pt | poly | rec), probs f32 [b, Nmax, rec_length], n int32 [b]."""
    b = len(results)
    ids = torch.zeros(b, max_instances, 34 + rec_length, dtype=torch.int32, device=device)
    probs = torch.zeros(b, max_instances, rec_length, dtype=torch.float32, device=device)
    n_inst = torch.zeros(b, dtype=torch.int32, device=device)
    for i, r in enumerate(results):
        if r is None:
            continue
        (pt, poly, rec), (pr,) = r
        n = min(max_instances, pt.numel() // 2)
        ids[i, :n, 0:2] = pt.reshape(-1, 2)[:n].to(device, torch.int32)
        ids[i, :n, 2:34] = poly.reshape(-1, 32)[:n].to(device, torch.int32)
        ids[i, :n, 34:] = rec.reshape(-1, rec_length)[:n].to(device, torch.int32)
        probs[i, :n] = pr[:n].to(device)
        n_inst[i] = n
    return ids, probs, n_inst


def unpack_results(ids, probs, n_inst):
    """Inverse of pack_results (per image: the reference's return structure, or None)."""
    out = []
    for i in range(ids.shape[0]):
        n = int(n_inst[i])
        if n == 0:
            out.append(None)
            continue
        t = ids[i, :n].long()
        out.append(([t[:, 0:2].reshape(1, -1), t[:, 2:34].reshape(1, -1), t[:, 34:].unsqueeze(0)], [probs[i, :n]]))
    return out


def pack_payload(ids, probs, n_inst=None):
    """ids int32 [b, N, T], probs fp32 [b, N, L] (and n_inst int32 [b]) -> ONE int32 tensor [b, N * (T + L) (+ 1)]: the probabilities
    travel as their bit patterns, so an engine call costs a single collective (its latency term, not its bandwidth, is what counts)."""
    b = ids.shape[0]
    parts = [ids.reshape(b, -1), probs.reshape(b, -1).contiguous().view(torch.int32)]
    if n_inst is not None:
        parts.append(n_inst.reshape(b, 1).to(torch.int32))
    return torch.cat(parts, dim=1).contiguous()


def unpack_payload(buf, ids_shape, probs_shape, with_n=True):
    """Inverse of pack_payload for a gathered buffer [B, ...]: ids_shape / probs_shape are the per-image shapes (N, T) / (N, L)."""
    B = buf.shape[0]
    ni, npb = ids_shape[0] * ids_shape[1], probs_shape[0] * probs_shape[1]
    ids = buf[:, :ni].reshape((B,) + tuple(ids_shape))
    probs = buf[:, ni:ni + npb].contiguous().view(torch.float32).reshape((B,) + tuple(probs_shape))
    n_inst = buf[:, ni + npb].contiguous() if with_n else None
    return ids, probs, n_inst


def all_gather_results(ids, probs, n_inst):
    """ONE all-gather per engine call (ids, probabilities and instance counts in one int32 payload); every rank must contribute the
    same local batch size."""
    rank, ws = world()
    if ws == 1:
        return ids, probs, n_inst
    payload = pack_payload(ids, probs, n_inst)
    buf = torch.empty((ws * payload.shape[0], payload.shape[1]), dtype=torch.int32, device=payload.device)
    dist.all_gather_into_tensor(buf, payload)
    return unpack_payload(buf, ids.shape[1:], probs.shape[1:])
