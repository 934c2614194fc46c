"""Batched inference driver: the MI355X replacement for the reference's batch-1 `validate` loop
(OCR/OmniParser/engine/val.py:11-100), plus a `predict()` convenience API.

  * prompts are built exactly like val.py:25-33;
  * B images per call (the reference asserts 1), optionally image-sharded over ranks with one
    all-gather of the decoded sequences (utils/dist.py);
  * results are formatted like decode_pred_seq (val.py:70-100): {image_id, pts, score, polys, rec}.
"""
import json
import os

import torch

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


@torch.no_grad()
def predict(model, images, args, targets=None, orig_sizes=None):
    """images: list of (3,H,W) tensors, a (B,3,H,W) tensor or a NestedTensor.
    Returns one entry per image: list of records (text spotting), list of tuples (KIE) or []."""
    if isinstance(images, NestedTensor):
        nt = images
    elif isinstance(images, torch.Tensor):
        nt = NestedTensor(images, torch.zeros(images.shape[0], images.shape[2], images.shape[3], dtype=torch.bool))
    else:
        nt = nested_tensor_from_tensor_list(list(images))
    dev = next(model.parameters()).device
    B = nt.tensors.shape[0]
    has_padding = bool(nt.mask.any())
    nt = nt.to(dev)
    seqs = build_prompts(args)
    if args.infer_vie:
        if orig_sizes is None:
            orig_sizes = [(int(nt.tensors.shape[2]), int(nt.tensors.shape[3]))] * B
        seqs.append([torch.tensor(s) for s in orig_sizes])
    raw = model.infer(nt.tensors, nt.mask, seqs, has_padding=has_padding)
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


@torch.no_grad()
def validate(model, dataloader, epoch, args, batch_size=None):
    """Drop-in for engine.validate: iterates (samples, targets) like the reference dataloader yields them,
    writes <output_folder>/results/epXXX/<dataset>.json on rank 0 (the reference writes from every rank)."""
    model.eval()
    rank, ws = udist.world()
    results = []
    last = None
    for samples, targets in dataloader:
        recs = predict(model, samples, args, targets=targets,
                       orig_sizes=[t['orig_size'] for t in targets] if args.infer_vie else None)
        for r in recs:
            results.extend(r)
        last = targets[0]
    if rank == 0 and last is not None and args.vie_categories == 0 and args.output_folder:
        folder = os.path.join(args.output_folder, 'results', 'ep%03d' % epoch)
        os.makedirs(folder, exist_ok=True)
        with open(os.path.join(folder, last.get('dataset_name', 'results') + '.json'), 'w') as f:
            f.write(json.dumps(results, indent=4))
    return results
