"""TEST INFRASTRUCTURE -- CPU restatement of the two-stage chain out of the reference's own pieces:
OmniParser eval forward (oracle/omniparser_ref.py) -> records like engine/val.py:70-100 -> polygon bounding boxes
(min / max over the points, transformer.py:186-196) -> PIL crop + `resize((128, 32), Image.BICUBIC)` + ToTensor
(OCR/MGP-STR/dataset.py:462, the real Pillow) -> MGP-STR forward and fused decoding (oracle/mgp_str_ref.py).
Only tests/ import this."""
import math

import numpy as np
import torch
from PIL import Image

from oracle import mgp_str_ref as R
from oracle import omniparser_ref as O
from oracle import preprocess_ref as P


def records_from_output(out, args, orig_hw):
    """engine/val.py:70-100 on the oracle's raw output: [(pts, polys(16x2 in pixels), rec ids)]"""
    if out is None:
        return []
    h, w = orig_hw
    nb = float(args.num_bins)
    pt = out[0][0].reshape(-1, 2)
    poly = out[0][1].reshape(-1, 32).float() / nb * torch.tensor([w, h] * 16, dtype=torch.float32)
    recs = []
    for i in range(pt.shape[0]):
        recs.append(dict(polys=poly[i].reshape(-1, 2).tolist(), rec_ids=out[0][2][0][i].tolist()))
    return recs


def box(polys, w, h):
    xs, ys = [p[0] for p in polys], [p[1] for p in polys]
    x0 = max(0, min(int(math.floor(min(xs))), w - 1))
    y0 = max(0, min(int(math.floor(min(ys))), h - 1))
    return x0, y0, max(x0 + 1, min(int(math.ceil(max(xs))), w)), max(y0 + 1, min(int(math.ceil(max(ys))), h))


def chain(sd_omni, args, depths, sd_mgp, cfg_mgp, images_u8, min_size, max_size):
    """-> per image list of dict(box, char_ids, bpe_ids, wp_ids, conf, choice)"""
    tens, mask, sizes = P.preprocess_batch(images_u8, min_size, max_size)
    seqs = O.default_prompts(args)
    results = []
    for b, im in enumerate(images_u8):
        H, W = im.shape[0], im.shape[1]
        with torch.no_grad():
            out = O.forward(sd_omni, args, torch.from_numpy(tens[b:b + 1]), torch.from_numpy(mask[b:b + 1]), seqs, depths=depths)
        recs = records_from_output(out, args, (H, W))
        crops = []
        for r in recs:
            r['box'] = box(r['polys'], W, H)
            x0, y0, x1, y1 = r['box']
            c = Image.fromarray(im).crop((x0, y0, x1, y1)).resize((cfg_mgp['img'][1], cfg_mgp['img'][0]), Image.BICUBIC)
            crops.append(torch.from_numpy(np.asarray(c).astype(np.float32) / np.float32(255.0)).permute(2, 0, 1))
        if crops:
            with torch.no_grad():
                _, ch, bp, wp = R.forward(sd_mgp, cfg_mgp, torch.stack(crops))
            for r, d in zip(recs, R.decode(ch, bp, wp)):
                r.update(d)
        results.append(recs)
    return results
