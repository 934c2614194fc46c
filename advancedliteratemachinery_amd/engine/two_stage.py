"""Two-stage reading: OmniParser detections -> word crops on the device -> MGP-STR recogniser ->
fused multi-granularity decoding.

The reference ships the two models as separate projects and no glue between them, so the interfaces on both sides are
the reference's own: detections are OmniParser's records in ORIGINAL image coordinates (engine/val.py:70-100,
`polys` = 16 points), a word crop is the axis-aligned bounding box of its polygon (the rectangle the reference's KIE
path derives from a polygon, transformer.py:186-196: min / max over the points), and the recogniser sees each crop the
way MGP-STR's own evaluation does (OCR/MGP-STR/dataset.py:462: `image.resize((128, 32), Image.BICUBIC)` + ToTensor;
test_final.py:145-240: greedy ids, max-softmax confidences cumprod-ed to the first EOS, the most confident of the
char / BPE / WordPiece heads wins).  Everything between the uint8 image and the token ids runs on the MI355X: no crop
ever visits the host.
"""
import math

import torch

from ..utils.preprocess import CropResizer
from .inference import predict_images


def polygon_box(polys, width, height):
    """16 (x, y) points in original-image pixels -> integer box (x0, y0, x1, y1), x1 / y1 exclusive, clipped to the image and
    at least one pixel wide and high (min / max over the points as transformer.py:186-196, floor / ceil to whole pixels)."""
    xs = [p[0] for p in polys]
    ys = [p[1] for p in polys]
    x0 = max(0, min(int(math.floor(min(xs))), width - 1))
    y0 = max(0, min(int(math.floor(min(ys))), height - 1))
    x1 = max(x0 + 1, min(int(math.ceil(max(xs))), width))
    y1 = max(y0 + 1, min(int(math.ceil(max(ys))), height))
    return x0, y0, x1, y1


@torch.no_grad()
def recognize_crops(mgp_model, images_u8, boxes, resizer=None, chunk=512):
    """boxes: (image index, x0, y0, x1, y1).  -> list of MGPSTR.recognize results (one dict per box)."""
    if not boxes:
        return [], resizer
    dev = images_u8[0].device
    if resizer is None:
        c = mgp_model.cfg
        resizer = CropResizer(dev, c['img'][0], c['img'][1])
    out = []
    for i in range(0, len(boxes), chunk):
        batch = resizer(images_u8, boxes[i:i + chunk])
        out.extend(mgp_model.recognize(batch))
    return out, resizer


@torch.no_grad()
def spot_and_recognize(omni_model, mgp_model, images_u8, args, file_names=None, preprocessor=None, resizer=None):
    """uint8 RGB [H, W, 3] images -> per image the OmniParser records, each extended by the recogniser's reading of its
    crop: `box` (x0, y0, x1, y1), `mgp_text` (character-head string up to its EOS), `mgp_conf` (char, bpe, wp),
    `mgp_choice` (0 char / 1 bpe / 2 wp / -1 none: the most confident head, test_final.py:172-236) and the raw ids of
    the three heads (the BPE / WordPiece STRINGS need the GPT-2 / BERT vocabulary files, which are not in the
    reference tree).  Returns (results, preprocessor, resizer) so the cached tables can be reused."""
    dev = next(omni_model.parameters()).device
    imgs = [torch.as_tensor(i).to(dev).contiguous() for i in images_u8]
    records, preprocessor = predict_images(omni_model, imgs, args, file_names=file_names, preprocessor=preprocessor)
    boxes, owner = [], []
    for b, recs in enumerate(records):
        h, w = int(imgs[b].shape[0]), int(imgs[b].shape[1])
        for r in recs:
            r['box'] = polygon_box(r['polys'], w, h)
            boxes.append((b,) + r['box'])
            owner.append(r)
    reads, resizer = recognize_crops(mgp_model, imgs, boxes, resizer)
    for r, m in zip(owner, reads):
        r['mgp_text'], r['mgp_conf'], r['mgp_choice'] = m['char_text'], m['conf'], m['choice']
        r['mgp_ids'] = dict(char=m['char_ids'], bpe=m['bpe_ids'], wp=m['wp_ids'])
    return records, preprocessor, resizer
