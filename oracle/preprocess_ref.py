"""CPU restatement of the reference's TEST-TIME image pre-processing.  TEST INFRASTRUCTURE
ONLY (tests/ and nothing on the product path import it).

What it follows (OCR/OmniParser/):
  dataset/__init__.py:109-113     val pipeline = RandomResize([test_min_size], test_max_size) -> ToTensor -> Normalize
  dataset/transforms.py:249-298   RandomResize: get_size_with_aspect_ratio + torchvision F.resize(PIL image, (oh, ow))
  dataset/transforms.py:312-339   ToTensor (uint8 HWC -> float CHW / 255), Normalize (ImageNet mean / std)
  utils/nested_tensor.py:37-54    zero-pad to the batch maximum, mask = True on padding
torchvision's F.resize of a PIL image is `img.resize((ow, oh), PIL.Image.BILINEAR)` (third-party dependencies:
torchvision, not installed here; Pillow, installed).  The resampler restated below is Pillow's
src/libImaging/Resample.c (ImagingResample, 8-bit path): separable triangle filter whose support grows with
the down-scale factor, coefficients normalised in double then quantised to 22 fractional bits, horizontal pass
first with the intermediate image rounded to uint8, then the vertical pass.
Pinned: tests/test_preprocess.py compares `resize_bilinear_u8` with PIL.Image.resize bit for bit.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2
MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)
STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)


def get_size_with_aspect_ratio(image_size, size, max_size=None):
    """transforms.py:275-296; image_size = (w, h) as PIL reports it; returns (oh, ow)."""
    w, h = image_size
    if max_size is not None:
        min_original_size = float(min((w, h)))
        max_original_size = float(max((w, h)))
        if max_original_size / min_original_size * size > max_size:
            size = int(round(max_size * min_original_size / max_original_size))
    if (w <= h and w == size) or (h <= w and h == size):
        return (h, w)
    if w < h:
        ow = size
        oh = int(size * h / w)
    else:
        oh = size
        ow = int(size * w / h)
    return (oh, ow)


def _bilinear(a):
    a = -a if a < 0.0 else a
    return 1.0 - a if a < 1.0 else 0.0


def _bicubic(x):
    """Resample.c bicubic_filter (a = -0.5), the filter MGP-STR's AlignCollate resizes word crops with
    (OCR/MGP-STR/dataset.py:454,462: image.resize(..., Image.BICUBIC))."""
    a = -0.5
    x = -x if x < 0.0 else x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


FILTERS = {'bilinear': (_bilinear, 1.0), 'bicubic': (_bicubic, 2.0)}


def precompute_coeffs(in_size, out_size, filt='bilinear'):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc over the whole axis.
    -> (ksize, bounds int32 [out, 2] = (first source index, count), kk int32 [out, ksize])."""
    fn, fsupport = FILTERS[filt]
    scale = float(np.float32(in_size) - np.float32(0.0)) / out_size
    filterscale = max(scale, 1.0)
    support = fsupport * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = np.zeros(ksize, dtype=np.float64)
        ww = 0.0
        for x in range(xmax):
            w = fn((x + xmin - center + 0.5) * ss)
            k[x] = w
            ww += w
        if ww != 0.0:
            k[:xmax] /= ww
        for x in range(ksize):
            v = k[x] * (1 << PRECISION_BITS)
            kk[xx, x] = int(-0.5 + v) if k[x] < 0 else int(0.5 + v)
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, kk


def _clip8(acc):
    return np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resize_bilinear_u8(img, oh, ow):
    """img uint8 [H, W, C] -> uint8 [oh, ow, C], bit-identical to PIL.Image.fromarray(img).resize((ow, oh), BILINEAR)."""
    return resize_u8(img, oh, ow, 'bilinear')


def resize_u8(img, oh, ow, filt='bilinear'):
    """Pillow's two-pass 8-bit resampler with the given filter ('bilinear' | 'bicubic')."""
    H, W, _ = img.shape
    if (H, W) == (oh, ow):
        return img.copy()          # Image.resize returns a copy when nothing changes
    need_h, need_v = ow != W, oh != H
    _, bh, kh = precompute_coeffs(W, ow, filt)
    _, bv, kv = precompute_coeffs(H, oh, filt)
    src = img.astype(np.int64)
    half = 1 << (PRECISION_BITS - 1)
    if need_h:
        y0, y1 = int(bv[0, 0]), int(bv[-1, 0] + bv[-1, 1])      # only the rows the vertical pass reads
        rows = src[y0:y1]
        tmp = np.empty((rows.shape[0], ow, img.shape[2]), dtype=np.uint8)
        for xx in range(ow):
            x0, n = int(bh[xx, 0]), int(bh[xx, 1])
            acc = half + (rows[:, x0:x0 + n, :] * kh[xx, :n].astype(np.int64)[None, :, None]).sum(axis=1)
            tmp[:, xx, :] = _clip8(acc)
        src = tmp.astype(np.int64)
        bv = bv.copy()
        bv[:, 0] -= y0
    if not need_v:
        return src.astype(np.uint8)
    out = np.empty((oh, src.shape[1], img.shape[2]), dtype=np.uint8)
    for yy in range(oh):
        y0, n = int(bv[yy, 0]), int(bv[yy, 1])
        acc = half + (src[y0:y0 + n] * kv[yy, :n].astype(np.int64)[:, None, None]).sum(axis=0)
        out[yy] = _clip8(acc)
    return out


def to_tensor_normalize(img_u8):
    """ToTensor + Normalize (transforms.py:312-322): float32 CHW, ((p / 255) - mean) / std in float32."""
    x = img_u8.astype(np.float32) / np.float32(255.0)
    x = (x - MEAN) / STD
    return np.ascontiguousarray(x.transpose(2, 0, 1))


def preprocess_batch(images_u8, test_min_size, test_max_size):
    """list of uint8 [H, W, 3] -> (tensors float32 [B, 3, Hmax, Wmax], mask bool [B, Hmax, Wmax], sizes [(oh, ow)])
    exactly as the val pipeline + nested_tensor_from_tensor_list produce them."""
    outs, sizes = [], []
    for im in images_u8:
        oh, ow = get_size_with_aspect_ratio((im.shape[1], im.shape[0]), test_min_size, test_max_size)
        outs.append(to_tensor_normalize(resize_bilinear_u8(im, oh, ow)))
        sizes.append((oh, ow))
    Hm, Wm = max(s[0] for s in sizes), max(s[1] for s in sizes)
    tensors = np.zeros((len(outs), 3, Hm, Wm), dtype=np.float32)
    mask = np.ones((len(outs), Hm, Wm), dtype=bool)
    for b, (t, (oh, ow)) in enumerate(zip(outs, sizes)):
        tensors[b, :, :oh, :ow] = t
        mask[b, :oh, :ow] = False
    return tensors, mask, sizes
