"""Test-time image pre-processing on the MI355X (the step right before the hot path).

Mirror of the reference's val pipeline (OCR/OmniParser/dataset/__init__.py:109-113: RandomResize([test_min_size],
test_max_size) -> ToTensor -> Normalize) followed by nested_tensor_from_tensor_list (utils/nested_tensor.py:37-54),
for uint8 RGB images that already sit in device memory (decoded on the host or by a hardware decoder): one
omp_resize_normalize_pad launch per image writes its slice of the padded batch tensor and of the mask.  The resize
is Pillow's bilinear resampler (what torchvision's F.resize calls for PIL images), reproduced bit for bit: this
module builds Pillow's coefficient tables on the host (a few KB per image size, cached), the kernel applies them.
"""
import math

import numpy as np
import torch

from .. import _lib, ops
from .nested_tensor import NestedTensor

PRECISION_BITS = 32 - 8 - 2          # Pillow Resample.c: 8-bit pixels, 22 fractional bits
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def get_size_with_aspect_ratio(image_size, size, max_size=None):
    """reference dataset/transforms.py:275-296; image_size = (w, h); returns (oh, ow)."""
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


def _filter_weights(a, filt):
    """Resample.c bilinear_filter / bicubic_filter (a = -0.5) on an array of (already scaled) distances."""
    a = np.abs(a)
    if filt == 'bilinear':
        return np.where(a < 1.0, 1.0 - a, 0.0)
    if filt == 'bicubic':
        c = -0.5
        inner = ((c + 2.0) * a - (c + 3.0)) * a * a + 1
        outer = (((a - 5) * a + 8) * a - 4) * c
        return np.where(a < 1.0, inner, np.where(a < 2.0, outer, 0.0))
    raise ValueError(filt)


def resize_coeffs(in_size, out_size, filt='bilinear'):
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc for the bilinear (triangle, the OmniParser val transform) or
    bicubic (MGP-STR's word-crop resize) filter over a whole axis, vectorised over the output positions; every
    floating-point operation happens in the order Resample.c performs it (double precision; the row sum is accumulated
    left to right), so the tables are bit-identical.
    -> (ksize, bounds int32 [out, 2] = (first source index, count), coefficients int32 [out, ksize])."""
    scale = float(np.float32(in_size) - np.float32(0.0)) / out_size
    filterscale = max(scale, 1.0)
    support = (1.0 if filt == 'bilinear' else 2.0) * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    xx = np.arange(out_size, dtype=np.float64)
    center = 0.0 + (xx + 0.5) * scale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)          # C (int) cast: truncation
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size) - xmin
    x = np.arange(ksize, dtype=np.int64)[None, :]
    w = _filter_weights(((x + xmin[:, None]).astype(np.float64) - center[:, None] + 0.5) * ss, filt)
    w = np.where(x < xmax[:, None], w, 0.0)
    ww = np.zeros(out_size, dtype=np.float64)
    for j in range(ksize):                                                    # sequential, like the C loop
        ww = np.where(j < xmax, ww + w[:, j], ww)
    k = np.where((ww != 0.0)[:, None] & (x < xmax[:, None]), w / np.where(ww == 0.0, 1.0, ww)[:, None], w)
    v = k * float(1 << PRECISION_BITS)
    kk = np.where(k < 0, (-0.5 + v), (0.5 + v)).astype(np.int64).astype(np.int32)   # (int) cast truncates
    bounds = np.stack([xmin, xmax], axis=1).astype(np.int32)
    return ksize, bounds, kk


def normalize_lut(mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """float32 [3, 256]: ToTensor + Normalize of every possible byte, with the reference's own float32 operations."""
    p = torch.arange(256, dtype=torch.uint8).to(torch.float32).div(255)
    m = torch.tensor(mean, dtype=torch.float32)[:, None]
    s = torch.tensor(std, dtype=torch.float32)[:, None]
    return ((p[None, :] - m) / s).contiguous()


class DevicePreprocessor(object):
    """images (uint8 [H, W, 3] device tensors) -> NestedTensor(tensors fp32 [B,3,Hmax,Wmax], mask bool [B,Hmax,Wmax])."""

    def __init__(self, test_min_size, test_max_size, device, mean=IMAGENET_MEAN, std=IMAGENET_STD):
        self.min_size, self.max_size = test_min_size, test_max_size
        self.device = torch.device(device)
        self.lut = normalize_lut(mean, std).to(self.device)
        self._tables = {}

    def _axis(self, n_in, n_out):
        key = (n_in, n_out)
        if key not in self._tables:
            if n_in == n_out:
                self._tables[key] = (0, None, None)
            else:
                ks, b, k = resize_coeffs(n_in, n_out)
                self._tables[key] = (ks, torch.from_numpy(b).to(self.device), torch.from_numpy(k).to(self.device))
        return self._tables[key]

    def output_size(self, h, w):
        return get_size_with_aspect_ratio((w, h), self.min_size, self.max_size)

    @torch.no_grad()
    def __call__(self, images):
        sizes = []
        for im in images:
            if im.dtype != torch.uint8 or im.dim() != 3 or im.shape[2] != 3 or not im.is_cuda:
                raise ValueError('DevicePreprocessor takes uint8 [H, W, 3] device tensors')
            sizes.append(self.output_size(int(im.shape[0]), int(im.shape[1])))
        B = len(images)
        Hm, Wm = max(s[0] for s in sizes), max(s[1] for s in sizes)
        out = torch.empty(B, 3, Hm, Wm, dtype=torch.float32, device=self.device)
        mask = torch.empty(B, Hm, Wm, dtype=torch.uint8, device=self.device)
        h = _lib.lib()
        for b, (im, (oh, ow)) in enumerate(zip(images, sizes)):
            if im.stride(2) != 1 or im.stride(1) != 3:
                im = im.contiguous()
            ksx, xb, kx = self._axis(int(im.shape[1]), ow)
            ksy, yb, ky = self._axis(int(im.shape[0]), oh)
            rc = h.omp_resize_normalize_pad(ops.ptr(im), im.stride(0), int(im.shape[0]), int(im.shape[1]), ops.ptr(xb), ops.ptr(kx), ksx,
                                            ops.ptr(yb), ops.ptr(ky), ksy, ops.ptr(self.lut), ops.ptr(out[b]), ops.ptr(mask[b]),
                                            oh, ow, Hm, Wm, ops.stream())
            _lib.check(rc, 'omp_resize_normalize_pad')
        return NestedTensor(out, mask.to(torch.bool)), sizes


class CropResizer(object):
    """Word crops for the recogniser: axis-aligned boxes of uint8 [H, W, 3] device images -> fp32
    [N, 3, out_h, out_w] in [0, 1], exactly what MGP-STR's AlignCollate feeds its model for each crop
    (OCR/MGP-STR/dataset.py:462: image.resize((imgW, imgH), Image.BICUBIC) then ToTensor), bit for bit: Pillow's
    bicubic coefficient tables are built here, omp_resize_normalize_pad applies them to the crop in place (the
    source pointer is offset to the box, the pitch stays the image's) -- no crop copy, no host round trip."""

    def __init__(self, device, out_h=32, out_w=128, filt='bicubic'):
        self.device = torch.device(device)
        self.out_h, self.out_w, self.filt = out_h, out_w, filt
        p = torch.arange(256, dtype=torch.uint8).to(torch.float32).div(255)     # ToTensor: p / 255 in float32
        self.lut = p[None, :].expand(3, 256).contiguous().to(self.device)
        self._tables = {}

    def _axis(self, n_in, n_out):
        key = (n_in, n_out)
        if key not in self._tables:
            if len(self._tables) > 4096:
                self._tables.clear()
            if n_in == n_out:
                self._tables[key] = (0, None, None)
            else:
                ks, b, k = resize_coeffs(n_in, n_out, self.filt)
                self._tables[key] = (ks, torch.from_numpy(b).to(self.device), torch.from_numpy(k).to(self.device))
        return self._tables[key]

    @torch.no_grad()
    def __call__(self, images, boxes):
        """images: list of uint8 [H, W, 3] device tensors; boxes: list of (image index, x0, y0, x1, y1) integer pixel
        boxes (x1, y1 exclusive, already clipped, at least 1 x 1).  -> fp32 [N, 3, out_h, out_w]."""
        N = len(boxes)
        out = torch.empty(N, 3, self.out_h, self.out_w, dtype=torch.float32, device=self.device)
        h = _lib.lib()
        for n, (bi, x0, y0, x1, y1) in enumerate(boxes):
            im = images[bi]
            if im.dtype != torch.uint8 or im.dim() != 3 or im.shape[2] != 3 or not im.is_cuda or not im.is_contiguous():
                raise ValueError('CropResizer takes contiguous uint8 [H, W, 3] device tensors')
            cw, ch = x1 - x0, y1 - y0
            if cw < 1 or ch < 1 or x0 < 0 or y0 < 0 or x1 > im.shape[1] or y1 > im.shape[0]:
                raise ValueError('bad crop box %s for image %s' % ((x0, y0, x1, y1), tuple(im.shape)))
            ksx, xb, kx = self._axis(cw, self.out_w)
            ksy, yb, ky = self._axis(ch, self.out_h)
            src = ctypes_ptr(im, (y0 * im.shape[1] + x0) * 3)
            rc = h.omp_resize_normalize_pad(src, im.stride(0), ch, cw, ops.ptr(xb), ops.ptr(kx), ksx, ops.ptr(yb), ops.ptr(ky), ksy,
                                            ops.ptr(self.lut), ops.ptr(out[n]), None, self.out_h, self.out_w, self.out_h, self.out_w,
                                            ops.stream())
            _lib.check(rc, 'omp_resize_normalize_pad')
        return out


def ctypes_ptr(t, byte_offset=0):
    import ctypes
    return ctypes.c_void_p(t.data_ptr() + byte_offset)
