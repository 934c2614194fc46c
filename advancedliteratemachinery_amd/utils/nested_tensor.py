"""Padded image batch container.

Same contract as the reference's `NestedTensor` / `nested_tensor_from_tensor_list`
(OCR/OmniParser/utils/nested_tensor.py:7-54): `tensors` is (B, 3, Hmax, Wmax) zero padded,
`mask` is (B, Hmax, Wmax) bool with True on padding.
"""
from typing import List, Optional

import torch
from torch import Tensor


class NestedTensor(object):
    def __init__(self, tensors: Tensor, mask: Optional[Tensor]):
        self.tensors = tensors
        self.mask = mask

    def to(self, device):
        m = self.mask.to(device) if self.mask is not None else None
        return NestedTensor(self.tensors.to(device), m)

    def decompose(self):
        return self.tensors, self.mask

    def unpad_tensors(self):
        out = []
        for img, m in zip(self.tensors, self.mask):
            w = m.shape[1] - int(m[0, :].sum())
            h = m.shape[0] - int(m[:, 0].sum())
            out.append(img[:, :h, :w])
        return out

    def __repr__(self):
        return str(self.tensors)


def nested_tensor_from_tensor_list(tensor_list: List[Tensor]) -> NestedTensor:
    if tensor_list[0].ndim != 3:
        raise ValueError('not supported')
    c = tensor_list[0].shape[0]
    hmax = max(int(t.shape[1]) for t in tensor_list)
    wmax = max(int(t.shape[2]) for t in tensor_list)
    b = len(tensor_list)
    ref = tensor_list[0]
    batch = torch.zeros((b, c, hmax, wmax), dtype=ref.dtype, device=ref.device)
    mask = torch.ones((b, hmax, wmax), dtype=torch.bool, device=ref.device)
    for i, img in enumerate(tensor_list):
        _, h, w = img.shape
        batch[i, :, :h, :w].copy_(img)
        mask[i, :h, :w] = False
    return NestedTensor(batch, mask)
