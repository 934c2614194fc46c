"""Image pre-processing: the oracle is pinned to Pillow itself (the reference's resize IS
PIL.Image.resize through torchvision), the engine's host-side coefficient tables to the oracle, the HIP kernel to
the oracle bit for bit (gpu)."""
import numpy as np
import pytest
import torch

from advancedliteratemachinery_amd.utils import preprocess as PP
from oracle import preprocess_ref as P

CASES = [(37, 53, 74, 106), (100, 80, 37, 29), (64, 64, 64, 31), (50, 70, 91, 70), (200, 300, 67, 100), (33, 47, 33, 47),
         (480, 640, 300, 400), (17, 400, 9, 211), (301, 157, 1000, 521)]


def _img(H, W, seed):
    return np.random.default_rng(seed).integers(0, 256, size=(H, W, 3), dtype=np.uint8)


def test_oracle_resize_is_bit_identical_to_pillow():
    Image = pytest.importorskip('PIL.Image')
    for i, (H, W, oh, ow) in enumerate(CASES):
        img = _img(H, W, i)
        ref = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BILINEAR))
        assert np.array_equal(P.resize_bilinear_u8(img, oh, ow), ref), (H, W, oh, ow)


def test_size_rule_matches_reference_arithmetic():
    # transforms.py:275-296 on the shapes of SURVEY.md 8 (test.sh: min 1024 / max 1824, and the 1920 default)
    assert P.get_size_with_aspect_ratio((1280, 720), 1024, 1824) == (1024, 1820)     # (oh, ow)
    assert P.get_size_with_aspect_ratio((720, 1280), 1024, 1824) == (1820, 1024)
    assert P.get_size_with_aspect_ratio((4000, 1000), 1024, 1824) == (456, 1824)
    assert P.get_size_with_aspect_ratio((1024, 2000), 1024, 1824) == (1824, 934)
    assert P.get_size_with_aspect_ratio((800, 800), 800, 1333) == (800, 800)
    for sz in ((1280, 720), (333, 500), (1000, 4000), (640, 640)):
        assert PP.get_size_with_aspect_ratio(sz, 1024, 1824) == P.get_size_with_aspect_ratio(sz, 1024, 1824)


def test_engine_coefficient_tables_equal_oracle():
    for n_in, n_out in [(53, 106), (80, 29), (300, 100), (640, 400), (400, 211), (157, 521), (1920, 1024), (1000, 1824), (7, 3)]:
        ks0, b0, k0 = P.precompute_coeffs(n_in, n_out)
        ks1, b1, k1 = PP.resize_coeffs(n_in, n_out)
        assert ks0 == ks1 and np.array_equal(b0, b1) and np.array_equal(k0, k1), (n_in, n_out)


def test_normalize_lut_equals_reference_ops():
    lut = PP.normalize_lut()
    img = _img(5, 7, 3)
    ref = P.to_tensor_normalize(img)                       # numpy float32 restatement
    got = torch.stack([lut[c][torch.from_numpy(img[:, :, c].astype(np.int64))] for c in range(3)])
    assert np.array_equal(got.numpy(), ref)


def test_batch_padding_and_mask():
    imgs = [_img(40, 60, 1), _img(90, 30, 2)]
    t, m, sizes = P.preprocess_batch(imgs, 32, 64)
    assert t.shape[0] == 2 and t.shape[2] == max(s[0] for s in sizes) and t.shape[3] == max(s[1] for s in sizes)
    for b, (oh, ow) in enumerate(sizes):
        assert not m[b, :oh, :ow].any() and m[b].sum() == m[b].size - oh * ow
        assert (t[b][:, oh:, :] == 0).all() and (t[b][:, :, ow:] == 0).all()


@pytest.mark.gpu
def test_device_preprocessor_is_bit_exact():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    pre = PP.DevicePreprocessor(48, 100, 'cuda:0')
    imgs = [_img(40, 60, 1), _img(90, 30, 2), _img(48, 48, 3), _img(200, 333, 4), _img(31, 64, 5)]
    ref_t, ref_m, ref_sizes = P.preprocess_batch(imgs, 48, 100)
    nt, sizes = pre([torch.from_numpy(i).cuda() for i in imgs])
    assert sizes == ref_sizes
    assert np.array_equal(nt.mask.cpu().numpy(), ref_m)
    assert np.array_equal(nt.tensors.cpu().numpy(), ref_t)          # float32, bit for bit
    # a production-sized case: 720x1280 -> 1024x1820 through the same path, against Pillow directly
    Image = pytest.importorskip('PIL.Image')
    big = _img(720, 1280, 9)
    pre2 = PP.DevicePreprocessor(1024, 1824, 'cuda:0')
    nt2, sz2 = pre2([torch.from_numpy(big).cuda()])
    pil = np.asarray(Image.fromarray(big).resize((sz2[0][1], sz2[0][0]), Image.BILINEAR))
    assert np.array_equal(nt2.tensors[0].cpu().numpy(), P.to_tensor_normalize(pil))
