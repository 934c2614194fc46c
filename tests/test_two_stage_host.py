"""Host-side pieces of the two-stage pipeline (engine/two_stage.py) on the CPU: polygon -> crop box arithmetic against the
oracle's restatement, bicubic coefficient tables against Pillow itself (through the oracle resampler, which
tests/test_preprocess.py pins to PIL bit for bit)."""
import numpy as np
import torch
from PIL import Image

from advancedliteratemachinery_amd.engine.two_stage import polygon_box
from advancedliteratemachinery_amd.utils.preprocess import resize_coeffs
from oracle import preprocess_ref as P
from oracle import two_stage_ref as T


def test_polygon_box_matches_oracle_and_stays_inside_the_image():
    rng = np.random.RandomState(0)
    for _ in range(200):
        w, h = int(rng.randint(8, 300)), int(rng.randint(8, 300))
        pts = (rng.rand(16, 2) * np.array([w * 1.2, h * 1.2]) - np.array([w * 0.1, h * 0.1])).tolist()
        b = polygon_box(pts, w, h)
        assert b == T.box(pts, w, h)
        x0, y0, x1, y1 = b
        assert 0 <= x0 < x1 <= w and 0 <= y0 < y1 <= h
    assert polygon_box([[5.2, 3.9]] * 16, 20, 10) == (5, 3, 6, 4)          # degenerate polygon: one pixel
    assert polygon_box([[-4.0, -2.0], [50.0, 40.0]] * 8, 20, 10) == (0, 0, 20, 10)


def test_bicubic_oracle_is_pillow_and_engine_tables_equal_oracle():
    rng = np.random.RandomState(1)
    for (H, W) in ((37, 91), (200, 33), (11, 300), (5, 7), (32, 128)):
        img = rng.randint(0, 256, (H, W, 3), dtype=np.uint8)
        ref = np.asarray(Image.fromarray(img).resize((128, 32), Image.BICUBIC))
        assert np.array_equal(P.resize_u8(img, 32, 128, 'bicubic'), ref)
        for n_in, n_out in ((W, 128), (H, 32)):
            if n_in != n_out:
                a, b = P.precompute_coeffs(n_in, n_out, 'bicubic'), resize_coeffs(n_in, n_out, 'bicubic')
                assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
