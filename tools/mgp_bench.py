"""MGP-STR forward throughput on one GPU (BASELINE config 5: ViT-B patch4 32x128, batch 512 cropped words, bf16).
A parity-test configuration, not the bench line (bench.py measures config 2); this prints words/s and the
achieved TFLOP/s against the 49.8 GFLOP/word of SURVEY.md 8(d).   python tools/mgp_bench.py [batch] [iters]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from advancedliteratemachinery_amd.model.mgp_str import MGPSTR  # noqa: E402
from advancedliteratemachinery_amd.utils import synthetic as R  # noqa: E402  (seeded procedural checkpoint: data only)


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    c = R.mgp_cfg()
    sd = R.make_mgp_state_dict(c, seed=0)
    model = MGPSTR(engine_dtype='bf16')
    model.load_reference_state_dict({'module.' + k: v for k, v in sd.items()})
    model = model.to('cuda:0')
    img = (torch.rand(B, 3, 32, 128, generator=torch.Generator().manual_seed(1)) * 2 - 1).to('cuda:0')
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        model(img)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            outs = model(img)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
        # phase split
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record()
        x, Bn, T = model.encode(img)
        ev[1].record()
        for n in ('char', 'bpe', 'wp'):
            model._a3_head(x, Bn, T, n, False)
        ev[2].record()
        torch.cuda.synchronize()
    print('MGP-STR bf16 B=%d: %.1f ms/forward, %.0f words/s, %.1f TFLOP/s (49.8 GFLOP/word); encoder %.1f ms, A3+heads %.1f ms; logits %s'
          % (B, dt * 1e3, B / dt, B * 49.8e9 / dt / 1e12, ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2]),
             [tuple(o.shape) for o in outs]), flush=True)


if __name__ == '__main__':
    main()
