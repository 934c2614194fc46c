"""The two decoder cross-attention kernels in isolation at the shapes of one bench.py engine call (default: 128
images, M = 4096 memory tokens, bf16), for the rocprofv3 --pmc passes that feed roofline.traffic when profiling the
whole bench at that size is impractical (PMC mode serialises ~16 k dispatches per pass).  Launch mix 22 : 10 =
the bench's 3752 : 1708 (point-decoder launches with 1 row per image : polygon / recognition launches with 64).
    python tools/cross_pmc.py [images] [split]      split: the parity engine's kernels (fp32 q / out over split-bf16 plane slabs)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from advancedliteratemachinery_amd import ops  # noqa: E402
from advancedliteratemachinery_amd.model.transformer import Decoder  # noqa: E402


def main():
    I = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    split = len(sys.argv) > 2 and sys.argv[2] == 'split'
    M, nH, d, KB = 4096, 8, 512, 32
    dev = 'cuda'
    g = torch.Generator(device='cpu').manual_seed(0)
    if split:   # 32-key blocks of [hi plane | lo plane]
        K = torch.randn(2, I, nH, M // KB, 2, KB, 64, generator=g).to(dev, torch.bfloat16)
        Vt = torch.randn(2, I, nH, M // KB, 2, 64, KB, generator=g).to(dev, torch.bfloat16)
    else:
        K = torch.randn(2, I, nH, M, 64, generator=g).to(dev, torch.bfloat16)            # two layer slabs, alternated
        Vt = torch.randn(2, I, nH, M // KB, 64, KB, generator=g).to(dev, torch.bfloat16)
    qdt = torch.float32 if split else torch.bfloat16
    stride = nH * M * 64 * (2 if split else 1)

    def run(rows_per_img, n_launch, S):
        counts = [rows_per_img] * I
        groups, qt = Decoder.make_tiles(counts)
        gd = torch.tensor(groups, dtype=torch.int32, device=dev)
        R = sum(counts)
        q = torch.randn(R, d, device=dev).to(qdt)
        out = torch.empty(R, d, device=dev, dtype=qdt)
        partial = torch.empty(R, nH, S, 68, device=dev)
        for i in range(n_launch):
            ops.dec_cross_attn_step(q, K[i & 1], Vt[i & 1], stride, M, None, gd, len(groups), qt, partial, out, M, nH, S)
        torch.cuda.synchronize()
        print('rows/img %d: %d launches, q_tiles %d, S %d, groups %d' % (rows_per_img, n_launch, qt, S, len(groups)), flush=True)

    # key splits as Decoder._n_split picks them for this many images (transformer.py)
    s_pt = 8
    while s_pt > 1 and I * nH * s_pt > 1024:
        s_pt //= 2
    s_q4 = 1
    while s_q4 < 16 and I * nH * s_q4 < 512 and M >= 8 * KB * s_q4:
        s_q4 *= 2
    run(1, 22, s_pt)
    run(64, 10, s_q4)


if __name__ == '__main__':
    main()
