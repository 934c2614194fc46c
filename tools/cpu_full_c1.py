"""BASELINE config 1 run IN FULL on host cores (one complete, not extrapolated, CPU run): the oracle
(= the reference's algorithm: no KV cache, every prefix re-decoded, memory broadcast per instance) on ONE 640x640
image, Swin-T widths (the patched-width extension of tests/golden/swint_nofpn.pt: embed 96, depths 2-2-6-2, no FPN),
seeded weights, pt_seq_length 32 -> 16 text instances, each with its 32-token polygon and 25-token transcription.
    python tools/cpu_full_c1.py [threads] > profiles/rNN_cpu_full_c1.txt"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from oracle import gen_golden as G  # noqa: E402
from oracle import omniparser_ref as O  # noqa: E402


def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else (os.cpu_count() or 1)
    torch.set_num_threads(threads)
    case = dict(args=dict(tfm_pre_norm=True, use_fpn=False, use_char_window_prompt=True, pt_seq_length=32),
                hw=(640, 640), depths=(2, 2, 6, 2), swin=dict(embed_dim=96, num_heads=(3, 6, 12, 24)))
    args, sd, img, mask, seqs = G.case_inputs(case)
    with torch.no_grad():
        t0 = time.time()
        enc = O.encode(sd, args, img, mask, case['depths'], case['swin']['num_heads'])
        t1 = time.time()
        pt_seq, pt_probs = O.decode_pt_seq(sd, args, seqs[0], enc['memory'], enc['mask'], enc['pos'], None)
        t2 = time.time()
        res = O.spot(sd, args, pt_seq, seqs[1], seqs[2], enc['memory'], enc['mask'], enc['pos']) if pt_seq.numel() else None
        t3 = time.time()
    n = 0 if res is None else res[0][0].numel() // 2
    print('config 1 (Swin-T widths, 640x640, %d instances) on %d host threads, oracle fp32: encode %.2f s, point decoder %.2f s '
          '(%d tokens), polygon + recognition %.2f s; total %.2f s -> %.5f images/s, %.2f chars/s'
          % (n, threads, t1 - t0, t2 - t1, pt_seq.numel(), t3 - t2, t3 - t0, 1.0 / (t3 - t0), n * args.rec_length / (t3 - t0)))


if __name__ == '__main__':
    main()
