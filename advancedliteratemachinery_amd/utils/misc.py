"""Token ids -> points / polygons / strings (the step right after the hot path).

Mirror of decode_seq in the reference (OCR/OmniParser/utils/misc.py:147-189); pure host code.
"""


def decode_seq(seq, args, decode_type='pt', probs=None):
    nb = args.num_bins
    if decode_type == 'pt':
        return [{'point': ((row[0] / nb).item(), (row[1] / nb).item())} for row in seq.reshape(-1, 2)]
    if decode_type == 'poly':
        return [{'polygon': row / nb} for row in seq.reshape(-1, 32)]
    if decode_type != 'rec':
        raise ValueError(decode_type)
    seq = seq.reshape(-1, args.rec_length)
    probs = probs.reshape(-1, args.rec_length)
    results, scores = [], []
    for ids, pr in zip(seq.tolist(), probs.tolist()):
        chars, kept = [], []
        for t, p in zip(ids, pr):
            if t == args.recog_pad_index or t == args.rec_eos_index:
                break
            if t == args.recog_pad_index - 1:  # 'unknown' character: skipped, not terminating
                continue
            chars.append(args.chars[t - nb])
            kept.append(p)
        scores.append(sum(kept) / (len(kept) + 1e-5))
        results.append({'rec': ''.join(chars)})
    return results, scores
