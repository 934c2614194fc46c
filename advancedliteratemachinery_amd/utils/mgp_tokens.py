"""BPE / WordPiece string decoding for MGP-STR's sub-word heads, offline.

The reference decodes its BPE and WordPiece predictions with Hugging Face tokenizers fetched from the hub
(OCR/MGP-STR/utils.py:23-24: GPT2Tokenizer.from_pretrained("gpt2"), BertTokenizer.from_pretrained("bert-base-uncased");
bpe_decode / wp_decode :68-87; string pruning and fusion test_final.py:196-236).  Decoding needs only the id -> token
tables, so this module reads them from LOCAL files (GPT-2 `vocab.json`, BERT `vocab.txt`) and restates the two decoders:

  * GPT-2 byte-level BPE: tokens are concatenated and every character is mapped back through the byte <-> unicode table of
    the GPT-2 encoder, the bytes decoded as UTF-8 (errors='replace');
  * BERT WordPiece: ' '.join(tokens) with ' ##' removed, then -- as the reference does -- all whitespace dropped;
  * both followed by the tokenizers' `clean_up_tokenization` (' .' -> '.', " n't" -> "n't", ...), the default of the
    transformers 4.x releases the reference was written against.

tests/test_mgp_tokens.py checks both against the installed transformers' GPT2Tokenizer / BertTokenizer built from the same
synthetic files, and the pruning / fusion against a restatement of test_final.py."""
import json


def bytes_to_unicode():
    """GPT-2's reversible byte -> printable-unicode table (encoder.py of the GPT-2 release; transformers' tokenization_gpt2)."""
    bs = list(range(ord('!'), ord('~') + 1)) + list(range(ord('\xa1'), ord('\xac') + 1)) + list(range(ord('\xae'), ord('\xff') + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return dict(zip(bs, [chr(c) for c in cs]))


def clean_up_tokenization(s):
    """transformers' PreTrainedTokenizerBase.clean_up_tokenization"""
    for a, b in ((' .', '.'), (' ?', '?'), (' !', '!'), (' ,', ','), (" ' ", "'"), (" n't", "n't"), (" 'm", "'m"), (" 's", "'s"),
                 (" 've", "'ve"), (" 're", "'re")):
        s = s.replace(a, b)
    return s


class BpeVocab(object):
    """GPT-2 vocabulary (vocab.json: token -> id).  EOS of MGP-STR's BPE head is id 2 ('#' in the GPT-2 vocabulary)."""

    def __init__(self, vocab_json):
        with open(vocab_json, encoding='utf-8') as f:
            enc = json.load(f)
        self.tokens = {int(i): t for t, i in enc.items()}
        self.byte_decoder = {c: b for b, c in bytes_to_unicode().items()}

    def decode(self, ids, clean_up=True):
        """clean_up: the transformers 4.x default the reference ran with (5.x no longer applies it to byte-level BPE)"""
        text = ''.join(self.tokens.get(int(i), '') for i in ids)
        raw = bytearray(self.byte_decoder[c] for c in text if c in self.byte_decoder).decode('utf-8', errors='replace')
        return clean_up_tokenization(raw) if clean_up else raw


class WordPieceVocab(object):
    """BERT vocabulary (vocab.txt: one token per line, id = line number).  EOS of the WordPiece head is id 102 ('[SEP]')."""

    def __init__(self, vocab_txt):
        with open(vocab_txt, encoding='utf-8') as f:
            self.tokens = [line.rstrip('\n') for line in f]

    def decode(self, ids):
        toks = [self.tokens[int(i)] if 0 <= int(i) < len(self.tokens) else '[UNK]' for i in ids]
        return clean_up_tokenization(' '.join(toks).replace(' ##', '').strip())


def decode_strings(results, bpe=None, wp=None):
    """Adds the sub-word strings and the fused prediction to MGPSTR.recognize() records, exactly as test_final.py:196-236:
    bpe_text = decode(bpe ids) pruned at the first '#', wp_text = decode(wp ids) with whitespace removed, pruned at the first
    '[SEP]'; `text` = the string of the granularity with the highest confidence (`choice`; '' if none).  A granularity
    without a vocabulary keeps its ids only and cannot be chosen as `text` (then `text` is None when it wins)."""
    for r in results:
        if bpe is not None:
            s = bpe.decode(r['bpe_ids'])
            r['bpe_text'] = s[:s.find('#')]           # find() == -1 drops the last character, as in the reference
        if wp is not None:
            s = ''.join(wp.decode(r['wp_ids']).split())
            r['wp_text'] = s[:s.find('[SEP]')]
        cand = (r['char_text'], r.get('bpe_text'), r.get('wp_text'))
        r['text'] = cand[r['choice']] if r['choice'] >= 0 else ''
    return results
