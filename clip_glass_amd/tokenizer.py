"""CLIP byte-level BPE tokenizer (host side, init-time only).

Own implementation of the published algorithm the reference vendors in
clip/simple_tokenizer.py:62-132 + clip/clip.py:125-138 (lower-cased, whitespace-collapsed
text; byte -> printable-unicode alphabet; words end in '</w>'; greedy lowest-rank pair
merging; <|startoftext|> ... <|endoftext|>, zero padded to the context length).
The merge table is the reference's data asset `assets/bpe_simple_vocab_16e6.txt.gz`
(path given by the caller; default = the reference's cwd-relative default,
simple_tokenizer.py:11-13).  ftfy is not available here: `fix_text` is the identity for the
plain-ASCII prompts the CLI takes.
"""
import gzip
import html
import os

import numpy as np

try:
    import regex as _re
    _WORD = _re.compile(r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+",
                        _re.IGNORECASE)
except ImportError:  # pragma: no cover
    _re = None

DEFAULT_BPE = "./assets/bpe_simple_vocab_16e6.txt.gz"


def _byte_alphabet():
    """256 bytes -> 256 printable unicode characters (printable latin-1 first, the rest shifted past 255)."""
    keep = [b for b in range(256) if 33 <= b <= 126 or 161 <= b <= 172 or 174 <= b <= 255]
    table, extra = {}, 0
    for b in keep:
        table[b] = chr(b)
    for b in range(256):
        if b not in table:
            table[b] = chr(256 + extra)
            extra += 1
    return table, keep + [b for b in range(256) if b not in set(keep)]


class ClipTokenizer:
    def __init__(self, bpe_path=DEFAULT_BPE, context_length=77):
        if _re is None:
            raise RuntimeError("the `regex` package is required for the CLIP tokenizer")
        if not os.path.exists(bpe_path):
            raise FileNotFoundError("CLIP BPE merges not found: %s (the reference ships it under assets/)" % bpe_path)
        self.context_length = context_length
        self.byte_to_char, order = _byte_alphabet()
        lines = gzip.open(bpe_path).read().decode("utf-8").split("\n")
        merges = [tuple(l.split()) for l in lines[1:49152 - 256 - 2 + 1] if l.strip()]
        base = [self.byte_to_char[b] for b in order]
        vocab = base + [c + "</w>" for c in base] + ["".join(m) for m in merges] + ["<|startoftext|>", "<|endoftext|>"]
        self.token_id = {tok: i for i, tok in enumerate(vocab)}
        self.rank = {m: i for i, m in enumerate(merges)}
        self.sot, self.eot = self.token_id["<|startoftext|>"], self.token_id["<|endoftext|>"]
        self._memo = {}

    def _merge_word(self, word):
        """word: string over the byte alphabet -> list of BPE symbols."""
        if word in self._memo:
            return self._memo[word]
        syms = list(word[:-1]) + [word[-1] + "</w>"]
        while len(syms) > 1:
            best, best_rank = None, None
            for i in range(len(syms) - 1):
                r = self.rank.get((syms[i], syms[i + 1]))
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = (syms[i], syms[i + 1]), r
            if best is None:
                break
            out, i = [], 0
            while i < len(syms):
                if i + 1 < len(syms) and syms[i] == best[0] and syms[i + 1] == best[1]:
                    out.append(best[0] + best[1])
                    i += 2
                else:
                    out.append(syms[i])
                    i += 1
            syms = out
        self._memo[word] = syms
        return syms

    def encode(self, text):
        text = html.unescape(html.unescape(text)).strip()
        text = " ".join(text.split()).lower()
        ids = []
        for w in _WORD.findall(text):
            w = "".join(self.byte_to_char[b] for b in w.encode("utf-8"))
            ids.extend(self.token_id[s] for s in self._merge_word(w))
        return ids

    def tokenize(self, texts):
        """clip.tokenize (clip/clip.py:125-138): int64 [n, context_length]; raises RuntimeError when too long."""
        if isinstance(texts, str):
            texts = [texts]
        out = np.zeros((len(texts), self.context_length), dtype=np.int64)
        for i, t in enumerate(texts):
            ids = [self.sot] + self.encode(t) + [self.eot]
            if len(ids) > self.context_length:
                raise RuntimeError("Input %s is too long for context length %d" % (t, self.context_length))
            out[i, :len(ids)] = ids
        return out
