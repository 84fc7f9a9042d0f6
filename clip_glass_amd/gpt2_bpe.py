"""GPT-2 byte-level BPE (host side) — own implementation of the published algorithm the reference vendors
in gpt2/encoder.py:41-115: regex pre-tokenisation (leading-space words), byte -> printable-unicode
alphabet, greedy lowest-rank pair merging, ids from `encoder.json`, merges from `vocab.bpe`
(data assets given by path: config.encoder / config.vocab, config.py:15-16)."""
import json

try:
    import regex as _re
    _PAT = _re.compile(r"'s|'t|'re|'ve|'m|'ll|'d| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+")
except ImportError:  # pragma: no cover
    _re = None

from .tokenizer import _byte_alphabet


class Gpt2Bpe:
    def __init__(self, encoder_json, vocab_bpe):
        if _re is None:
            raise RuntimeError("the `regex` package is required for the GPT-2 tokenizer")
        with open(encoder_json, "r") as f:
            self.token_id = json.load(f)
        with open(vocab_bpe, "r", encoding="utf-8") as f:
            merges = [tuple(l.split()) for l in f.read().split("\n")[1:-1]]
        self.id_token = {v: k for k, v in self.token_id.items()}
        self.rank = {m: i for i, m in enumerate(merges)}
        self.byte_to_char, _ = _byte_alphabet()
        self.char_to_byte = {c: b for b, c in self.byte_to_char.items()}
        self.eot = self.token_id["<|endoftext|>"]
        self._memo = {}

    def _merge(self, word):
        if word in self._memo:
            return self._memo[word]
        syms = list(word)
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
                    out.append(best[0] + best[1]); i += 2
                else:
                    out.append(syms[i]); i += 1
            syms = out
        self._memo[word] = syms
        return syms

    def encode(self, text):
        ids = []
        for w in _PAT.findall(text):
            w = "".join(self.byte_to_char[b] for b in w.encode("utf-8"))
            ids.extend(self.token_id[s] for s in self._merge(w))
        return ids

    def decode(self, tokens):
        text = "".join(self.id_token[int(t)] for t in tokens)
        return bytearray(self.char_to_byte[c] for c in text).decode("utf-8", errors="replace")
