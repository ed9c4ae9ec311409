"""Byte-level BPE tokenizer of the OpenCLIP text tower: what `open_clip.tokenize` does for
`FrozenOpenCLIPEmbedder.forward(text)` (`lvdm/modules/encoders/condition.py:205-209`).

`open_clip_torch==2.30.0` (requirements.txt:196) is a third-party dependency that is absent from this image together with its
vocabulary file (`bpe_simple_vocab_16e6.txt.gz`), so this is a restatement of the PUBLISHED algorithm (Radford et al., CLIP,
`simple_tokenizer.py`; Sennrich et al. BPE with GPT-2's byte-to-unicode table), not a copy, and its parity is UNPINNED: the tests
exercise it on a synthetic merge table only.  With the real merges file (`SimpleTokenizer(path)`; gzip or plain text, first line a
header, one merge "a b" per line, the first 48 894 used) the ids are the standard CLIP ids: 256 byte symbols, the same 256 with
`</w>`, one id per merge, `<start_of_text>` = 49406, `<end_of_text>` = 49407.

Differences from open_clip stated plainly: `ftfy.fix_text` (mojibake repair) is not applied (ftfy is not installed); html
unescaping, whitespace collapsing and lower-casing are."""
from __future__ import annotations

import gzip
import html
import re as _re

import torch

try:                                   # \\p{L} / \\p{N} classes need the `regex` package (installed here); plain `re` gets an ASCII fallback
    import regex as _rx
    _PAT = _rx.compile(r"<start_of_text>|<end_of_text>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+", _rx.IGNORECASE)
except ImportError:                    # pragma: no cover
    _PAT = _re.compile(r"<start_of_text>|<end_of_text>|'s|'t|'re|'ve|'m|'ll|'d|[A-Za-z]+|[0-9]|[^\sA-Za-z0-9]+", _re.IGNORECASE)

N_MERGES = 49152 - 256 - 2             # merges kept from the vocabulary file


def bytes_to_unicode():
    """GPT-2's reversible byte -> printable-unicode table: printable latin-1 bytes map to themselves, the rest to 256 + k"""
    keep = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    chars, extra = list(keep), 0
    for b in range(256):
        if b not in keep:
            keep.append(b)
            chars.append(256 + extra)
            extra += 1
    return dict(zip(keep, (chr(c) for c in chars)))


def _pairs(word):
    return {(a, b) for a, b in zip(word[:-1], word[1:])}


class SimpleTokenizer:
    def __init__(self, bpe_path=None, merges=None):
        """`bpe_path`: the CLIP merges file; or `merges`: a list of (left, right) symbol pairs in priority order (tests)"""
        if (bpe_path is None) == (merges is None):
            raise ValueError("give the merges file of the CLIP vocabulary (bpe_path) or an explicit merge list")
        if merges is None:
            opener = gzip.open if str(bpe_path).endswith(".gz") else open
            with opener(bpe_path, "rt", encoding="utf-8") as f:
                lines = f.read().split("\n")
            merges = [tuple(l.split()) for l in lines[1:N_MERGES + 1] if l.strip()]
        self.byte_encoder = bytes_to_unicode()
        vocab = list(self.byte_encoder.values())
        vocab = vocab + [v + "</w>" for v in vocab] + ["".join(m) for m in merges] + ["<start_of_text>", "<end_of_text>"]
        self.encoder = {s: i for i, s in enumerate(vocab)}
        self.decoder = {i: s for s, i in self.encoder.items()}
        self.ranks = {tuple(m): i for i, m in enumerate(merges)}
        self.cache = {"<start_of_text>": "<start_of_text>", "<end_of_text>": "<end_of_text>"}
        self.sot, self.eot = self.encoder["<start_of_text>"], self.encoder["<end_of_text>"]

    def bpe(self, token):
        """merge the lowest-ranked adjacent pair until none of the word's pairs is in the merge table"""
        if token in self.cache:
            return self.cache[token]
        word = tuple(token[:-1]) + (token[-1] + "</w>",)
        pairs = _pairs(word)
        while pairs:
            best = min(pairs, key=lambda p: self.ranks.get(p, float("inf")))
            if best not in self.ranks:
                break
            a, b = best
            out, i = [], 0
            while i < len(word):
                if i + 1 < len(word) and word[i] == a and word[i + 1] == b:
                    out.append(a + b)
                    i += 2
                else:
                    out.append(word[i])
                    i += 1
            word = tuple(out)
            pairs = _pairs(word) if len(word) > 1 else set()
        self.cache[token] = " ".join(word)
        return self.cache[token]

    def encode(self, text):
        text = _re.sub(r"\s+", " ", html.unescape(html.unescape(text)).strip()).lower()
        ids = []
        for tok in _PAT.findall(text):
            sym = "".join(self.byte_encoder[b] for b in tok.encode("utf-8"))
            ids.extend(self.encoder[t] for t in self.bpe(sym).split(" "))
        return ids

    def decode(self, ids):
        inv = {c: b for b, c in self.byte_encoder.items()}
        text = "".join(self.decoder[int(i)] for i in ids if int(i) not in (self.sot, self.eot))
        return bytearray(inv[c] for c in text).decode("utf-8", errors="replace").replace("</w>", " ")

    def __call__(self, texts, context_length=77):
        """`open_clip.tokenize`: [B, context_length] int64, `<start_of_text>` ids `<end_of_text>`, zero padded; longer texts are
        truncated and keep `<end_of_text>` as their last token"""
        if isinstance(texts, str):
            texts = [texts]
        out = torch.zeros(len(texts), context_length, dtype=torch.long)
        for i, t in enumerate(texts):
            ids = [self.sot] + self.encode(t) + [self.eot]
            if len(ids) > context_length:
                ids = ids[:context_length]
                ids[-1] = self.eot
            out[i, :len(ids)] = torch.tensor(ids, dtype=torch.long)
        return out
