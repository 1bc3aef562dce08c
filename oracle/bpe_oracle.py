"""Pure-Python restatement of the CLIP byte-level BPE tokenizer.  TEST INFRASTRUCTURE ONLY (the product tokenizer is
the C++ core behind `fc_bpe_*`, fitclip_amd/csrc/bpe.cpp, wrapped by fitclip_amd/bpe.py).

`ClipVideoTextEncoder.get_tokenizer()` of the reference returns `clip.tokenize(texts, truncate=True)`
(`aligner/encoder/clip_video_text_encoder.py:64-65`); the algorithm is the one vendored in
`aligner/encoder/slip.py:75-164`: clean + lower-case the text, split it with the CLIP regular expression, map the
UTF-8 bytes of each piece to printable code points, greedily apply the ranked merge list, look the pieces up in the
vocabulary {bytes, bytes + "</w>", merges, <|startoftext|>, <|endoftext|>}, then frame as SOT ... EOT, zero-pad to the
context length and, when too long, cut and put EOT back in the last slot (the `truncate=True` behaviour).

The merge list is read from a LOCAL `bpe_simple_vocab_16e6.txt.gz`-style file: the real file is not available offline,
so the tests pin this implementation against the reference's own `SimpleTokenizer` class on a synthetic merges file
(`tests/golden/bpe_toy.*`).  `ftfy` is not installed: texts are assumed to be well-formed Unicode.
"""
from __future__ import annotations

import gzip
import html
from functools import lru_cache
from typing import Dict, Iterable, List, Mapping, Sequence, Tuple, Union

import regex
import torch

SOT, EOT = "<|startoftext|>", "<|endoftext|>"
_PIECES = regex.compile(r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+",
                        regex.IGNORECASE)
_FULL_MERGES = 49152 - 256 - 2  # merges kept from the published vocabulary file


@lru_cache()
def byte_alphabet() -> Tuple[str, ...]:
    """byte value -> one printable character.  Printable Latin-1 bytes stand for themselves; the 68 remaining ones
    (controls, space, DEL, 0x80-0xA0, soft hyphen) are moved to U+0100 onwards, in byte order."""
    keep = set(range(0x21, 0x7F)) | set(range(0xA1, 0xAD)) | set(range(0xAE, 0x100))
    table, spill = [], 0
    for b in range(256):
        if b in keep:
            table.append(chr(b))
        else:
            table.append(chr(256 + spill))
            spill += 1
    return tuple(table)


class ClipBpeTokenizer:
    def __init__(self, bpe_path: str, context_length: int = 77) -> None:
        with gzip.open(bpe_path, "rt", encoding="utf-8") as f:
            lines = f.read().split("\n")
        merges = [tuple(l.split()) for l in lines[1:_FULL_MERGES + 1]]  # line 0 is the version header
        self.merge_rank: Dict[Tuple[str, ...], int] = {m: i for i, m in enumerate(merges)}
        alphabet = sorted(byte_alphabet(), key=_reference_order)
        vocab = alphabet + [c + "</w>" for c in alphabet] + ["".join(m) for m in merges] + [SOT, EOT]
        self.token_id: Dict[str, int] = {tok: i for i, tok in enumerate(vocab)}
        self.id_token: Dict[int, str] = {i: tok for tok, i in self.token_id.items()}
        self.context_length = context_length
        self._byte_of = {c: b for b, c in enumerate(byte_alphabet())}
        self._cache: Dict[str, List[str]] = {}

    @property
    def vocab_size(self) -> int:
        return len(self.token_id)

    # ------------------------------------------------------------------------------------------------------- BPE
    def _merge_word(self, word: str) -> List[str]:
        if word in (SOT, EOT):
            return [word]
        hit = self._cache.get(word)
        if hit is not None:
            return hit
        symbols = list(word[:-1]) + [word[-1] + "</w>"]
        while len(symbols) > 1:
            ranked = [(self.merge_rank.get((a, b), None), k) for k, (a, b) in enumerate(zip(symbols, symbols[1:]))]
            candidates = [(r, k) for r, k in ranked if r is not None]
            if not candidates:
                break
            best_rank, at = min(candidates)
            first, second = symbols[at], symbols[at + 1]
            merged, k = [], 0
            while k < len(symbols):
                if k + 1 < len(symbols) and symbols[k] == first and symbols[k + 1] == second:
                    merged.append(first + second)
                    k += 2
                else:
                    merged.append(symbols[k])
                    k += 1
            symbols = merged
        self._cache[word] = symbols
        return symbols

    def encode(self, text: str) -> List[int]:
        text = regex.sub(r"\s+", " ", html.unescape(html.unescape(text)).strip()).strip().lower()  # slip.py:63-72,138
        alphabet = byte_alphabet()
        ids: List[int] = []
        for piece in _PIECES.findall(text):
            word = "".join(alphabet[b] for b in piece.encode("utf-8"))
            ids.extend(self.token_id[s] for s in self._merge_word(word))
        return ids

    def decode(self, ids: Iterable[int]) -> str:
        text = "".join(self.id_token[int(i)] for i in ids)  # "</w>" is spelled with alphabet characters
        return bytearray(self._byte_of[c] for c in text).decode("utf-8", errors="replace").replace("</w>", " ")

    def __call__(self, texts: Union[str, Iterable[str]]) -> Mapping[str, torch.Tensor]:
        texts = [texts] if isinstance(texts, str) else list(texts)
        sot, eot = self.token_id[SOT], self.token_id[EOT]
        out = torch.zeros((len(texts), self.context_length), dtype=torch.long)
        for i, text in enumerate(texts):
            toks = [sot] + self.encode(text) + [eot]
            if len(toks) > self.context_length:  # clip.tokenize(truncate=True): cut and keep EOT last
                toks = toks[:self.context_length]
                toks[-1] = eot
            out[i, :len(toks)] = torch.tensor(toks)
        return {"input_ids": out}


def _reference_order(ch: str) -> int:
    """Position of a byte character in the published vocabulary: the printable bytes first (in byte order), then the
    relocated ones (in byte order) - i.e. the order of the published `bytes_to_unicode()` table."""
    code = ord(ch)
    printable = list(range(0x21, 0x7F)) + list(range(0xA1, 0xAD)) + list(range(0xAE, 0x100))
    return printable.index(code) if code < 256 else len(printable) + (code - 256)
