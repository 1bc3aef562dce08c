"""CLIP byte-level BPE tokenizer: Python face of the C++ core `fc_bpe_*`
(fitclip_amd/csrc/bpe.cpp, C ABI in include/fitclip_hip.h).

`ClipVideoTextEncoder.get_tokenizer()` of the reference returns `clip.tokenize(texts, truncate=True)`
(`aligner/encoder/clip_video_text_encoder.py:64-65`); the algorithm is the one vendored in
`aligner/encoder/slip.py:75-164`.  Split of the work:
  * here: the text CLEANING of slip.py:63-72,138 - `html.unescape` twice, strip, `regex` white-space collapse,
    `str.lower()` - i.e. the very library calls the reference makes; `ftfy.fix_text` (slip.py:64) runs first when the
    package is importable (it is not in the offline image: one warning, and texts are then assumed to be well-formed
    Unicode - mojibake would tokenize differently from the reference; the golden ids only hold well-formed text);
  * C++: the CLIP pattern over UTF-8 (letter / number / space classes generated from the `regex` module,
    csrc/unicode_ranges.inc), byte-level merges by rank, vocabulary ids, SOT / EOT framing, truncation, padding.
The merge list is read from a LOCAL `bpe_simple_vocab_16e6.txt.gz`-style file (no network here).
"""
from __future__ import annotations

import ctypes as C
import html
from typing import Iterable, List, Mapping, Union

import regex
import torch

from . import _lib

SOT, EOT = "<|startoftext|>", "<|endoftext|>"


def _load_fix_text():
    try:
        import ftfy
        return ftfy.fix_text
    except ImportError:
        import warnings
        warnings.warn("ftfy is not installed: CLIP's text cleaning runs without ftfy.fix_text (slip.py:64); well-formed "
                      "Unicode tokenizes identically, mojibake does not", RuntimeWarning, stacklevel=3)
        return None


_fix_text = False  # resolved on first use: False = not looked up yet, None = ftfy unavailable


def clean_text(text: str) -> str:
    """`whitespace_clean(basic_clean(text)).lower()` (slip.py:63-72,138): ftfy.fix_text when available, `html.unescape`
    twice, strip, white-space collapse, lower case."""
    global _fix_text
    if _fix_text is False:
        _fix_text = _load_fix_text()
    if _fix_text is not None:
        text = _fix_text(text)
    return regex.sub(r"\s+", " ", html.unescape(html.unescape(text)).strip()).strip().lower()


class ClipBpeTokenizer:
    def __init__(self, bpe_path: str, context_length: int = 77) -> None:
        self._lib = _lib.load()
        handle = _lib._vp()
        _lib.check(self._lib.fc_bpe_create(str(bpe_path).encode(), context_length, C.byref(handle)), "fc_bpe_create")
        self._h = handle
        self.context_length = context_length

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                self._lib.fc_bpe_destroy(h)
            except Exception:  # interpreter shutdown
                pass

    @property
    def vocab_size(self) -> int:
        return self._lib.fc_bpe_vocab_size(self._h)

    @property
    def sot_token(self) -> int:
        return self._lib.fc_bpe_sot(self._h)

    @property
    def eot_token(self) -> int:
        return self._lib.fc_bpe_eot(self._h)

    def encode(self, text: str) -> List[int]:
        """`SimpleTokenizer.encode`: ids without SOT / EOT."""
        raw = clean_text(text).encode("utf-8")
        cap = max(16, len(raw) + 1)
        buf = (_lib._i64 * cap)()
        n = self._lib.fc_bpe_encode(self._h, raw, buf, cap)
        if n < 0:
            _lib.check(n, "fc_bpe_encode")
        return list(buf[:n])

    def decode(self, ids: Iterable[int]) -> str:
        ids = [int(i) for i in ids]
        arr = (_lib._i64 * max(1, len(ids)))(*ids)
        n = self._lib.fc_bpe_decode(self._h, arr, len(ids), None, 0)
        if n < 0:
            _lib.check(n, "fc_bpe_decode")
        out = C.create_string_buffer(n + 1)
        self._lib.fc_bpe_decode(self._h, arr, len(ids), out, n + 1)
        return out.raw[:n].decode("utf-8", errors="replace")

    def __call__(self, texts: Union[str, Iterable[str]]) -> Mapping[str, torch.Tensor]:
        """`clip.tokenize(texts, truncate=True)` -> {"input_ids": int64 [n, context_length]}."""
        texts = [texts] if isinstance(texts, str) else list(texts)
        out = torch.zeros((len(texts), self.context_length), dtype=torch.long)
        if texts:
            raw = [clean_text(t).encode("utf-8") for t in texts]
            arr = (C.c_char_p * len(raw))(*raw)
            _lib.check(self._lib.fc_bpe_tokenize(self._h, arr, len(raw), 1, out.data_ptr()), "fc_bpe_tokenize")
        return {"input_ids": out}
