"""Streaming detokenizers - the linear-time forms of the reference's `mlx_vlm/tokenizer_utils.py:121-285` (SPM and
byte-level BPE) next to the naive one (`utils.py::NaiveStreamingDetokenizer`, reference 71-118), and the choice between
them from `tokenizer.json`'s decoder section (reference 413-480).

Both fast forms rest on one observation: a token that starts a new word (SentencePiece: leading U+2581; byte-level BPE:
first byte 0x20) closes everything before it, so text can be committed word by word and no token is decoded twice.  Here
every vocabulary entry is classified ONCE at construction into (starts_word, payload) tables - for BPE the payload is the
token's raw BYTES (the GPT-2 printable alphabet undone up front), so the streaming path appends bytes and decodes a word
when it closes; for SPM the payload is the piece with U+2581 already turned into a space, or a single byte for the
`<0xNN>` fallback tokens, which are collected and decoded together.
"""
from __future__ import annotations

import json
import os
from typing import List, Optional, Sequence

_WORD_MARK = "▁"


def _gpt2_alphabet() -> dict:
    """printable stand-in character -> byte value (the inverse of GPT-2's bytes_to_unicode)"""
    keep = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    table, extra = {}, 0
    for b in range(256):
        if b in keep:
            table[chr(b)] = b
        else:
            table[chr(256 + extra)] = b
            extra += 1
    return table


class StreamingDetokenizer:
    """Common surface (reference tokenizer_utils.py:19-68): `reset`, `add_token`, `finalize`, the committed `text`, the
    `tokens` seen, and `last_segment` = the text committed since the previous call."""

    def reset(self):
        self.offset = 0
        self.text = ""
        self.tokens: List[int] = []

    @property
    def last_segment(self) -> str:
        seg = self.text[self.offset:]
        self.offset = len(self.text)
        return seg

    def _commit(self, piece: str):
        # the very first word loses its leading space when trim_space is set (SentencePiece's dummy prefix)
        if not self.text and self.trim_space and piece[:1] == " ":
            piece = piece[1:]
        self.text += piece


class SPMStreamingDetokenizer(StreamingDetokenizer):
    """SentencePiece-style vocabularies (reference 121-197)."""

    def __init__(self, tokenizer, trim_space: bool = True):
        self.trim_space = trim_space
        vocab = tokenizer.vocab
        n = max(vocab.values()) + 1 if vocab else 0
        self._piece: List[Optional[str]] = [None] * n        # text of the piece, word mark -> space
        self._starts: List[bool] = [False] * n
        self._byte: List[int] = [-1] * n                     # <0xNN> fallback tokens
        for piece, idx in vocab.items():
            if len(piece) >= 6 and piece.startswith("<0x") and piece[5] == ">":
                try:
                    self._byte[idx] = int(piece[3:5], 16)
                    continue
                except ValueError:
                    pass
            self._starts[idx] = piece.startswith(_WORD_MARK)
            self._piece[idx] = piece.replace(_WORD_MARK, " ")
        self.reset()

    def reset(self):
        super().reset()
        self._word = ""                  # the open word
        self._raw = bytearray()          # pending byte-fallback tokens

    def _drain_bytes(self):
        if self._raw:
            self._word += self._raw.decode("utf-8", errors="replace")
            self._raw = bytearray()

    def add_token(self, token, skip_special_token_ids: Sequence[int] = ()):
        if token in skip_special_token_ids:
            return
        b = self._byte[token]
        if b >= 0:
            self._raw.append(b)
            return
        self._drain_bytes()
        piece = self._piece[token] or ""
        if self._starts[token]:
            self._commit(self._word)
            self._word = piece
        else:
            self._word += piece

    def finalize(self):
        self._drain_bytes()
        self._commit(self._word)
        self._word = ""


class BPEStreamingDetokenizer(StreamingDetokenizer):
    """OpenAI-style byte-level BPE vocabularies (reference 200-284)."""

    _alphabet = None

    def __init__(self, tokenizer, trim_space: bool = False):
        self.trim_space = trim_space
        if BPEStreamingDetokenizer._alphabet is None:
            BPEStreamingDetokenizer._alphabet = _gpt2_alphabet()
        alpha = BPEStreamingDetokenizer._alphabet
        vocab = tokenizer.vocab
        n = max(vocab.values()) + 1 if vocab else 0
        self._bytes: List[bytes] = [b""] * n
        for piece, idx in vocab.items():
            out = bytearray()
            for ch in piece:
                v = alpha.get(ch)
                if v is None:                     # not a byte-level piece (an added token spelled in plain text)
                    out += ch.encode("utf-8")
                else:
                    out.append(v)
            self._bytes[idx] = bytes(out)
        self.reset()

    def reset(self):
        super().reset()
        self._word = bytearray()

    def add_token(self, token, skip_special_token_ids: Sequence[int] = ()):
        if token in skip_special_token_ids:
            return
        raw = self._bytes[token]
        if raw[:1] == b" ":
            self._commit(self._word.decode("utf-8", errors="replace"))
            self._word = bytearray(raw)
        else:
            self._word += raw

    def finalize(self):
        self._commit(self._word.decode("utf-8", errors="ignore"))
        self._word = bytearray()


# ------------------------------------------------------------------ which one fits a tokenizer (reference 413-480)
_SPM_STEPS = [{"type": "Replace", "pattern": {"String": _WORD_MARK}, "content": " "}, {"type": "ByteFallback"}, {"type": "Fuse"}]
_SPM_STRIP = {"type": "Strip", "content": " ", "start": 1, "stop": 0}


def detokenizer_class_for(model_path: str):
    """-> a callable `cls(tokenizer)`: chosen from the `decoder` section of `model_path/tokenizer.json`; the naive
    detokenizer when the file is missing or describes anything else."""
    from functools import partial

    from .utils import NaiveStreamingDetokenizer

    path = os.path.join(str(model_path), "tokenizer.json")
    if not os.path.exists(path):
        return NaiveStreamingDetokenizer
    with open(path, "r", encoding="utf-8") as f:
        dec = json.load(f).get("decoder")          # a malformed file raises JSONDecodeError, as the reference does
    # the reference compares the WHOLE decoder description (tokenizer_utils.py:413-452: same keys, same values, nothing
    # extra anywhere in the structure - python's == on the parsed JSON is that comparison, bool / int type included) for the
    # two SPM forms, and only the type for ByteLevel; SPM forms are tried first, as there
    if dec is not None and _same_json(dec, {"type": "Sequence", "decoders": _SPM_STEPS + [_SPM_STRIP]}):
        return SPMStreamingDetokenizer
    if dec is not None and _same_json(dec, {"type": "Sequence", "decoders": _SPM_STEPS}):
        return partial(SPMStreamingDetokenizer, trim_space=False)
    if isinstance(dec, dict) and dec.get("type") == "ByteLevel":
        return BPEStreamingDetokenizer
    return NaiveStreamingDetokenizer


def _same_json(a, b) -> bool:
    """structural equality with equal TYPES at every node (1 != True, 1 != 1.0), no extra or missing keys / items"""
    if type(a) is not type(b):
        return False
    if isinstance(a, dict):
        return len(a) == len(b) and all(k in b and _same_json(v, b[k]) for k, v in a.items())
    if isinstance(a, list):
        return len(a) == len(b) and all(_same_json(x, y) for x, y in zip(a, b))
    return a == b
