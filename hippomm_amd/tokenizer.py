"""CLIP byte-pair-encoding tokenizer for the text tower: question strings -> (B,77) int64 token ids.

Host-side mirror of upstream ``imagebind.models.multimodal_preprocessors.SimpleTokenizer`` [upstream, recalled -- itself
OpenAI CLIP's ``simple_tokenizer.py``], which ``imagebind.data.load_and_transform_text`` applies at the reference's
``foundation_models.py:75-78``.  The merge table ``bpe_simple_vocab_16e6.txt.gz`` is data that ships with the upstream
package (``bpe/`` directory); it is not under /root/reference and cannot be downloaded here, so the path is an argument
and nothing is bundled.

Algorithm: lower-cased, whitespace-collapsed text is split by CLIP's regular expression; every piece is mapped byte by
byte to printable unicode stand-ins, the last symbol gets the ``</w>`` end-of-word marker, and adjacent symbol pairs are
merged greedily in merge-table rank order.  Vocabulary ids: 256 byte symbols, the same 256 with ``</w>``, one id per
merge (48 894 of them), ``<|startoftext|>`` = 49406, ``<|endoftext|>`` = 49407.  A row is ``[SOT] + ids + [EOT]`` cut
to 77 and zero-padded; the text tower picks the EOT position by ``argmax`` (EOT has the largest id).
"""
from __future__ import annotations

import gzip
import html
from functools import lru_cache
from pathlib import Path
from typing import Dict, Iterable, List, Optional, Tuple

import regex as re
import torch

CONTEXT_LENGTH = 77
N_MERGES = 49152 - 256 - 2                       # merges read from the table: lines[1 : 48895]
PATTERN = r"""<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+"""


@lru_cache()
def bytes_to_unicode() -> Dict[int, str]:
    """Reversible byte -> printable character table (printable latin-1 bytes map to themselves, the rest to 256+)."""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    cs = bs[:]
    n = 0
    for b in range(2 ** 8):
        if b not in bs:
            bs.append(b)
            cs.append(2 ** 8 + n)
            n += 1
    return dict(zip(bs, (chr(c) for c in cs)))


def read_merges(bpe_path: str, limit: Optional[int] = N_MERGES) -> List[Tuple[str, str]]:
    """Merge table file: a header line, then one ``left right`` pair per line (gzip or plain text)."""
    raw = Path(bpe_path).read_bytes()
    text = (gzip.decompress(raw) if raw[:2] == b"\x1f\x8b" else raw).decode("utf-8")
    lines = text.split("\n")[1:]
    if limit is not None:
        lines = lines[:limit]
    return [tuple(line.split()) for line in lines if line.strip()]


def basic_clean(text: str) -> str:
    try:                                         # upstream runs ftfy.fix_text first; identity on well-formed text
        import ftfy
        text = ftfy.fix_text(text)
    except ImportError:
        pass
    return html.unescape(html.unescape(text)).strip()


def whitespace_clean(text: str) -> str:
    return re.sub(r"\s+", " ", text).strip()


class SimpleTokenizer:
    def __init__(self, bpe_path: str, context_length: int = CONTEXT_LENGTH, merges: Optional[Iterable[Tuple[str, str]]] = None):
        self.byte_encoder = bytes_to_unicode()
        merges = list(merges) if merges is not None else read_merges(bpe_path)
        vocab = list(self.byte_encoder.values())
        vocab = vocab + [v + "</w>" for v in vocab]
        vocab.extend("".join(m) for m in merges)
        vocab.extend(["<|startoftext|>", "<|endoftext|>"])
        self.encoder = dict(zip(vocab, range(len(vocab))))
        self.decoder = {v: k for k, v in self.encoder.items()}
        self.byte_decoder = {v: k for k, v in self.byte_encoder.items()}
        self.bpe_ranks = dict(zip(merges, range(len(merges))))
        self.cache = {"<|startoftext|>": "<|startoftext|>", "<|endoftext|>": "<|endoftext|>"}
        self.pat = re.compile(PATTERN, re.IGNORECASE)
        self.context_length = context_length

    def bpe(self, token: str) -> str:
        if token in self.cache:
            return self.cache[token]
        word = tuple(token[:-1]) + (token[-1] + "</w>",)
        while len(word) > 1:
            pairs = set(zip(word[:-1], word[1:]))
            bigram = min(pairs, key=lambda p: self.bpe_ranks.get(p, float("inf")))
            if bigram not in self.bpe_ranks:
                break
            first, second = bigram
            merged, i = [], 0
            while i < len(word):
                if i < len(word) - 1 and word[i] == first and word[i + 1] == second:
                    merged.append(first + second)
                    i += 2
                else:
                    merged.append(word[i])
                    i += 1
            word = tuple(merged)
        out = " ".join(word)
        self.cache[token] = out
        return out

    def encode(self, text: str) -> List[int]:
        ids: List[int] = []
        text = whitespace_clean(basic_clean(text)).lower()
        for piece in re.findall(self.pat, text):
            piece = "".join(self.byte_encoder[b] for b in piece.encode("utf-8"))
            ids.extend(self.encoder[t] for t in self.bpe(piece).split(" "))
        return ids

    def decode(self, tokens: Iterable[int]) -> str:
        text = "".join(self.decoder[int(t)] for t in tokens)
        return bytearray(self.byte_decoder[c] for c in text).decode("utf-8", errors="replace").replace("</w>", " ")

    def __call__(self, texts, context_length: Optional[int] = None) -> torch.Tensor:
        if isinstance(texts, str):
            texts = [texts]
        context_length = context_length or self.context_length
        sot, eot = self.encoder["<|startoftext|>"], self.encoder["<|endoftext|>"]
        result = torch.zeros(len(texts), context_length, dtype=torch.long)
        for i, text in enumerate(texts):
            tokens = ([sot] + self.encode(text) + [eot])[:context_length]
            result[i, :len(tokens)] = torch.tensor(tokens)
        return result


def find_bpe_vocab(model_path: Optional[str]) -> Optional[Path]:
    """Where upstream keeps the merge table (``bpe/bpe_simple_vocab_16e6.txt.gz`` next to the code / checkpoints)."""
    import os
    names = ["bpe_simple_vocab_16e6.txt.gz", "bpe/bpe_simple_vocab_16e6.txt.gz"]
    roots = [Path(p) for p in (model_path, ".checkpoints", "bpe", ".") if p]
    if os.environ.get("IMAGEBIND_BPE"):
        return Path(os.environ["IMAGEBIND_BPE"])
    for root in roots:
        for name in names:
            if (root / name).is_file():
                return root / name
    return None
