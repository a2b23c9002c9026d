"""CLIP byte-pair-encoding tokenizer reading LOCAL vocabulary files (nothing is downloaded).

The reference tokenises through two third-party packages that are absent here:
  * ``open_clip.tokenize`` (/root/reference/models/modeling/meta_arch/clip.py:147-149; open-clip-torch 2.0.2 ``SimpleTokenizer``,
    vocabulary ``bpe_simple_vocab_16e6.txt.gz`` inside that package) for the label / caption embeddings: pads with 0;
  * HuggingFace ``CLIPTokenizer`` inside ldm's ``FrozenCLIPEmbedder`` (stable-diffusion-sdkit; ldm.py:105 ``embed_text([""])``) for the
    Stable-Diffusion text encoder: ``vocab.json`` + ``merges.txt`` of the local ``openai/clip-vit-large-patch14`` directory the
    reference's README has the user unpack (README.md:28-35); pads with the end-of-text id.
Both are the SAME published algorithm and vocabulary (OpenAI CLIP, 49408 entries, <|startoftext|> = 49406, <|endoftext|> = 49407),
restated here: lower-case, split with CLIP's pattern, map the UTF-8 bytes of a word to printable code points, merge the pair with the
lowest rank until none is left, the last symbol of a word carrying ``</w>``.  Text clean-up: html.unescape twice + whitespace
collapse as in both packages; ``ftfy.fix_text`` (mojibake repair, absent here) is skipped - PARITY UNPINNED for text that ftfy
would change (none of the ScanNet label names / generated captions contain such characters).
"""
from __future__ import annotations

import gzip
import html
import json
import os
from functools import lru_cache

import torch

SOT_TOKEN, EOT_TOKEN = "<|startoftext|>", "<|endoftext|>"


@lru_cache()
def bytes_to_unicode():
    """byte -> printable unicode character, the reversible table of the GPT-2 / CLIP tokenizers"""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("\xa1"), ord("\xac") + 1)) + list(range(ord("\xae"), ord("\xff") + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return dict(zip(bs, [chr(c) for c in cs]))


def _pairs(word):
    return set(zip(word[:-1], word[1:]))


class ClipBPE:
    """encode(text) -> token ids; __call__(texts) -> (n, context_length) int64 tensor [SOT, ids..., EOT, pad...]"""

    def __init__(self, encoder: dict, merges: list):
        import regex

        self.encoder = dict(encoder)
        self.decoder = {v: k for k, v in self.encoder.items()}
        self.bpe_ranks = {tuple(m): i for i, m in enumerate(merges)}
        self.byte_encoder = bytes_to_unicode()
        self.sot, self.eot = self.encoder[SOT_TOKEN], self.encoder[EOT_TOKEN]
        self.cache = {SOT_TOKEN: SOT_TOKEN, EOT_TOKEN: EOT_TOKEN}
        self.pat = regex.compile(r"""<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+""",
                                 regex.IGNORECASE)

    # ---- construction from local files
    @classmethod
    def from_files(cls, vocab_json, merges_txt):
        """HuggingFace layout: vocab.json (token -> id) + merges.txt (one 'a b' pair per line after a '#version' header)"""
        with open(vocab_json, encoding="utf-8") as f:
            encoder = json.load(f)
        with open(merges_txt, encoding="utf-8") as f:
            lines = f.read().split("\n")
        merges = [tuple(l.split()) for l in lines if l and not l.startswith("#version") and len(l.split()) == 2]
        return cls(encoder, merges)

    @classmethod
    def from_openclip_vocab(cls, path):
        """open_clip / OpenAI layout: bpe_simple_vocab_16e6.txt(.gz); the vocabulary is rebuilt from the merges in their order"""
        opener = gzip.open if path.endswith(".gz") else open
        with opener(path, "rt", encoding="utf-8") as f:
            lines = f.read().split("\n")
        merges = [tuple(l.split()) for l in lines[1:49152 - 256 - 2 + 1]]
        vocab = list(bytes_to_unicode().values())
        vocab = vocab + [v + "</w>" for v in vocab] + ["".join(m) for m in merges] + [SOT_TOKEN, EOT_TOKEN]
        return cls({t: i for i, t in enumerate(vocab)}, merges)

    @classmethod
    def from_dir(cls, d):
        """a directory holding either layout (searched one level deep, e.g. openai/clip-vit-large-patch14/); None if neither"""
        cands = [d] + [os.path.join(d, s) for s in sorted(os.listdir(d)) if os.path.isdir(os.path.join(d, s))] if os.path.isdir(d) else []
        for c in cands:
            v, m = os.path.join(c, "vocab.json"), os.path.join(c, "merges.txt")
            if os.path.isfile(v) and os.path.isfile(m):
                return cls.from_files(v, m)
            for name in ("bpe_simple_vocab_16e6.txt.gz", "bpe_simple_vocab_16e6.txt"):
                if os.path.isfile(os.path.join(c, name)):
                    return cls.from_openclip_vocab(os.path.join(c, name))
        return None

    # ---- the algorithm
    def bpe(self, token):
        if token in self.cache:
            return self.cache[token]
        word = tuple(token[:-1]) + (token[-1] + "</w>",)
        pairs = _pairs(word)
        if not pairs:
            return token + "</w>"
        while True:
            bigram = min(pairs, key=lambda p: self.bpe_ranks.get(p, float("inf")))
            if bigram not in self.bpe_ranks:
                break
            first, second = bigram
            new, i = [], 0
            while i < len(word):
                try:
                    j = word.index(first, i)
                except ValueError:
                    new.extend(word[i:])
                    break
                new.extend(word[i:j])
                i = j
                if word[i] == first and i < len(word) - 1 and word[i + 1] == second:
                    new.append(first + second)
                    i += 2
                else:
                    new.append(word[i])
                    i += 1
            word = tuple(new)
            if len(word) == 1:
                break
            pairs = _pairs(word)
        out = " ".join(word)
        self.cache[token] = out
        return out

    @staticmethod
    def clean(text):
        text = html.unescape(html.unescape(text)).strip()
        return " ".join(text.split()).strip().lower()

    def encode(self, text):
        ids = []
        for token in self.pat.findall(self.clean(text)):
            token = "".join(self.byte_encoder[b] for b in token.encode("utf-8"))
            ids.extend(self.encoder[t] for t in self.bpe(token).split(" "))
        return ids

    def decode(self, ids):
        inv = {v: k for k, v in self.byte_encoder.items()}
        text = "".join(self.decoder[int(i)] for i in ids)
        return bytearray(inv[c] for c in text).decode("utf-8", errors="replace").replace("</w>", " ")

    def __call__(self, texts, context_length=77, pad_id=0):
        """open_clip.tokenize semantics (2.0.2): [SOT] + ids + [EOT], truncated to context_length with EOT kept last, padded with
        pad_id (0 for open_clip; pass self.eot for the HuggingFace tokenizer of the Stable-Diffusion text encoder)"""
        if isinstance(texts, str):
            texts = [texts]
        out = torch.full((len(texts), context_length), int(pad_id), dtype=torch.long)
        for i, t in enumerate(texts):
            ids = [self.sot] + self.encode(t) + [self.eot]
            if len(ids) > context_length:
                ids = ids[:context_length]
                ids[-1] = self.eot
            out[i, : len(ids)] = torch.tensor(ids)
        return out
