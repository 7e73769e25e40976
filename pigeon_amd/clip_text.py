"""Scoring stage, text side: CLIP ViT-H-14 TEXT tower + its byte-pair tokenizer on the MI355X.

What Evaluation/eval_utils.py:115,126 does through open_clip for every target text —
`open_clip.get_tokenizer('ViT-H-14')(texts)` and `model.encode_text(tokens)` of the laion2b-s32b-b79K checkpoint —
so that `CLIPScorer.score(images, texts=[...])` needs nothing from outside this package.

open_clip is a third-party wheel absent from /root/reference and from this image: PARITY UNPINNED for the tower (same
status as the image tower, clip_vision.py).  The module restates open_clip's `CLIP` text half with its state-dict
names (`token_embedding.weight`, `positional_embedding`, `transformer.resblocks.*`, `ln_final.*`, `text_projection`,
all at the top level of `open_clip_pytorch_model.bin`): token + positional embedding, 24 pre-norm residual blocks
whose 16 x 64 heads attend CAUSALLY (pgn_attention_causal_fwd), GELU MLPs, `ln_final`, the row at the end-of-text token,
`text_projection`.  LayerNorms (+ residual adds) run in pgn_add_layernorm, GEMMs on hipBLASLt; CPU tensors raise.

The tokenizer restates the published CLIP byte-level BPE (lower-cased, whitespace-collapsed text; the pattern
`'s|'t|'re|'ve|'m|'ll|'d|letters+|digit|other+`; bytes mapped to printable code points; merges ranked by their line in
`bpe_simple_vocab_16e6.txt.gz`; ids = 256 byte symbols, the same 256 with `</w>`, 48894 merges, start / end of text).
The merges file ships with open_clip (and as `merges.txt` with every Hugging Face CLIP checkpoint), not with this
package or the reference: pass its path.  Cross-checked against transformers' independent CLIPTokenizer on a
synthetic merges table (tests/test_clip_text.py).  open_clip additionally runs `ftfy.fix_text` when ftfy is
installed (mojibake repair); without ftfy, as here, both implementations skip it.
"""
import gzip
import html
from functools import lru_cache
from typing import Dict, List, Sequence, Tuple, Union

import torch
import torch.nn as nn

from .clip_vision import _Transformer, _layer_norm

TEXT_VIT_H_14 = dict(context_length=77, vocab_size=49408, width=1024, heads=16, layers=24, mlp_ratio=4.0,
                     output_dim=1024)
N_MERGES = 49152 - 256 - 2          # lines of the merges file that are used (after its one header line)


@lru_cache()
def _byte_symbols() -> Dict[int, str]:
    """byte value -> the printable code point that stands for it: printable latin-1 bytes map to themselves, the other
    68 to U+0100 upwards in byte order."""
    keep = list(range(ord("!"), ord("~") + 1)) + list(range(0xA1, 0xAC + 1)) + list(range(0xAE, 0xFF + 1))
    table, spare = {}, 0
    for b in range(256):
        if b in keep:
            table[b] = chr(b)
        else:
            table[b] = chr(256 + spare)
            spare += 1
    return table


class SimpleTokenizer:
    """`open_clip.get_tokenizer('ViT-H-14')`: texts -> int64 [len(texts), context_length], zero-padded,
    <start_of_text> first, <end_of_text> last (kept when a text is truncated)."""

    def __init__(self, merges_path: str, context_length: int = 77, n_merges: int = N_MERGES):
        import regex
        opener = gzip.open if str(merges_path).endswith(".gz") else open
        with opener(merges_path, "rt", encoding="utf-8") as f:
            lines = f.read().split("\n")
        pairs = [tuple(ln.split()) for ln in lines[1:1 + n_merges] if ln.strip()]
        if any(len(p) != 2 for p in pairs):
            raise ValueError(f"{merges_path}: every merges line after the header must hold two symbols")
        base = [_byte_symbols()[b] for b in sorted(_byte_symbols(), key=lambda b: _order_key(b))]
        vocab = base + [s + "</w>" for s in base] + ["".join(p) for p in pairs] + ["<start_of_text>", "<end_of_text>"]
        self.encoder = {tok: i for i, tok in enumerate(vocab)}
        self.decoder = {i: tok for tok, i in self.encoder.items()}
        self.rank = {p: i for i, p in enumerate(pairs)}
        self.sot, self.eot = self.encoder["<start_of_text>"], self.encoder["<end_of_text>"]
        self.context_length = context_length
        self._split = regex.compile(
            r"<start_of_text>|<end_of_text>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+", regex.IGNORECASE)
        self._word_cache: Dict[str, Tuple[str, ...]] = {"<start_of_text>": ("<start_of_text>",),
                                                        "<end_of_text>": ("<end_of_text>",)}

    @property
    def vocab_size(self) -> int:
        return len(self.encoder)

    def _merge_word(self, word: str) -> Tuple[str, ...]:
        """One pre-token (already in byte symbols) -> its BPE symbols: repeatedly fuse the adjacent pair with the lowest
        merge rank, every occurrence left to right, until no adjacent pair is in the table."""
        hit = self._word_cache.get(word)
        if hit is not None:
            return hit
        syms = list(word[:-1]) + [word[-1] + "</w>"]
        while len(syms) > 1:
            best = min(zip(syms, syms[1:]), key=lambda p: self.rank.get(p, float("inf")))
            if best not in self.rank:
                break
            fused, i = [], 0
            while i < len(syms):
                if i + 1 < len(syms) and (syms[i], syms[i + 1]) == best:
                    fused.append(syms[i] + syms[i + 1])
                    i += 2
                else:
                    fused.append(syms[i])
                    i += 1
            syms = fused
        out = tuple(syms)
        self._word_cache[word] = out
        return out

    @staticmethod
    def clean(text: str) -> str:
        text = html.unescape(html.unescape(text)).strip()
        return " ".join(text.split()).strip().lower()

    def encode(self, text: str) -> List[int]:
        table = _byte_symbols()
        ids: List[int] = []
        for piece in self._split.findall(self.clean(text)):
            word = "".join(table[b] for b in piece.encode("utf-8"))
            ids.extend(self.encoder[s] for s in self._merge_word(word))
        return ids

    def decode(self, ids: Sequence[int]) -> str:
        inv = {v: k for k, v in _byte_symbols().items()}
        text = "".join(self.decoder[int(i)] for i in ids)
        return bytearray(inv[c] for c in text).decode("utf-8", errors="replace").replace("</w>", " ")

    def __call__(self, texts: Union[str, Sequence[str]], context_length: int = 0) -> torch.Tensor:
        if isinstance(texts, str):
            texts = [texts]
        L = context_length or self.context_length
        out = torch.zeros(len(texts), L, dtype=torch.int64)
        for r, t in enumerate(texts):
            ids = [self.sot] + self.encode(t) + [self.eot]
            if len(ids) > L:
                ids = ids[:L]
                ids[-1] = self.eot
            out[r, :len(ids)] = torch.tensor(ids, dtype=torch.int64)
        return out


def _order_key(b: int) -> Tuple[int, int]:
    # vocabulary order of the 256 byte symbols: the printable bytes first (in byte order), then the 68 remapped ones
    s = _byte_symbols()[b]
    return (0, b) if ord(s) == b else (1, ord(s))


class CLIPTextTransformer(nn.Module):
    """open_clip `CLIP.encode_text` (normalize=False): int64 tokens [B, L <= context_length] -> features
    [B, output_dim] in the parameter dtype."""

    def __init__(self, context_length=77, vocab_size=49408, width=1024, heads=16, layers=24, mlp_ratio=4.0,
                 output_dim=1024):
        super().__init__()
        self.config = dict(context_length=context_length, vocab_size=vocab_size, width=width, heads=heads,
                           layers=layers, mlp_ratio=mlp_ratio, output_dim=output_dim)
        if context_length > 128:
            raise NotImplementedError("pgn_attention_causal_fwd handles sequences of at most 128 tokens")
        self.token_embedding = nn.Embedding(vocab_size, width)
        self.positional_embedding = nn.Parameter(0.01 * torch.randn(context_length, width))
        self.transformer = _Transformer(width, layers, heads, mlp_ratio)
        self.ln_final = nn.LayerNorm(width)
        self.text_projection = nn.Parameter(width ** -0.5 * torch.randn(width, output_dim))

    @property
    def dtype(self):
        return self.token_embedding.weight.dtype

    @classmethod
    def from_pretrained(cls, path: str, torch_dtype=None, **config):
        """`open_clip_pytorch_model.bin` / `.safetensors` of laion/CLIP-ViT-H-14-laion2B-s32B-b79K: the text half's
        tensors are kept; `visual.*`, `logit_scale` and the `attn_mask` buffer are ignored."""
        if path.endswith(".safetensors"):
            from safetensors.torch import load_file
            sd = load_file(path)
        else:
            sd = torch.load(path, map_location="cpu", weights_only=True)
        keep = ("token_embedding.", "positional_embedding", "transformer.", "ln_final.", "text_projection")
        sd = {k: v for k, v in sd.items() if k.startswith(keep)}
        model = cls(**{**TEXT_VIT_H_14, **config})
        model.load_state_dict(sd)
        return (model.to(torch_dtype) if torch_dtype is not None else model).eval()

    def forward(self, text: torch.Tensor) -> torch.Tensor:
        if text.dtype != torch.int64 or text.dim() != 2 or text.shape[1] > self.config["context_length"]:
            raise ValueError("text must be int64 [B, L <= context_length] (SimpleTokenizer's output)")
        if text.device.type != "cuda":
            raise RuntimeError("CLIPTextTransformer runs on an MI355X (device 'cuda:N'): no CPU fallback")
        B, L = text.shape
        x = (self.token_embedding(text) + self.positional_embedding[:L]).contiguous()
        pending = None                                                    # mlp output not yet added to x
        for blk in self.transformer.resblocks:
            x, h = _layer_norm(x, pending, blk.ln_1)                      # x += mlp(prev) ; h = ln_1(x)
            a = blk.attn(h, causal=True)
            x, h = _layer_norm(a, x, blk.ln_2)                            # x += attn ; h = ln_2(x)
            pending = blk.mlp(h)
        rows = torch.arange(B, device=text.device)
        eot = text.argmax(dim=-1)                                         # <end_of_text> has the highest id
        x = (x[rows, eot] + pending[rows, eot]).contiguous()              # LayerNorm is per row: only these are read out
        _, pooled = _layer_norm(x, None, self.ln_final)
        return pooled @ self.text_projection.to(pooled.dtype)
