"""Scoring stage, text side (Evaluation/eval_utils.py:115,126): byte-pair tokenizer, causal attention kernel, text tower.
PARITY UNPINNED against open_clip itself (absent here and from /root/reference, as is its merges file): the tokenizer
is cross-checked against transformers' independent CLIPTokenizer on a synthetic merges table, the kernel and the tower
against the fp32 oracle."""
import json
import random

import pytest
import torch

from oracle import ops_ref, scoring_ref, weights
from oracle.models_ref import Policy

TINY_TEXT = dict(context_length=77, vocab_size=600, width=128, heads=2, layers=2, mlp_ratio=4.0, output_dim=64)
CORPUS = ("a photo of a cat . the quick brown fox jumps over the lazy dog ! an oil painting of two dogs' heads , "
          "it's 12:30 in paris ; they're here & we've seen 3 movies - i'd say he'll like that one ? "
          "café naïve über straße 你好 привет movie poster of the matrix ( 1999 )")


def _train_merges(text, n):
    """A small BPE table learned from CORPUS the usual way (most frequent adjacent pair first)."""
    from pigeon_amd.clip_text import _byte_symbols
    table = _byte_symbols()
    words = {}
    for w in text.lower().split():
        sym = [table[b] for b in w.encode("utf-8")]
        sym[-1] += "</w>"
        words[tuple(sym)] = words.get(tuple(sym), 0) + 1
    merges = []
    for _ in range(n):
        counts = {}
        for w, c in words.items():
            for p in zip(w, w[1:]):
                counts[p] = counts.get(p, 0) + c
        if not counts:
            break
        best = max(sorted(counts), key=lambda p: counts[p])
        merges.append(best)
        new = {}
        for w, c in words.items():
            out, i = [], 0
            while i < len(w):
                if i + 1 < len(w) and (w[i], w[i + 1]) == best:
                    out.append(w[i] + w[i + 1]); i += 2
                else:
                    out.append(w[i]); i += 1
            new[tuple(out)] = new.get(tuple(out), 0) + c
        words = new
    return merges


@pytest.fixture(scope="module")
def tok_files(tmp_path_factory):
    d = tmp_path_factory.mktemp("bpe")
    merges = _train_merges(CORPUS, 80)
    (d / "merges.txt").write_text("#version: 0.2\n" + "\n".join(" ".join(m) for m in merges) + "\n", encoding="utf-8")
    return d, merges


def test_tokenizer_layout_and_round_trip(tok_files):
    from pigeon_amd.clip_text import N_MERGES, SimpleTokenizer
    d, merges = tok_files
    tok = SimpleTokenizer(str(d / "merges.txt"), n_merges=len(merges))
    assert tok.vocab_size == 512 + len(merges) + 2 and tok.sot == tok.vocab_size - 2 and tok.eot == tok.vocab_size - 1
    assert 512 + N_MERGES + 2 == 49408                       # the published vocabulary size with the full table
    assert tok.encoder["!"] == 0 and tok.encoder["!</w>"] == 256 and tok.encoder["".join(merges[0])] == 512
    out = tok(["A photo of a cat.", "", "  it's   12:30 in&amp;amp;out  "])
    assert out.shape == (3, 77) and out.dtype == torch.int64
    assert out[1].tolist()[:3] == [tok.sot, tok.eot, 0]      # empty text: start, end, padding
    row = out[0].tolist()
    n = row.index(tok.eot)
    assert row[0] == tok.sot and all(t == 0 for t in row[n + 1:]) and int(out[0].argmax()) == n
    assert tok.decode(row[1:n]).strip() == "a photo of a cat ."
    assert tok.decode(tok.encode("it's   12:30 in&amp;amp;out")).strip() == "it 's 1 2 : 3 0 in & out"   # unescaped twice
    long = tok(["cat " * 200])[0].tolist()
    assert len(long) == 77 and long[-1] == tok.eot and 0 not in long                 # truncated, end token kept
    assert tok(["cat"], context_length=8).shape == (1, 8)


def test_tokenizer_agrees_with_transformers_clip_tokenizer(tok_files):
    # transformers.CLIPTokenizer is an independent implementation of the same published algorithm; built over the SAME
    # synthetic merges table (vocab.json in the published id order) both must produce the same ids
    transformers = pytest.importorskip("transformers")
    from pigeon_amd.clip_text import SimpleTokenizer
    d, merges = tok_files
    tok = SimpleTokenizer(str(d / "merges.txt"), n_merges=len(merges))
    vocab = dict(tok.encoder)
    vocab["<|startoftext|>"] = vocab.pop("<start_of_text>")
    vocab["<|endoftext|>"] = vocab.pop("<end_of_text>")
    (d / "vocab.json").write_text(json.dumps(vocab), encoding="utf-8")
    hf = transformers.CLIPTokenizer(str(d / "vocab.json"), str(d / "merges.txt"))
    rnd = random.Random(0)
    words = CORPUS.split()
    texts = ["a photo of a cat.", "The Matrix (1999)", "it's 12:30 -- they're here!", "café über straße",
             "你好 привет", "x"]
    texts += [" ".join(rnd.choice(words) for _ in range(rnd.randint(1, 40))) for _ in range(40)]
    for t in texts:
        want = hf(t)["input_ids"]
        got = [tok.sot] + tok.encode(t) + [tok.eot]
        assert got == want, t


def test_text_tower_has_the_published_size_and_open_clip_names(tmp_path):
    from pigeon_amd.clip_text import TEXT_VIT_H_14, CLIPTextTransformer
    with torch.device("meta"):
        m = CLIPTextTransformer(**TEXT_VIT_H_14)
    # ViT-H-14 text half: 49408 x 1024 embedding, 77 x 1024 positions, 24 blocks of 12.6 M, ln_final, 1024^2 projection
    assert sum(p.numel() for p in m.parameters()) == 354_032_640
    keys = set(m.state_dict())
    for k in ("token_embedding.weight", "positional_embedding", "ln_final.weight", "text_projection",
              "transformer.resblocks.23.attn.in_proj_weight", "transformer.resblocks.0.mlp.c_fc.weight"):
        assert k in keys, k
    small = CLIPTextTransformer(**TINY_TEXT)
    sd = dict(weights.synth_like(small, seed=4))
    full = {**sd, "logit_scale": torch.tensor(4.6), "visual.proj": torch.zeros(3, 3), "attn_mask": torch.zeros(77, 77)}
    torch.save(full, tmp_path / "open_clip_pytorch_model.bin")
    got = CLIPTextTransformer.from_pretrained(str(tmp_path / "open_clip_pytorch_model.bin"), **TINY_TEXT)
    assert torch.equal(got.state_dict()["text_projection"], sd["text_projection"])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        got(torch.zeros(1, 77, dtype=torch.int64))


def test_causal_attention_oracle_known_answers():
    g = torch.Generator().manual_seed(0)
    q, k, v = (torch.randn(2, 9, 32, generator=g) for _ in range(3))
    o = ops_ref.causal_attention_ref(q, k, v, heads=2)
    assert torch.allclose(o[:, 0], v[:, 0], atol=1e-6)                    # the first query sees only the first key
    full = ops_ref.attention_ref(q, k, v, heads=2)
    assert torch.allclose(o[:, -1], full[:, -1], atol=1e-5)               # the last query sees every key
    o5 = ops_ref.causal_attention_ref(q[:, :5], k[:, :5], v[:, :5], heads=2)
    assert torch.allclose(o[:, :5], o5, atol=1e-6)                        # a prefix does not depend on what follows


# ---- GPU ---------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_causal_attention_kernel_matches_oracle(dtype):
    from pigeon_amd import ops
    g = torch.Generator().manual_seed(3)
    for B, H, N, D in ((3, 16, 77, 64), (1, 1, 1, 64), (2, 3, 128, 128), (2, 2, 65, 80), (4, 5, 20, 8)):
        qkv = (torch.randn(B, N, 3 * H * D, generator=g) * 1.5).to(dtype)
        q, k, v = qkv[..., :H * D], qkv[..., H * D:2 * H * D], qkv[..., 2 * H * D:]     # strided views, as the tower uses
        want = ops_ref.causal_attention_ref(q.float(), k.float(), v.float(), H, out_dtype=torch.float32)
        dq = qkv.cuda()
        got = ops.attention(dq[..., :H * D], dq[..., H * D:2 * H * D], dq[..., 2 * H * D:], H, causal=True)
        assert got.shape == (B, N, H * D) and got.dtype == dtype
        err = (got.float().cpu() - want).abs().max().item()
        # fp32 scores and probabilities over 16-bit inputs, ONE rounding of the output: half an ulp of the largest value
        bar = (2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11) * max(1.0, want.abs().max().item())
        assert err <= bar, (B, H, N, D, err, bar)
    x = torch.randn(1, 129, 64, device="cuda").to(dtype)
    with pytest.raises(NotImplementedError):
        ops.attention(x, x, x, 1, causal=True)                            # longer than the kernel's 128-token limit
    y = torch.randn(1, 16, 64, device="cuda").to(dtype)
    with pytest.raises(ValueError):
        ops.attention(y, x, x, 1, causal=True)                            # causal means self-attention


@pytest.mark.gpu
def test_text_tower_and_scorer_texts_on_gpu_match_oracle(tok_files):
    from pigeon_amd.clip_text import CLIPTextTransformer, SimpleTokenizer
    from pigeon_amd.clip_vision import CLIPScorer, CLIPVisionTransformer
    d, merges = tok_files
    tok = SimpleTokenizer(str(d / "merges.txt"), n_merges=len(merges))
    cfg = dict(TINY_TEXT, vocab_size=tok.vocab_size)
    texts = ["a photo of a cat.", "the quick brown fox jumps over the lazy dog", "", "movie poster of the matrix (1999) " * 9]
    tokens = tok(texts)
    for dt, bar in ((torch.float16, 0.001), (torch.bfloat16, 0.008)):     # measured relL2: see the printed line
        m = CLIPTextTransformer(**cfg).eval()
        sd = {k: v.to(dt) for k, v in weights.synth_like(m, seed=31).items()}
        m.load_state_dict(sd)
        m = m.to("cuda", dt).requires_grad_(False)
        with torch.no_grad():
            got = m(tokens.cuda()).float().cpu()
        want = scoring_ref.clip_text_forward(Policy(torch.float32), {k: v.float() for k, v in sd.items()},
                                             cfg["heads"], tokens)
        rel = ((got - want).norm() / want.norm()).item()
        print(f"clip text tower {dt} relL2 vs fp32 oracle {rel:.5f}")
        assert got.shape == (4, 64) and rel < bar
    # texts straight into the scorer: same scores as passing the text features
    vis = CLIPVisionTransformer(image_size=56, patch_size=14, width=128, layers=1, heads=2, mlp_ratio=4.0, output_dim=64)
    vis.load_state_dict(weights.synth_like(vis, seed=2))
    scorer = CLIPScorer(vis, "cuda", torch.bfloat16, text=m, tokenizer=tok)
    images = torch.randint(0, 256, (4, 56, 56, 3), generator=torch.Generator().manual_seed(1), dtype=torch.uint8)
    a = scorer.score(images, texts=texts)
    b = scorer.score(images, text_features=scorer.encode_text(texts))
    assert torch.equal(a[:, 0], b[:, 0]) and torch.isnan(a[:, 1]).all()
    with pytest.raises(RuntimeError):
        CLIPScorer(vis, "cuda").score(images, texts=texts)
