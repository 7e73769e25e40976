"""Caller side of the path (inference.py:345-458 batching + on-disk layout) and the N>1 sharding, on CPU.

The multi-rank cases run world_size-2 `gloo` process groups (the GPU path uses the same code over backend "nccl" =
RCCL); a stand-in `lavit` with the reference's generate_image signature replaces the GPU pipeline so that what is
tested is the runner: shard arithmetic, global-noise slicing, padded all-gather, file layout."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pigeon_amd import runner


class _FakeImage:
    def __init__(self, value):
        self.value = value

    def save(self, path):
        with open(path, "w") as f:
            f.write(repr(self.value))


class _FakeLavit:
    """generate_image(...) -> (images, prompt_embeds, pooled): image j = (sum of its token ids, checksum of the noise
    rows it was given), so any mistake in token slicing or noise slicing changes the files written."""

    class _U:
        class config:
            in_channels = 4

    unet = _U()
    device = torch.device("cpu")

    def __init__(self):
        self.calls = []

    def generate_image(self, image_tokens=None, num_return_images=1, width=512, height=512, original_size=None,
                       crops_coords_top_left=None, num_inference_steps=25, guidance_scale_for_diffusion=7.0,
                       generator=None, latents=None):
        n = len(image_tokens) * num_return_images
        if latents is None:
            latents = torch.randn((n, 4, height // 8, width // 8), generator=generator)
        assert latents.shape[0] == n
        self.calls.append((len(image_tokens), num_return_images))
        images = []
        for i in range(n):
            tok = image_tokens[i // num_return_images]
            images.append(_FakeImage((int(tok.sum()), round(float(latents[i].double().sum()), 6))))
        return images, None, None


class _FakeLatentLavit(_FakeLavit):
    """The PigeonPixelDecoder surface the runner uses to collect latents: generate_image(output_type="latent") returns
    the final latents (here: noise * 0.5 + token-sum / 1e6, so they depend on both inputs), decode_latents turns them
    into images."""

    def generate_image(self, image_tokens=None, num_return_images=1, width=512, height=512, original_size=None,
                       crops_coords_top_left=None, num_inference_steps=25, guidance_scale_for_diffusion=7.0,
                       generator=None, latents=None, output_type="pil"):
        assert output_type == "latent"
        n = len(image_tokens) * num_return_images
        if latents is None:
            latents = torch.randn((n, 4, height // 8, width // 8), generator=generator)
        self.calls.append((len(image_tokens), num_return_images))
        shift = torch.tensor([float(image_tokens[i // num_return_images].sum()) / 1e6 for i in range(n)])
        return latents * 0.5 + shift[:, None, None, None], None, None

    def decode_latents(self, latents, output_type="pil"):
        return [_FakeImage((0, round(float(x.double().sum()), 6))) for x in latents]


def _fake_scorer(images, text_features, history_features):
    """(token-sum of the image mod 97, its task's text feature): depends on which image and which task row it got."""
    return torch.tensor([[img.value[0] % 97, float(text_features[i, 0])] for i, img in enumerate(images)])


def _inputs(n_tasks, nrs, seed=0):
    g = torch.Generator().manual_seed(seed)
    tokens = [torch.randint(32002, 48386, (int(torch.randint(5, 30, (1,), generator=g)),), generator=g)
              for _ in range(n_tasks * nrs)]
    uids = list(range(100, 100 + n_tasks))
    genres = [i % 3 for i in range(n_tasks)]
    targets = [1000 + i for i in range(n_tasks)]
    history = [torch.arange(5) + i for i in range(n_tasks)]
    labels = [torch.arange(3) + i for i in range(n_tasks)]
    return tokens, labels, uids, genres, targets, history


def _read_tree(folder):
    out = {}
    for root, _, files in os.walk(folder):
        for f in files:
            if f.endswith(".jpg"):
                p = os.path.join(root, f)
                out[os.path.relpath(p, folder)] = open(p).read()
    return out


def test_shard_range_partitions_contiguously():
    for total in (0, 1, 7, 16, 64, 129):
        for world in (1, 2, 3, 8):
            spans = [runner.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and a <= b and c <= d
            per = -(-total // world) if total else 0
            assert all(hi - lo <= per for lo, hi in spans)


def test_generate_images_layout_matches_inference_py(tmp_path):
    """inference.py:413-458: {uid}/{target}/{genre}/{seq}/{j}.jpg, all_res[uid][target][genre]['seqs'][seq] with
    image paths relative to the eval folder, image index (i*nrs+seq)*nri + j, all_res_dm_scale{g}.npy."""
    nrs, nri, n_tasks, bs = 2, 3, 5, 2
    tokens, labels, uids, genres, targets, history = _inputs(n_tasks, nrs)
    lavit = _FakeLavit()
    save = str(tmp_path / "eval_x" / "out")
    res = runner.generate_images(lavit, tokens, labels, uids, genres, targets, history, num_return_sequences=nrs,
                                 batch_size=bs, save_folder=save, eval_name="eval_x", num_return_images=nri,
                                 width=64, height=64, guidance_scale_for_dm=7.0,
                                 generator=torch.Generator().manual_seed(123))
    assert lavit.calls == [(4, 3), (4, 3), (2, 3)]            # ceil(10 / (2*2)) outer batches (inference.py:376-377)
    assert sorted(res) == uids
    for i, uid in enumerate(uids):
        rec = res[uid][targets[i]][genres[i]]
        assert torch.equal(rec["history"], history[i]) and torch.equal(rec["labels"], labels[i])
        assert sorted(rec["seqs"]) == list(range(nrs))
        for s in range(nrs):
            seq = rec["seqs"][s]
            assert torch.equal(seq["tokens"], tokens[i * nrs + s])
            assert seq["images"] == [f"out/dm_scale7.0/{uid}/{targets[i]}/{genres[i]}/{s}/{j}.jpg" for j in range(nri)]
            for j in range(nri):
                body = open(os.path.join(str(tmp_path / "eval_x"), seq["images"][j])).read()
                assert body.startswith(f"({int(tokens[i * nrs + s].sum())},")
    loaded = np.load(os.path.join(save, "all_res_dm_scale7.0.npy"), allow_pickle=True).item()  # our own file
    assert sorted(loaded) == uids


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, tmp, n_tasks, nrs, nri, bs):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tokens, labels, uids, genres, targets, history = _inputs(n_tasks, nrs)
        # (1) sharded run writes this rank's files
        runner.generate_images(_FakeLavit(), tokens, labels, uids, genres, targets, history,
                               num_return_sequences=nrs, batch_size=bs, save_folder=os.path.join(tmp, "w2"),
                               eval_name="e", num_return_images=nri, width=64, height=64, rank=rank,
                               world_size=world, seed=123)
        # (1b) scored run: every rank scores its own images, rows are all-gathered per outer batch, rank 0 writes the
        #      select_scores file; the result must equal the single-rank scoring of the same images
        _, sel = runner.generate_images(_FakeLavit(), tokens, labels, uids, genres, targets, history,
                                        num_return_sequences=nrs, batch_size=bs, save_folder=os.path.join(tmp, "w2s"),
                                        eval_name="e", num_return_images=nri, width=64, height=64, rank=rank,
                                        world_size=world, seed=123, scorer=_fake_scorer,
                                        text_features=torch.arange(n_tasks, dtype=torch.float32)[:, None])
        torch.save(sel, os.path.join(tmp, f"sel_rank{rank}.pt"))
        # (1c) a pipeline that hands over latents: final latents and scores travel in ONE all-gather per outer batch
        #      and every rank ends up with the latents of the whole run, in global image order
        res, sel2, lat = runner.generate_images(_FakeLatentLavit(), tokens, labels, uids, genres, targets, history,
                                                num_return_sequences=nrs, batch_size=bs,
                                                save_folder=os.path.join(tmp, "w2l"), eval_name="e",
                                                num_return_images=nri, width=64, height=64, rank=rank,
                                                world_size=world, seed=123, scorer=lambda im, tf, hf: torch.tensor(
                                                    [[x.value[1], float(tf[i, 0])] for i, x in enumerate(im)]),
                                                text_features=torch.arange(n_tasks, dtype=torch.float32)[:, None],
                                                return_latents=True)
        torch.save((res, sel2, lat), os.path.join(tmp, f"lat_rank{rank}.pt"))
        # (2) padded all-gather with uneven counts: rank r contributes r+2 rows
        counts = [r + 2 for r in range(world)]
        lat = torch.full((counts[rank], 4, 2, 2), float(rank + 1))
        sc = torch.arange(counts[rank] * 2, dtype=torch.float32).reshape(-1, 2) + 100 * rank
        g_lat, g_sc = runner.all_gather_results(lat, sc, counts)
        assert g_lat.shape[0] == sum(counts) and g_sc.shape == (sum(counts), 2)
        off = 0
        for r in range(world):
            assert torch.all(g_lat[off:off + counts[r]] == r + 1)
            assert torch.equal(g_sc[off:off + counts[r]],
                               torch.arange(counts[r] * 2, dtype=torch.float32).reshape(-1, 2) + 100 * r)
            off += counts[r]
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_world2_gloo_sharded_run_equals_single_rank_run(tmp_path):
    """ranks draw the GLOBAL noise from the same seed and slice their rows, so the union of the
    files written by 2 ranks equals the files of the 1-rank run (same generator seed, inference.py:317)."""
    n_tasks, nrs, nri, bs = 5, 2, 2, 3
    tokens, labels, uids, genres, targets, history = _inputs(n_tasks, nrs)
    runner.generate_images(_FakeLavit(), tokens, labels, uids, genres, targets, history, num_return_sequences=nrs,
                           batch_size=bs, save_folder=str(tmp_path / "w1"), eval_name="e", num_return_images=nri,
                           width=64, height=64, generator=torch.Generator().manual_seed(123))
    port = _free_port()
    mp.spawn(_rank_main, args=(2, port, str(tmp_path), n_tasks, nrs, nri, bs), nprocs=2, join=True)
    one, two = _read_tree(str(tmp_path / "w1")), _read_tree(str(tmp_path / "w2"))
    assert len(one) == n_tasks * nrs * nri
    assert one == two
    # scores: gathered select_scores of both ranks == single-rank select_scores, in cal_scores.py's layout
    _, sel1 = runner.generate_images(_FakeLavit(), tokens, labels, uids, genres, targets, history,
                                     num_return_sequences=nrs, batch_size=bs, save_folder=str(tmp_path / "w1s"),
                                     eval_name="e", num_return_images=nri, width=64, height=64,
                                     generator=torch.Generator().manual_seed(123), scorer=_fake_scorer,
                                     text_features=torch.arange(n_tasks, dtype=torch.float32)[:, None])
    assert torch.load(tmp_path / "sel_rank0.pt") == sel1 == torch.load(tmp_path / "sel_rank1.pt")
    rec = sel1[uids[3]][targets[3]][genres[3]][1]
    assert rec["hist_score"] == [3.0] * nri and rec["tgt_score"] == [float(int(tokens[3 * nrs + 1].sum()) % 97)] * nri
    on_disk = np.load(tmp_path / "w2s" / "select_scores_dm_scale7.0.npy", allow_pickle=True).item()
    assert on_disk == sel1
    # all_res: ONE reference-format file, written by rank 0 from the merged fragments, equal to the 1-rank file
    assert not os.path.exists(tmp_path / "w2" / "all_res_dm_scale7.0.rank0.npy")
    res1 = np.load(tmp_path / "w1" / "all_res_dm_scale7.0.npy", allow_pickle=True).item()
    res2 = np.load(tmp_path / "w2" / "all_res_dm_scale7.0.npy", allow_pickle=True).item()
    assert list(res1) == list(res2) == uids          # task order survives the merge
    for i, uid in enumerate(uids):
        a, b = res1[uid][targets[i]][genres[i]], res2[uid][targets[i]][genres[i]]
        assert torch.equal(a["history"], b["history"]) and torch.equal(a["labels"], b["labels"])
        assert sorted(a["seqs"]) == sorted(b["seqs"]) == list(range(nrs))
        for sq in range(nrs):
            assert torch.equal(a["seqs"][sq]["tokens"], b["seqs"][sq]["tokens"])
            assert [x.replace("/w1/", "/w2/") for x in a["seqs"][sq]["images"]] == b["seqs"][sq]["images"]
    # latents: the 2-rank run returns, on BOTH ranks, the latents of the 1-rank run (same seed, global noise sliced)
    res_1, sel_1, lat_1 = runner.generate_images(
        _FakeLatentLavit(), tokens, labels, uids, genres, targets, history, num_return_sequences=nrs, batch_size=bs,
        save_folder=str(tmp_path / "w1l"), eval_name="e", num_return_images=nri, width=64, height=64,
        generator=torch.Generator().manual_seed(123), scorer=lambda im, tf, hf: torch.tensor(
            [[x.value[1], float(tf[i, 0])] for i, x in enumerate(im)]),
        text_features=torch.arange(n_tasks, dtype=torch.float32)[:, None], return_latents=True)
    assert lat_1.shape == (n_tasks * nrs * nri, 4, 8, 8) and lat_1.dtype == torch.float32
    for r in range(2):
        res_r, sel_r, lat_r = torch.load(tmp_path / f"lat_rank{r}.pt")
        assert torch.equal(lat_r, lat_1) and sel_r == sel_1 and list(res_r) == uids
    assert _read_tree(str(tmp_path / "w1l")) == _read_tree(str(tmp_path / "w2l"))
    with pytest.raises(ValueError):
        runner.generate_images(_FakeLavit(), tokens, labels, uids, genres, targets, history, return_latents=True)


def test_lowres_sharded_noise_comes_from_the_host_stream():
    # lavit_for_pigeon.py:858-860: the lowres decoder draws noise from a CPU generator; the sharded runner must draw
    # the global noise from the same stream (here world_size 2 simulated rank by rank without a process group)
    calls = []

    class Lowres(_FakeLavit):
        pixel_decoding = "lowres"
        device = torch.device("meta")      # any non-CPU device: the generator must still be a CPU one

        def generate_image(self, latents=None, generator=None, **kw):
            calls.append((latents.device.type, generator.device.type))
            raise StopIteration

    tokens, labels, uids, genres, targets, history = _inputs(2, 1)
    with pytest.raises(StopIteration):
        runner.generate_images(Lowres(), tokens, labels, uids, genres, targets, history, batch_size=2, width=64,
                               height=64, rank=0, world_size=2, seed=5)
    assert calls == [("cpu", "cpu")]
