"""CPU suite: `python bench.py --gpus N` from a plain shell starts its own N ranks (the
reference's multi-GPU switch is the broken WORLD_SIZE branch of Pigeon/inference.py:120-124), and the version-keyed
tuning tables say so when the running libraries refuse them."""
import json
import os
import subprocess
import sys
import textwrap
import time
import warnings

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (import only: no GPU call at import time)


def _stub(tmp_path, body):
    p = tmp_path / "rank_stub.py"
    p.write_text(textwrap.dedent(body))
    return str(p)


def test_launcher_gives_every_rank_its_environment_and_relays_rank_zero_only(tmp_path, capfd):
    stub = _stub(tmp_path, """
        import json, os, sys
        keys = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")
        rec = {k: os.environ.get(k) for k in keys}
        rec["argv"] = sys.argv[1:]
        open(os.path.join(os.environ["STUB_OUT"], "rank%s.json" % rec["RANK"]), "w").write(json.dumps(rec))
        print(json.dumps({"from_rank": int(rec["RANK"])}))     # rank 0: relayed; others: must not reach stdout
    """)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["STUB_OUT"] = str(tmp_path)
    rc = bench.launch_ranks(3, ["--gpus", "3", "--steps", "1"], script=stub, env=env)
    out = capfd.readouterr()
    assert rc == 0
    lines = [ln for ln in out.out.splitlines() if ln.strip()]
    assert lines == ['{"from_rank": 0}']                        # ONE line on stdout
    assert '{"from_rank": 1}' in out.err and '{"from_rank": 2}' in out.err
    recs = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(3)]
    assert [r["RANK"] for r in recs] == ["0", "1", "2"] and [r["LOCAL_RANK"] for r in recs] == ["0", "1", "2"]
    assert {r["WORLD_SIZE"] for r in recs} == {"3"} and {r["MASTER_ADDR"] for r in recs} == {"127.0.0.1"}
    assert len({r["MASTER_PORT"] for r in recs}) == 1 and int(recs[0]["MASTER_PORT"]) > 0
    assert recs[1]["argv"] == ["--gpus", "3", "--steps", "1"]


def test_launcher_returns_a_failed_ranks_code_and_stops_the_ranks_that_wait_for_it(tmp_path):
    stub = _stub(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(7)
        time.sleep(600)          # a rank stuck in the rendezvous of a job whose peer died
    """)
    env = {k: v for k, v in os.environ.items() if k != "WORLD_SIZE"}
    t0 = time.time()
    rc = bench.launch_ranks(2, [], script=stub, env=env)
    assert rc == 7 and time.time() - t0 < 60


def test_two_rank_gloo_job_through_the_launcher(tmp_path, capfd):
    """The launcher's environment is enough for a real torch.distributed rendezvous (gloo, CPU): each rank joins,
    all-gathers its rank through runner.all_gather_results, and rank 0 reports the world size it saw."""
    stub = _stub(tmp_path, f"""
        import json, os, sys
        sys.path.insert(0, {ROOT!r})
        import torch, torch.distributed as dist
        from pigeon_amd import runner
        dist.init_process_group("gloo")
        r, w = dist.get_rank(), dist.get_world_size()
        lat, _ = runner.all_gather_results(torch.full((1, 4, 2, 2), float(r)), None, [1] * w)
        if r == 0:
            print(json.dumps({{"n_gpus": w, "rows": lat[:, 0, 0, 0].tolist()}}))
        dist.destroy_process_group()
    """)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    rc = bench.launch_ranks(2, [], script=stub, env=env)
    out = capfd.readouterr().out
    assert rc == 0
    assert json.loads(out.strip().splitlines()[-1]) == {"n_gpus": 2, "rows": [0.0, 1.0]}


def test_plain_bench_with_two_gpus_becomes_a_launcher_before_any_gpu_call():
    """No GPU in the CPU container: the two children must be the ones that report it (the parent never asks)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline", "--dist-backend", "gloo", "--single-device"],
                       env=env, capture_output=True, text=True, timeout=300)
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU box: the real two-rank rehearsal is tests/test_gpu_runner_dist.py")
    assert r.returncode != 0
    assert "bench.py needs an MI355X" in r.stderr and "stopping the other ranks" in r.stderr
    assert r.stdout.strip() == ""


def test_world_size_must_match_gpus_flag():
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "--gpus 8 but WORLD_SIZE=4" in r.stderr


# ---- version-keyed tuning artefacts (call sites lavit_for_pigeon.py:187-191) --------------

class _FakeTunable:
    """torch.cuda.tunable needs a GPU; this stand-in applies its rule: a table is accepted only when every Validator
    row equals the running library's."""

    def __init__(self, running):
        self.running, self.enabled = running, False

    def is_enabled(self):
        return self.enabled

    def enable(self, v):
        self.enabled = v

    def tuning_enable(self, v):
        pass

    def record_untuned_enable(self, v):
        pass

    def get_validators(self):
        return tuple(self.running.items())

    def read_file(self, path):
        import pigeon_amd
        return pigeon_amd._csv_validators(path) == self.running


def test_doctored_validator_row_makes_the_gemm_table_warn(tmp_path):
    import pigeon_amd
    shipped = os.path.join(ROOT, "pigeon_amd", "tunableop", "gfx950_bf16.csv")
    running = pigeon_amd._csv_validators(shipped)
    assert {"PT_VERSION", "HIPBLASLT_VERSION", "ROCBLAS_VERSION", "GCN_ARCH_NAME"} <= set(running)
    with warnings.catch_warnings():
        warnings.simplefilter("error")                      # the matching table loads silently
        assert pigeon_amd.enable_tuned_gemms(shipped, _tn=_FakeTunable(running)) is True
    pigeon_amd._gemm_tables.clear()
    doctored = tmp_path / "gfx950_bf16.csv"
    text = open(shipped).read().replace("Validator,HIPBLASLT_VERSION," + running["HIPBLASLT_VERSION"],
                                        "Validator,HIPBLASLT_VERSION,100100-20260101-1-gdeadbeef")
    doctored.write_text(text)
    with pytest.warns(UserWarning, match="refused the shipped GEMM table") as rec:
        assert pigeon_amd.enable_tuned_gemms(str(doctored), _tn=_FakeTunable(running)) is False
    msg = str(rec[0].message)
    assert "HIPBLASLT_VERSION" in msg and "gdeadbeef" in msg and "tune_gemm.py" in msg
    assert pigeon_amd.gemm_table_mismatches(str(doctored), running) == [
        f"HIPBLASLT_VERSION: table '100100-20260101-1-gdeadbeef', running {running['HIPBLASLT_VERSION']!r}"]
    pigeon_amd._gemm_tables.clear()


def test_find_db_of_another_miopen_build_warns(tmp_path):
    import pigeon_amd
    shipped = os.path.join(ROOT, "pigeon_amd", "miopen_db")
    name = [f for f in os.listdir(shipped) if f.endswith(".ufdb.txt")][0]
    arch_cu, _, ver = name[:-len(".ufdb.txt")].partition(".HIP.")
    maj, mnr, pat, build = ver.split("_", 3)
    running = {"arch": "gfx950", "cus": 256, "version": (int(maj), int(mnr), int(pat)), "build": build}
    assert arch_cu == "gfx950100"
    st = pigeon_amd.miopen_db_status(running, shipped)
    assert st["hit"] and st["expected"] == name
    assert pigeon_amd.miopen_db_status(dict(running, build=None), shipped)["hit"]       # build id unknown: prefix match
    for other in (dict(running, build="20260101-1-gdeadbeef"), dict(running, version=(3, 6, 0)), dict(running, cus=304)):
        assert not pigeon_amd.miopen_db_status(other, shipped)["hit"]
    pigeon_amd._miopen_warned = False
    with pytest.warns(UserWarning, match="no MIOpen find-db for this MIOpen build") as rec:
        assert pigeon_amd.warn_if_miopen_db_missed(dict(running, build="20260101-1-gdeadbeef"), shipped) is False
    assert "tune_miopen.py" in str(rec[0].message) and "gdeadbeef" in str(rec[0].message)
    with warnings.catch_warnings():
        warnings.simplefilter("error")                      # once per process
        assert pigeon_amd.warn_if_miopen_db_missed(dict(running, build="20260101-1-gdeadbeef"), shipped) is False
        pigeon_amd._miopen_warned = False
        assert pigeon_amd.warn_if_miopen_db_missed(running, shipped) is True            # a hit never warns
