"""bench.py's rank handling, on a host without a GPU: it must never print a line whose n_gpus differs from
--gpus, and --gpus N without a launcher must start N fresh rank processes."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(torch.cuda.is_available(), reason="exercises the no-GPU exits of bench.py")


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                          text=True, timeout=300)


def test_world_size_must_equal_gpus():
    r = _run(["--gpus", "8"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "refusing to print" in r.stderr and r.stdout.strip() == ""


def test_gpus_without_a_launcher_starts_that_many_ranks():
    r = _run(["--gpus", "3", "--steps", "1"])
    # every rank gets as far as the device check and fails loudly there: there is no CPU fallback
    assert r.returncode != 0
    assert r.stderr.count("needs an MI355X") == 3 and r.stdout.strip() == ""


def _ident(rank, pci, host="box", uuid=""):
    return dict(rank=rank, local_rank=rank, host=host, pci=pci, uuid=uuid or pci, name="gpu", cus=256)


def test_a_multi_gpu_line_needs_as_many_different_devices_as_ranks():
    sys.path.insert(0, ROOT)
    import bench
    ok = bench.check_ranks([_ident(1, "0000:0b:00"), _ident(0, "0000:0a:00")], 2, 2)
    assert [r["rank"] for r in ok] == [0, 1] and ok[0]["pci"] == "0000:0a:00"
    # the same PCI address on another host is another device
    assert len(bench.check_ranks([_ident(0, "0000:0a:00", "a"), _ident(1, "0000:0a:00", "b")], 2, 2)) == 2
    with pytest.raises(SystemExit, match="same device"):
        bench.check_ranks([_ident(0, "0000:0a:00"), _ident(1, "0000:0a:00")], 2, 2)
    with pytest.raises(SystemExit, match="refusing to print"):
        bench.check_ranks([_ident(0, "0000:0a:00")], 2, 1)              # world size differs from --gpus
    with pytest.raises(SystemExit, match="refusing to print"):
        bench.check_ranks([_ident(0, "0000:0a:00")], 2, 2)              # a rank did not report
    with pytest.raises(SystemExit, match="are not 0..1"):
        bench.check_ranks([_ident(0, "0000:0a:00"), _ident(0, "0000:0b:00")], 2, 2)


def test_normalise_is_refused_for_configs_that_do_not_have_the_stage():
    r = _run(["--config", "rcnn", "--normalise"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "--normalise applies" in r.stderr and r.stdout.strip() == ""


def test_two_rank_rehearsal_runs_the_multi_rank_bookkeeping_on_gloo():
    """SEDMI_BENCH_REHEARSAL=cpu: the code an N > 1 run goes through around its steps -- starting the ranks,
    rendezvous, the all-gathered device check, the asynchronous all-gather of the predictions and the statistics
    all-reduce every step, per-rank times, the JSON line -- executed with two processes on gloo and a stand-in step
    that runs no kernel.  The line is marked as a rehearsal and carries no throughput."""
    import json
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--normalise"], {"SEDMI_BENCH_REHEARSAL": "cpu"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                       # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["rehearsal"] is True and d["value"] is None and "REHEARSAL" in d["metric"]
    assert d["n_gpus"] == 2 and d["world_size"] == 2 and d["steps"] == 3
    assert [x["rank"] for x in d["ranks_seen"]] == [0, 1] and d["ranks_seen"][0]["pci"] != d["ranks_seen"][1]["pci"]
    assert len(d["ms_per_step_ranks"]["all"]) == 2 and d["ms_per_step_ranks"]["max"] >= d["ms_per_step_ranks"]["min"] > 0
    names = [c["name"] for c in d["collectives"]]
    assert any("all_reduce" in n for n in names) and any("all_gather" in n for n in names)
    gather = next(c for c in d["collectives"] if "all_gather" in c["name"])
    assert gather["bytes_per_rank_per_step"] == 8 * 250 * 10 * 4 and gather["blocking"] is False
    # a launcher's environment (the way the driver starts it) works the same
    r1 = _run(["--gpus", "1", "--steps", "1"], {"SEDMI_BENCH_REHEARSAL": "cpu", "WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    d1 = json.loads(r1.stdout.strip().splitlines()[-1])
    assert r1.returncode == 0 and d1["n_gpus"] == 1 and d1["collectives"] == [] and len(d1["ranks_seen"]) == 1
