"""bench.py's rank function on CPU ranks (gloo, world_size 2) with a stand-in for the device side, and its self-launch
path.  What is under test is the part the GPU cannot check for us before the driver's N > 1 run: the timing protocol
(barriers, MAX over ranks), the hypothesis-range sharding + the ONE all-reduce(MAX), every rank decoding the same
winner, and the shape of the JSON line (n_gpus, the world size each rank saw)."""
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


class _Ev:
    def record(self):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


class CpuStandIn:
    """Device side of bench.run_rank without a device: the warp "launch" sleeps, the search returns the packed keys
    of this rank's slice of the REFERENCE's 100 000 counts (tests/golden/g10_config5_search.npz)."""
    collective = "gloo"
    rehearsal = False

    def __init__(self):
        from ransac_with_homography_amd import kernels, sharded
        self.kernels, self.sharded, self.torch = kernels, sharded, torch
        self.dev = torch.device("cpu")

    def sync(self):
        pass

    def tensor(self, values, dtype="int64"):
        return torch.tensor(values, dtype=getattr(torch, dtype))

    def events(self):
        return _Ev(), _Ev()

    def make_warp(self, frames, src_w, src_h, seed, rows_of=None):
        out_h, out_w = src_h - 100, src_w - 60
        rows = (0, out_h) if rows_of is None else self.sharded.shard_range(out_h, *rows_of)
        return (lambda: time.sleep(2e-4)), out_h, out_w, rows, "stand-in", None

    def make_search(self, K, b, e, settle=False):
        counts = np.load(os.path.join(ROOT, "tests", "golden", "g10_config5_search.npz"))["counts"].astype(np.int64)
        assert K == counts.size
        need = self.kernels.need_count(185, 70, 4)
        i = np.arange(b, e)
        w0 = int(np.max((counts[b:e] << 32) | (0xFFFFFFFF - i)))
        hit = i[counts[b:e] >= need]
        w1 = int(0xFFFFFFFF - hit[0]) if hit.size else 0
        return lambda: torch.tensor([w0, w1], dtype=torch.int64)


def _rank(rank, world, port, path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    args = bench.parse_args(["--gpus", str(world), "--steps", "4", "--warmup", "1", "--frames", "2", "--src", "640x480"])
    bench.PREWARM_MS = 0.0
    line = bench.run_rank(args, CpuStandIn(), dist)
    assert (line is None) == (rank != 0)
    if rank == 0:
        json.dump(line, open(path, "w"))
    dist.barrier()
    dist.destroy_process_group()


def test_run_rank_world2_gloo(tmp_path):
    port = 29500 + (os.getpid() % 400) + 731
    out = str(tmp_path / "line.json")
    mp.spawn(_rank, args=(2, port, out), nprocs=2, join=True)
    line = json.load(open(out))
    assert line["n_gpus"] == 2 and line["steps"] == 4 and line["scaling"] == "weak" and line["unit"] == "Mpix/s"
    assert line["rccl"] == {"backend": "gloo", "world_size_seen_by_rank": [2, 2], "all_ranks_agree_on_winner": True}
    r = line["ransac"]["K=100000"]
    z = np.load(os.path.join(ROOT, "tests", "golden", "g10_config5_search.npz"))
    assert r["winner"] == int(z["winner"]) == 99206 and r["winner_count"] == 122 and r["all_ranks_decoded"] == [99206, 99206]
    assert line["strong_8k"]["scaling"] == "strong" and line["strong_8k"]["rows_of_rank0"][0] == 0
    # two ranks, each 2 frames of 380 x 580 per step, 4 steps in ~1 ms: the value is the whole job's
    assert line["value"] > 0 and abs(line["value"] - 2 * 2 * 380 * 580 / 1e6 * 4 / (line["ms_per_step"] * 4e-3)) < 1e-3 * line["value"]
    assert "cpu_baseline" not in line and "warp_8k" not in line      # N = 1 legs


def test_single_rank_line_has_the_contract_keys():
    args = bench.parse_args(["--steps", "3", "--warmup", "1", "--frames", "2", "--src", "640x480", "--no-cpu", "--no-extras"])
    bench.PREWARM_MS = 0.0
    line = bench.run_rank(args, CpuStandIn(), None)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["config"]["workload"].startswith("transformImageH warp 640x480")
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"):
        assert key in line["roofline"], key


def test_plain_command_self_launches():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment must start its own ranks (and here, without
    GPUs, fail only when the ranks reach for them), not stop at argument handling."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu"],
                       capture_output=True, text=True, env=env, timeout=300)
    log = p.stdout + p.stderr
    assert "launch with torch.distributed.run" not in log and "was started inside a job" not in log
    assert p.returncode != 0                                   # no GPUs in this container
    assert "set_device" in log or "HIP" in log or "cuda" in log.lower(), log[-2000:]
    assert log.count("Traceback") >= 1 and "ChildFailedError" in log


def test_counter_rows_of_the_committed_profile():
    """bench.py attaches the committed PMC row to every other warp kernel of the line (round-3 verdict item 3: a limiter per kernel) and reads
    the headline kernel's traffic / VALU-busy figure from the same file: the file must hold what those lookups expect."""
    import json
    import bench
    doc = json.load(open(os.path.join(ROOT, bench.PMC_JSON)))
    head = doc["rwh::warp_rgb8_fast8<unsigned char, 6>"]
    assert tuple(head["src"]) == (3840, 2160) and head["frames"] == 32
    assert 0.99 < head["hbm_bytes_per_launch"] / head["algorithmic_bytes_per_launch"] < 1.02          # no wasted HBM traffic
    assert head["valu_busy_frac"] > 0.9 and "limiter" in head
    for kernel in ("rwh::warp_rgb8_fast8<float, 6>", "rwh::warp_rgb8_nn<6>", "rwh::warp_rgb8_fast8h<6>", "rwh::warp_rgba8_fast8<6>",
                   "rwh::warp_generic<float, 4, float, 1>"):
        row = bench.pmc_row(kernel)["pmc"]
        assert row["limiter"] and row["source"] == bench.PMC_JSON and 0.0 < row["valu_busy"] <= 1.05, kernel
    assert bench.pmc_row("rwh::no_such_kernel") == {} and bench.pmc_row(None) == {}
    facts = bench.profile_facts("rwh::warp_rgb8_fast8<unsigned char, 6>", 32, (3840, 2160), 0.372, 1950.0)
    assert facts["traffic"] == head["hbm_bytes_per_launch"] and 0.85 < facts["valu"]["frac"] < 1.0
    assert bench.profile_facts("rwh::warp_rgb8_fast8<unsigned char, 6>", 32, (1920, 1080), 0.372) == {}     # another source size: not the profiled launch


def test_host_threads_follow_the_rank_share(monkeypatch):
    """The settle step's LAPACK loop takes this process's share of the cores it may run on: one rank per GPU under torchrun."""
    from ransac_with_homography_amd import ransac as impl
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(256)), raising=False)
    monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
    assert impl._host_thread_share() == 32
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    assert impl._host_thread_share() == 32
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(64)), raising=False)
    assert impl._host_thread_share() == 8
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "nonsense")
    assert impl._host_thread_share() == 32
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(3)), raising=False)
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    assert impl._host_thread_share() == 1
