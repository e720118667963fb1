"""bench.py's launch / shard / gather / report control flow on CPU: `--stub` swaps the HIP solve for
synthetic records and RCCL for gloo, everything else — self-spawning the ranks under
torch.distributed.run, the world-size checks, contiguous scenario shards (weak and strong), the
per-step all-gather of 32-byte records, max-over-ranks timing, the one JSON line — is the code
the GPU run executes.  The stub line says it is not a measurement."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(argv, env_extra=None, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + argv, cwd=ROOT, env=env, capture_output=True, text=True,
                          timeout=timeout)


def _json_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("scaling,scenarios,expect_sizes", [
    ("weak", 6, [6, 6]),          # 6 per rank
    ("strong", 7, [4, 3]),        # 7 in total: ragged shards
])
def test_bench_spawns_its_own_ranks_and_gathers_records(scaling, scenarios, expect_sizes):
    r = _run(["--stub", "--gpus", "2", "--steps", "5", "--warmup", "2", "--scenarios", str(scenarios),
              "--scaling", scaling, "--in-flight", "2"])
    assert r.returncode == 0, r.stderr[-3000:]
    line = _json_line(r.stdout)
    assert line["stub"] is True and "not a measurement" in line["metric"]
    assert line["n_gpus"] == 2 and line["steps"] == 5 and line["warmup"] == 2 and line["scaling"] == scaling
    cfg = line["config"]
    assert cfg["world_size"] == 2 and cfg["scenarios_per_gpu"] == expect_sizes
    assert cfg["scenarios_total"] == sum(expect_sizes)
    assert cfg["gathered_records_ok"] is True
    assert cfg["allgather_alone_us"] is not None and cfg["allgather_alone_us"] > 0
    assert line["value"] > 0 and line["ms_per_step"] > 0
    assert abs(line["value"] - sum(expect_sizes) * 5 / (line["ms_per_step"] * 5e-3)) < 1e-6 * line["value"]


@pytest.mark.parametrize("scaling,scenarios,expect_sizes", [
    ("weak", 3, [3] * 8),
    ("strong", 13, [2, 2, 2, 2, 2, 1, 1, 1]),
])
def test_bench_world_of_eight_over_gloo(scaling, scenarios, expect_sizes):
    """The driver's 8-GPU launch on CPU: eight ranks, weak and strong (ragged shards), every rank's records gathered."""
    r = _run(["--stub", "--gpus", "8", "--steps", "3", "--warmup", "1", "--scenarios", str(scenarios),
              "--scaling", scaling, "--in-flight", "2"], timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _json_line(r.stdout)
    cfg = line["config"]
    assert line["n_gpus"] == 8 and cfg["world_size"] == 8 and cfg["scenarios_per_gpu"] == expect_sizes
    assert cfg["gathered_records_ok"] is True and cfg["allgather_alone_us"] > 0


@pytest.mark.parametrize("world,scenarios,expect_sizes", [
    (3, 10, [4, 3, 3]),
    (5, 12, [3, 3, 2, 2, 2]),
])
def test_bench_odd_world_sizes_strong_scaling_ragged_shards(world, scenarios, expect_sizes):
    """World sizes the driver's 1 / 2 / 4 / 8 sweep does not have: the padded all-gather of ragged
    shards, the pre-flight gather on every slot's stream, every rank's records in global order."""
    r = _run(["--stub", "--gpus", str(world), "--steps", "4", "--warmup", "1", "--scenarios", str(scenarios),
              "--scaling", "strong", "--in-flight", "3"], timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _json_line(r.stdout)
    cfg = line["config"]
    assert line["n_gpus"] == world and cfg["world_size"] == world and cfg["scenarios_per_gpu"] == expect_sizes
    assert cfg["scenarios_total"] == scenarios and cfg["gathered_records_ok"] is True


def test_bench_single_rank_stub_line():
    r = _run(["--stub", "--steps", "3", "--warmup", "1", "--scenarios", "5"])
    assert r.returncode == 0, r.stderr[-3000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 1 and line["config"]["collective"].startswith("none")
    assert line["config"]["allgather_alone_us"] is None


def test_bench_refuses_a_world_size_that_is_not_gpus():
    """Launched with WORLD_SIZE=1 but --gpus 2 (what an external launcher mismatch looks like):
    no mislabelled line, non-zero exit."""
    r = _run(["--stub", "--gpus", "2", "--steps", "2", "--warmup", "0"],
             env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert "refusing" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_bench_refuses_more_gpus_than_devices():
    """Without --stub on this GPU-less container: `--gpus 2` must fail loudly, not run one rank."""
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("two devices present")
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "0"])
    assert r.returncode != 0
    assert "refusing" in r.stderr and "device" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
