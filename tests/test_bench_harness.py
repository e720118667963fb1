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


@pytest.mark.parametrize("config,expect_sizes,scaling", [
    (3, [8000] * 8, "strong"),     # configs[3]: 64k scenarios cut over 8 GPUs, add brokers 1000-1049
    (4, [8] * 8, "weak"),          # configs[4]: 8 variants of the 1M x 5k x RF 5 cluster per GPU, as replicas
    (2, [1000] * 8, "weak"),       # configs[2]: the headline's batch per GPU
])
def test_bench_config_presets_at_world_8(config, expect_sizes, scaling):
    """`bench.py --gpus 8 --config N`: the BASELINE configs beyond the headline as one flag, so that a driver
    run on an 8-GPU node needs no hand-assembled command line.  Over gloo with the stub: shards, gather, the line's labels."""
    r = _run(["--stub", "--gpus", "8", "--steps", "2", "--warmup", "1", "--config", str(config)], timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _json_line(r.stdout)
    cfg = line["config"]
    assert line["n_gpus"] == 8 and line["scaling"] == scaling and cfg["scenarios_per_gpu"] == expect_sizes
    assert cfg["preset"] == config and f"configs[{config}]" in cfg["workload"]
    assert cfg["gathered_records_ok"] is True
    if config == 3:
        assert "add50" in cfg["workload"] and cfg["batches_in_flight"] == 2
    if config == 4:
        assert cfg["partitions"] == 1000000 and cfg["brokers"] == 5000 and cfg["rf"] == 5 and "c5_norack" in cfg["workload"]


def test_bench_config_preset_yields_to_explicit_flags():
    import bench
    a = bench.parse_args(["--config", "3", "--scenarios", "128", "--in-flight", "4"])
    assert (a.scenarios, a.in_flight, a.actions, a.scaling) == (128, 4, "add50", "strong")
    a = bench.parse_args(["--config", "3"])
    assert (a.scenarios, a.in_flight) == (64000, 1)                # one GPU: one batch in flight (154 GB of tables)
    a = bench.parse_args(["--config", "3", "--gpus", "8"])
    assert a.in_flight == 2
    a = bench.parse_args(["--config", "4"])
    assert (a.partitions, a.brokers, a.racks, a.rf, a.scenarios, a.actions) == (1000000, 5000, 40, 5, 8, "c5,c5_norack")
    a = bench.parse_args([])
    assert (a.config, a.scenarios, a.cells) == (0, 1000, 32)


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


def test_bench_tables_with_16_bit_cells_are_the_index_form_of_flatten():
    """bench.py keeps its tables resident as 16-bit node-index cells, converted on the device by cells16_table; the library's
    tests convert on the host with flatten.to_cells16.  Same cells."""
    import numpy as np
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from kafka_assigner_amd import generator as G
    from kafka_assigner_amd.flatten import node_set_batch, to_cells16
    S, P, N, R, RF = 70, 300, 40, 8, 3                       # (more scenarios than one conversion chunk)
    cur = np.stack([G.random_assignment(11 + s, P, N, R, RF) for s in range(S)])
    sets = [G.scenario_action(5, s, N, R, actions=G.BENCH_ACTIONS, max_add=6)[1] for s in range(S)]
    ids = [b.node_id for b in sets]
    fb = node_set_batch(ids, [b.node_rack for b in sets], P, RF, RF, cur=cur)
    got = bench.cells16_table(torch, torch.from_numpy(cur), ids, N, torch.device("cpu")).numpy().view(np.uint16).reshape(-1)
    want = to_cells16(fb)
    np.testing.assert_array_equal(got, want)
    assert (want == 0xFFFF).any() and (want != 0xFFFF).any()


def test_bench_collective_at_one_rank_over_gloo():
    """`--rccl-at-1`: the data-path collective inside the steps of a ONE-rank run (over gloo with the stub; over an RCCL communicator
    of one rank on the GPU box, tests/test_zz_bench_rccl_one_rank.py) - what a one-GPU box can exercise of the N > 1 path."""
    r = _run(["--stub", "--gpus", "1", "--steps", "4", "--warmup", "1", "--scenarios", "5", "--in-flight", "2", "--rccl-at-1"])
    assert r.returncode == 0, r.stderr[-3000:]
    line = _json_line(r.stdout)
    cfg = line["config"]
    assert line["n_gpus"] == 1 and cfg["world_size"] == 1 and cfg["scenarios_per_gpu"] == [5]
    assert "ONE rank" in cfg["collective"] and cfg["gathered_records_ok"] is True and cfg["allgather_alone_us"] > 0
    r = _run(["--stub", "--gpus", "1", "--steps", "2", "--warmup", "1", "--scenarios", "5", "--in-flight", "2"])
    assert r.returncode == 0 and _json_line(r.stdout)["config"]["collective"] == "none (1 GPU)"
