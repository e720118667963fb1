"""The N > 1 path's collective on the one GPU a test box has: `bench.py --gpus 1 --rccl-at-1` initialises an RCCL communicator
of ONE rank on cuda:0 (backend "nccl" IS RCCL on ROCm), gathers every slot's 32-byte result records on the slot's own
non-default stream before anything is timed (the pre-flight) and after every solve inside the timed steps, and checks the bytes
that came back.  It cannot show a second GPU or an xGMI link; it shows that the library loads beside libkas_hip.so, that the
communicator comes up with `device_id`, and that the all-gather and the solver's kernels share streams and hardware queues
(GPU_MAX_HW_QUEUES) without upsetting each other - with the parity check of the records against the CPU solvers still on.
(The file sorts LAST in the suite on purpose: under `pytest -x` a communicator that cannot come up on some box for reasons of
its network set-up must not keep the parity tests from running.)"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_steps_with_the_all_gather_on_a_one_rank_rccl_communicator():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "8", "--warmup", "2", "--repeats", "2",
                        "--scenarios", "64", "--partitions", "4000", "--brokers", "80", "--racks", "8", "--in-flight", "4",
                        "--no-extras", "--cpu-seconds", "1", "--rccl-at-1"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    cfg = line["config"]
    assert line["n_gpus"] == 1 and cfg["world_size"] == 1 and "ONE rank" in cfg["collective"], cfg
    assert cfg["gathered_records_ok"] is True and cfg["allgather_alone_us"] > 0, cfg
    assert line["value"] > 0 and line.get("stub") is not True
