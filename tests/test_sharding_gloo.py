"""The N > 1 path on CPU: two processes over gloo shard a batch by scenario, solve their shards
independently and all-gather the 32-byte result records — the same code path bench.py runs over
RCCL (kafka_assigner_amd/sharding.py).  The per-shard solve here is the oracle (no GPU in this
container); what is under test is the sharding arithmetic, the record layout and the collective."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, n_scenarios: int, q):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    from kafka_assigner_amd import sharding
    from oracle_lib import oracle_solve
    from test_emu_parity import _batch
    from kafka_assigner_amd import generator as G
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = sharding.shard_range(n_scenarios, rank, world)
        full = _batch(4242, n_scenarios, 600, 30, 6, 3, G.ACTIONS)
        # this rank's shard: scenarios [lo, hi) as a batch of their own (same seeds)
        shard = _batch(4242, n_scenarios, 600, 30, 6, 3, G.ACTIONS)
        shard.scen = shard.scen[lo:hi].copy()
        shard.scen["topic_begin"] -= lo
        shard.topics = shard.topics[lo:hi].copy()
        out0 = int(shard.topics["out_off"][0]) if hi > lo else 0
        cur0 = int(shard.topics["cur_off"][0]) if hi > lo else 0
        shard.topics["out_off"] -= out0
        shard.topics["cur_off"] -= cur0
        shard.cur = full.cur[cur0:cur0 + (hi - lo) * 600 * 3].copy()
        shard.out_len = (hi - lo) * 600 * 3
        local = oracle_solve(shard).scenario_results[:hi - lo]
        buf = torch.from_numpy(local.view(np.uint8).copy())
        gathered = sharding.gather_records(buf, n_scenarios)
        rec = sharding.records_view(gathered)
        want = oracle_solve(full).scenario_results[:n_scenarios]
        ok = all((rec[f] == want[f]).all() for f in ("status", "fail_partition", "moved_replicas",
                                                       "moved_partitions", "digest"))
        q.put((rank, bool(ok), int(rec.shape[0])))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_scenarios", [6, 7])   # even shards and ragged shards
def test_two_ranks_shard_solve_and_all_gather_records(n_scenarios):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_scenarios, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0, "a rank died"
    got = sorted(q.get(timeout=10) for _ in range(2))
    assert got == [(0, True, n_scenarios), (1, True, n_scenarios)]


def test_shard_ranges_cover_the_batch_exactly():
    from kafka_assigner_amd import sharding
    for n in (0, 1, 7, 64, 1000, 64000):
        for world in (1, 2, 3, 8):
            r = [sharding.shard_range(n, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1 and sizes == sharding.shard_sizes(n, world)
