"""Consistency stress of solves IN FLIGHT: every slot of bench.py solves the SAME inputs
(`--same-batch`), so all solves must leave identical scenario records (status, movement, digest of
every emitted cell).  Issues ROUNDS x slots solves back to back on the slots' streams (no host
synchronisation in between, as bench.py's timed region does) and counts, on each slot's own stream
right behind its solve, the scenarios whose record differs from a reference solve that ran alone and
is checked against the oracle.  This is the regime that brought out the round-2 P4 overtaking bug
(one wrong scenario solve in ~70,000, never with one batch alone, never on the emulator): lock-free
LDS protocols between wavefronts are only really tested with other work sharing the CUs.

  python scripts/stress_inflight.py [ROUNDS] [bench.py flags, e.g. --waves 1 --in-flight 12]
  python scripts/stress_inflight.py --suite [MIN_SOLVES]     # every kernel family, >= MIN_SOLVES solves each
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from kafka_assigner_amd import abi, generator as G  # noqa: E402
from kafka_assigner_amd.flatten import node_set_batch  # noqa: E402
from oracle_lib import oracle_solve  # noqa: E402

# The kernel families and plan variants a solve can launch, each at a shape that takes it (name, must appear
# in the plan's description — "!x": x must NOT appear —, bench.py flags).  The first is the headline shape (BASELINE.json configs[2]).
SUITE = [
    ("headline: int32 broker ids in HBM in, ids out — slim fill kernel (per-chunk histograms) + first fit in kas_p4_kernel + relaxation form of the order kernel with the ids in the LDS",
     "kas_order_relax_kernel<3>[tiles of 64 rows, ids in LDS, dword mid rows]", ["--in-flight", "12", "--cells", "32"]),
    ("the same with the packed 16-bit mid rows (KAS_PLAN_NO_MID32: three uint16 a row in acceptance order, where the default is one dword of sorted holders)",
     "!dword mid rows", ["--plan-flags", "1048576", "--in-flight", "12", "--cells", "32"]),
    ("the same with kas_fill_kernel<3,4> for every scenario (KAS_PLAN_FULL_FILL: no slim kernel in front)",
     "!kas_fill_slim_kernel", ["--plan-flags", "16", "--in-flight", "12", "--cells", "32"]),
    ("the same kernels on 16-bit cells in HBM (kas_plan_create16)",
     "[16-bit cells]", ["--in-flight", "12", "--cells", "16"]),
    ("index rows (KAS_PLAN_INDEX_ROWS: the fill's first scan leaves node indices where the mid rows go, the second streams those)",
     "index rows]", ["--plan-flags", "128", "--in-flight", "12", "--cells", "32"]),
    ("first fit beside the order kernel in one workgroup (KAS_PLAN_P4_WITH_ORDER: the order wavefront follows first fit's progress through LDS words)",
     "kas_p4_order_kernel<3>[first fit beside kas_order_relax_kernel<3>[tiles of 64 rows", ["--plan-flags", str(0xC00000), "--in-flight", "12", "--cells", "32"]),
    ("... on 16-bit cells, double tiles (what small batches and host calls take)",
     "kas_p4_order_kernel<3>[first fit beside kas_order_relax_kernel<3>[tiles of 128 rows", ["--plan-flags", str(0xC00000 | 0x40000), "--in-flight", "12", "--cells", "16"]),
    ("first fit inside the fill workgroup (KAS_PLAN_FILL_WITH_P4: four wavefronts hand windows over through LDS, no kas_p4_kernel)",
     "!kas_p4_kernel", ["--plan-flags", "8388608", "--in-flight", "12", "--cells", "32"]),
    ("16-bit cells, first fit inside the fill workgroup + double tiles (what small batches and host calls take)",
     "[16-bit cells]", ["--plan-flags", str(8388608 | 262144), "--in-flight", "12", "--cells", "16"]),
    ("16-bit cells, round form of the order kernel (what a ticket-form request takes there)",
     "kas_order_round_kernel<3>", ["--plan-flags", "65536", "--in-flight", "12", "--scenarios", "200", "--cells", "16"]),
    ("relaxation form over double tiles (KAS_PLAN_RELAX_TILES(2): what batches of fewer than 512 scenarios take)",
     "kas_order_relax_kernel<3>[tiles of 128 rows", ["--plan-flags", "262144", "--in-flight", "12", "--cells", "32"]),
    ("packed ticket form, 2 scenarios per wavefront (KAS_PLAN_TICKET_ORDER)",
     "kas_order_ticket_kernel<3,2,true>", ["--plan-flags", "65536", "--in-flight", "12", "--cells", "32"]),
    ("one scenario per solver wavefront (G = 1)", "kas_order_ticket_kernel<3,1,true>", ["--groups", "1", "--in-flight", "12", "--cells", "32"]),
    ("4 x uint16 counter rows", "kas_order_ticket_kernel<3,2,false>", ["--plan-flags", "4", "--in-flight", "12", "--cells", "32"]),
    ("histogram for the whole topic + chunk-count pass", "kas_fill_kernel<3,4>[quota]", ["--plan-flags", "8", "--in-flight", "12", "--cells", "32"]),
    ("lists 5 wide: wide ticket form (five wavefronts, class lists, joint solve)", "kas_order_wide_kernel<5>",
     ["--scenarios", "96", "--partitions", "40000", "--brokers", "600", "--racks", "40", "--rf", "5",
      "--actions", "add_k,mixed", "--in-flight", "6"]),
    ("lists 4 wide: wide ticket form", "kas_order_wide_kernel<4>",
     ["--scenarios", "96", "--partitions", "40000", "--brokers", "600", "--racks", "40", "--rf", "4",
      "--actions", "add_k,mixed,remove_k", "--in-flight", "6"]),
    ("lists 5 wide, 1,150 rows per broker: wide ticket form with its count fields checked at the end", "count fields checked",
     ["--scenarios", "32", "--partitions", "46000", "--brokers", "200", "--racks", "20", "--rf", "5",
      "--actions", "add_k,mixed", "--in-flight", "4"]),
    ("spread fill (row scans over one-wavefront workgroups, kas_spread_p4_kernel) + wide ticket form", "kas_spread_",
     ["--scenarios", "16", "--partitions", "140000", "--brokers", "800", "--racks", "40", "--rf", "5",
      "--actions", "add_k,mixed", "--in-flight", "3"]),
    ("spread fill, lists 3 wide + relaxation form", "kas_spread_",
     ["--scenarios", "24", "--partitions", "140000", "--brokers", "1000", "--racks", "20", "--rf", "3",
      "--actions", "add_k,mixed", "--in-flight", "4", "--cells", "32"]),
    ("spread fill, lists 3 wide + ticket form", "kas_order_ticket_kernel<3,",
     ["--scenarios", "24", "--partitions", "140000", "--brokers", "1000", "--racks", "20", "--rf", "3",
      "--actions", "add_k,mixed", "--in-flight", "4", "--plan-flags", "65536", "--cells", "32"]),
]


def run_case(argv, rounds=None, min_solves=None):
    """One shape: (solves, wrong records, plan description, seconds).  rounds x slots solves, or enough rounds
    for min_solves."""
    args = bench.parse_args(list(argv))
    args.same_batch = True
    args.steps = max(args.steps, args.in_flight)
    mix = tuple(args.actions.split(",")) if args.actions else G.BENCH_ACTIONS
    t_start = time.perf_counter()
    run = bench.HipRun(args, 0, 1, 0, 0, args.scenarios, mix)
    S, P, RF = run.S, args.partitions, args.rf
    n_slots = len(run.slots)
    if rounds is None:
        rounds = -(-int(min_solves) // n_slots)
    run.solve(run.slots[0]); run.synchronize()
    ref_sr = run.slots[0]["sr"].clone()
    run.synchronize()                      # (the copy runs on torch's stream, the next solves on the slots' own)
    # the reference itself against the oracle (records of every scenario)
    sl0 = run.slots[0]
    sub = node_set_batch(run.check_ids(sl0), sl0["racks"], P, RF, RF, cur=run.check_cur(0))   # (16-bit cells: the index form)
    want = oracle_solve(sub, threads=0)
    got = ref_sr.cpu().numpy().view(abi.SCENARIO_RESULT_DTYPE)
    for f in ("status", "fail_partition", "moved_replicas", "moved_partitions", "digest"):
        assert (got[f] == want.scenario_results[f][:S]).all(), f"the reference solve differs from the oracle in {f}"
    ref_rec = ref_sr.view(S, 32)
    counts = [torch.zeros(S, dtype=torch.int32, device=run.dev) for _ in run.slots]
    for r in range(rounds):
        for i, sl in enumerate(run.slots):
            run.solve(sl)
            with run.stream_ctx(sl):
                counts[i] += (sl["sr"].view(S, 32) != ref_rec).any(dim=1).to(torch.int32)
    run.synchronize()
    tot = torch.stack(counts).sum(dim=0).cpu().numpy()
    bad = np.nonzero(tot)[0]
    n = rounds * n_slots
    describe = run.describe()
    detail = f" (scenarios {bad[:10].tolist()}, actions {[run.actions[s] for s in bad[:10]]})" if len(bad) else ""
    run.close()
    del run, counts
    torch.cuda.empty_cache()
    return n, int(tot.sum()), describe, time.perf_counter() - t_start, S, n_slots, detail


def main(argv):
    if argv and argv[0] == "--suite":
        min_solves = int(argv[1]) if len(argv) > 1 else 1000
        failed = 0
        for name, must, flags in SUITE:
            n, wrong, describe, secs, S, n_slots, detail = run_case(flags, min_solves=min_solves)
            launches = (must[1:] not in describe) if must.startswith("!") else (must in describe)
            okk = wrong == 0 and launches
            failed += 0 if okk else 1
            print(f"[{'ok' if okk else 'FAILED'}] {name}: {n} solves of {S} scenarios, {n_slots} in flight: {wrong} scenario "
                  f"records differ from the reference{detail} ({secs:.1f} s)\n     plan: {describe}", flush=True)
            if not launches:
                print(f"     the plan does not launch {must!r}: this case no longer tests what it is named for", flush=True)
        print(f"suite: {len(SUITE) - failed} of {len(SUITE)} kernel families clean")
        return 1 if failed else 0
    rounds = int(argv[0]) if argv and argv[0].isdigit() else 100
    n, wrong, describe, secs, S, n_slots, detail = run_case(argv[1:] if argv and argv[0].isdigit() else argv, rounds=rounds)
    print(f"{n} solves of {S} scenarios, {n_slots} in flight: {wrong} scenario records differ from the reference{detail}")
    print("plan:", describe)
    return 1 if wrong else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
