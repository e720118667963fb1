#!/usr/bin/env python
"""bench.py — assignment scenarios/sec on MI355X (BASELINE.json metric).

One "step" = one solve of one batch of synthetic cluster scenarios by the HIP path, with every
bulk table already resident in HBM (12 batches in flight on 12 streams by default, each with its own
plan, scratch and outputs; every slot's plan is run once during set-up).  At N=1 the workload is BASELINE.json configs[2] — the
configuration the metric is quoted on: a batch of 1k independent scenarios of 100k partitions x
1k brokers x 20 racks, RF 3, each with its own current assignment G(seed+s) and its own broker-set
perturbation drawn from {remove 1, remove k<=5, add k<=50, remove k<=5 + add j<=50} (SURVEY.md
8d; 'replace 1' is swapped for the mixed action in the headline because the reference itself
throws on most such scenarios, see generator.BENCH_ACTIONS — the literal SURVEY mix is measured
too and reported as config.literal_c3_mix).

Multi-GPU: `python bench.py --gpus N` launches its own N ranks (one process per GPU, re-exec under
torch.distributed.run on 127.0.0.1) unless it already runs under a launcher (WORLD_SIZE set), and
aborts unless world size == N and N devices are visible.  --scaling weak: every rank solves
--scenarios scenarios per step; --scaling strong: --scenarios is the total, cut into contiguous
ranges (sharding.shard_range).  Each step ends with the ONE data-path collective of the design:
an RCCL all-gather of the 32-byte per-scenario result records; its time alone is reported too.

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline     — algorithmic HBM bytes per launch against the 8 TB/s HBM3E peak, at the whole-job
                 rate (bytes / ms_per_step) and per kernel with ONE batch on the GPU (HIP events
                 on the launch stream); kernel names come from the plan (kas_plan_describe)
  cpu_baseline — B1 the CPU oracle (C restatement of the reference Java; no JVM exists here) and
                 B2 the flat-array CPU solver, each on one core (bounded sample) and scenario-
                 parallel on every host core inside one C call
and list-compares EVERY scenario of rank 0's batch with the oracle before reporting.

--stub is the harness self-test (tests/test_bench_harness.py): gloo on CPU tensors and a solve
that only writes synthetic records, so that the launch / shard / gather / report control flow runs
where there is no GPU.  It measures nothing and says so in its line.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

# The HIP runtime multiplexes streams onto 4 hardware queues by default; steps in flight on more
# streams than that would serialise in pairs.  Must be set before the runtime is loaded.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
# multi-process GPU work on this pool needs dmabuf IPC (RCCL otherwise fails in hipIpcGetMemHandle)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
C3_SHAPE = (1000, 100000, 1000, 20, 3)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--scenarios", type=int, default=1000,
                    help="scenarios per GPU per step (--scaling weak) or in total (--scaling strong)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--partitions", type=int, default=100000)
    ap.add_argument("--brokers", type=int, default=1000)
    ap.add_argument("--racks", type=int, default=20)
    ap.add_argument("--rf", type=int, default=3)
    ap.add_argument("--seed", type=int, default=2026)
    ap.add_argument("--actions", default="", help="comma list overriding the per-scenario action mix "
                    "(remove1,remove_k,add_k,mixed,replace1,add50); default generator.BENCH_ACTIONS")
    ap.add_argument("--check", type=int, default=-1,
                    help="scenarios of rank 0's batch list-compared against the oracle (-1 = all)")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="single-core CPU-baseline sample budget (each)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the one-batch-alone and literal-mix legs")
    ap.add_argument("--in-flight", type=int, default=12,
                    help="batches in flight: steps are issued round-robin on this many HIP streams, "
                         "each with its own plan scratch and output tables")
    ap.add_argument("--stats", default="", help="write the per-phase device counters (JSON) here")
    ap.add_argument("--waves", type=int, default=int(os.environ.get("KAS_BENCH_WAVES", "0")),
                    help="wavefronts per scenario workgroup (0 = the plan's choice)")
    ap.add_argument("--groups", type=int, default=int(os.environ.get("KAS_BENCH_GROUPS", "0")),
                    help="scenarios per wavefront of the ticket-form order kernel (0 = the plan's choice)")
    ap.add_argument("--plan-flags", type=int, default=0, help="KAS_PLAN_* switches (testing)")
    ap.add_argument("--stub", action="store_true",
                    help="harness self-test on CPU (gloo, synthetic records): NOT a measurement")
    return ap.parse_args(argv)


# -------------------------------------------------------------------------------------------------
# launching: one process per GPU
# -------------------------------------------------------------------------------------------------
def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run."""
    if not args.stub:
        import torch
        n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if n_dev < args.gpus:
            print(f"bench.py: --gpus {args.gpus} requested but {n_dev} HIP device(s) visible; refusing to "
                  f"run a mislabelled measurement", file=sys.stderr, flush=True)
            return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


# -------------------------------------------------------------------------------------------------
# the two runs behind one interface: HIP (the product) and the CPU stub (harness self-test)
# -------------------------------------------------------------------------------------------------
class HipRun:
    """Inputs in HBM, one plan + output tables + stream per in-flight slot."""

    backend = "nccl"

    def __init__(self, args, rank, world, local_rank, lo, hi, action_mix):
        import torch
        from kafka_assigner_amd import generator as G, native
        from kafka_assigner_amd.flatten import node_set_batch
        assert torch.cuda.is_available(), "bench.py needs a GPU: the product has no CPU path"
        assert torch.cuda.device_count() > local_rank, \
            f"rank {rank}: local rank {local_rank} has no device ({torch.cuda.device_count()} visible)"
        self.torch = torch
        self.args = args
        self.world = world
        self.dev = torch.device("cuda", local_rank)
        torch.cuda.set_device(self.dev)
        self.device_name = torch.cuda.get_device_properties(self.dev).gcnArchName
        self.device_index = local_rank
        S, P, N, R, RF = hi - lo, args.partitions, args.brokers, args.racks, args.rf
        self.S, self.lo = S, lo
        gen = torch.Generator(device=self.dev)
        gen.manual_seed(args.seed + 7919 * rank)
        self.d_cur = G.torch_random_assignment(gen, S, P, N, R, RF, self.dev)          # int32 [S, P, RF]
        self.ctx = native.DeviceContext(local_rank)
        self.n_slots = max(1, min(args.in_flight, args.steps))
        self.slots = []
        self.fb = None
        self.set_actions(action_mix)

    def set_actions(self, action_mix):
        """(Re)build the per-scenario broker sets and the plans for an action mix."""
        torch = self.torch
        from kafka_assigner_amd import generator as G, native
        from kafka_assigner_amd.flatten import node_set_batch
        args = self.args
        for sl in self.slots:
            sl["plan"].close()
        self.ids, self.racks, self.actions = [], [], []
        for s in range(self.S):
            act, bs = G.scenario_action(args.seed, self.lo + s, args.brokers, args.racks, actions=action_mix)
            self.actions.append(act); self.ids.append(bs.node_id); self.racks.append(bs.node_rack)
        self.fb = node_set_batch(self.ids, self.racks, args.partitions, args.rf, args.rf)
        S = self.S
        old = self.slots
        self.slots = []
        for i in range(self.n_slots):
            plan_ = native.Plan(self.ctx, self.fb)
            if args.waves or args.groups or args.plan_flags:
                plan_.set_flags((args.waves << 8) | (args.groups << 12) | args.plan_flags)
            if old:
                sl = old[i]; sl["plan"] = plan_
            else:
                # a dedicated HIP stream per slot, shared by its solver launches and its RCCL
                # all-gather (handle 0, torch's default stream, would select the library's own)
                sl = {"plan": plan_,
                      "out": torch.empty(self.fb.out_len, dtype=torch.int32, device=self.dev),
                      "tr": torch.zeros(S * 16, dtype=torch.uint8, device=self.dev),
                      "sr": torch.zeros(S * 32, dtype=torch.uint8, device=self.dev),
                      "stream": torch.cuda.Stream(self.dev)}
                sl["stream"].wait_stream(torch.cuda.current_stream(self.dev))
            self.slots.append(sl)
        # set-up, not warm-up: every slot's plan runs once so that no slot meets its first launch
        # (scratch first touched, kernels resident) inside the timed region when K is small
        for sl in self.slots:
            self.solve(sl)
        self.synchronize()
        self.step_no = 0

    def solve(self, sl):
        sl["plan"].solve_device(self.d_cur.data_ptr(), sl["out"].data_ptr(), sl["tr"].data_ptr(),
                                sl["sr"].data_ptr(), stream=sl["stream"].cuda_stream)

    def stream_ctx(self, sl):
        return self.torch.cuda.stream(sl["stream"])

    def synchronize(self):
        self.torch.cuda.synchronize(self.dev)

    def records_tensor(self, sl):
        return sl["sr"]

    def new_gather_buffer(self, total):
        return self.torch.zeros(total * 32, dtype=self.torch.uint8, device=self.dev)

    def reset_timers(self):
        for sl in self.slots:
            sl["plan"].kernel_time_us()

    def phase_times(self, slots=None):
        f_us, o_us, n_tot = 0.0, 0.0, 0
        for sl in (slots or self.slots):
            f, o, n = sl["plan"].phase_times_us()
            f_us += f * n; o_us += o * n; n_tot += n
        return (f_us / n_tot, o_us / n_tot, n_tot) if n_tot else (0.0, 0.0, 0)

    def describe(self):
        return self.slots[0]["plan"].describe()

    def algorithmic_bytes(self):
        return self.slots[0]["plan"].algorithmic_bytes

    def host_cur(self, idx):
        return self.torch.stack([self.d_cur[s] for s in idx]).cpu().numpy() if len(idx) != self.S \
            else self.d_cur.cpu().numpy()

    def close(self):
        for sl in self.slots:
            sl["plan"].close()


class StubRun:
    """Harness self-test: CPU tensors over gloo; a 'solve' writes records that depend only on the
    global scenario index, so every rank can check what the all-gather hands back."""

    backend = "gloo"

    def __init__(self, args, rank, world, local_rank, lo, hi, action_mix):
        import torch
        self.torch = torch
        self.args, self.world = args, world
        self.dev = torch.device("cpu")
        self.device_name, self.device_index = "cpu-stub", local_rank
        self.S, self.lo = hi - lo, lo
        self.n_slots = max(1, min(args.in_flight, args.steps))
        self.slots = [{"sr": torch.zeros(self.S * 32, dtype=torch.uint8)} for _ in range(self.n_slots)]
        self.step_no = 0
        self.actions = ["stub"] * self.S

    @staticmethod
    def expected_records(lo, hi):
        from kafka_assigner_amd import abi
        rec = np.zeros(hi - lo, dtype=abi.SCENARIO_RESULT_DTYPE)
        g = np.arange(lo, hi, dtype=np.int64)
        rec["status"] = 0; rec["fail_topic"] = -1; rec["fail_partition"] = -1
        rec["moved_replicas"] = (g * 7 + 3).astype(np.int32); rec["moved_partitions"] = (g * 5 + 1).astype(np.int32)
        rec["digest"] = (g.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)) ^ np.uint64(0xABCDEF)
        return rec

    def set_actions(self, action_mix):
        self.step_no = 0

    def solve(self, sl):
        rec = self.expected_records(self.lo, self.lo + self.S)
        sl["sr"].copy_(self.torch.from_numpy(rec.view(np.uint8).copy()))

    def stream_ctx(self, sl):
        import contextlib
        return contextlib.nullcontext()

    def synchronize(self):
        pass

    def records_tensor(self, sl):
        return sl["sr"]

    def new_gather_buffer(self, total):
        return self.torch.zeros(total * 32, dtype=self.torch.uint8)

    def reset_timers(self):
        pass

    def phase_times(self, slots=None):
        return 0.0, 0.0, 0

    def describe(self):
        return "stub (no kernels)"

    def algorithmic_bytes(self):
        a = self.args
        return self.S * (8 * a.partitions * a.rf + 8 * a.brokers)

    def close(self):
        pass


# -------------------------------------------------------------------------------------------------
# one rank
# -------------------------------------------------------------------------------------------------
def run_rank(args) -> int:
    import torch
    import torch.distributed as dist
    from kafka_assigner_amd import abi, generator as G, sharding

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: WORLD_SIZE {world} != --gpus {args.gpus}; refusing to run a mislabelled measurement",
              file=sys.stderr, flush=True)
        return 2

    # ---- shard: contiguous ranges of the global scenario index -----------------------------------
    total = args.scenarios * world if args.scaling == "weak" else args.scenarios
    lo, hi = sharding.shard_range(total, rank, world)
    sizes = sharding.shard_sizes(total, world)
    action_mix = tuple(a for a in args.actions.split(",") if a) or G.BENCH_ACTIONS

    Run = StubRun if args.stub else HipRun
    run = Run(args, rank, world, local_rank, lo, hi, action_mix)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.stub:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=run.dev)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: process group has {dist.get_world_size()} ranks, --gpus {args.gpus}")
        # which device does every rank drive?  (two ranks on one GPU would be a mislabelled run)
        mine = torch.tensor([run.device_index], dtype=torch.int64, device=run.dev)
        alld = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(alld, mine)
        rank_devices = [int(t.item()) for t in alld]
        if not args.stub and len(set(rank_devices)) != world:
            raise SystemExit(f"bench.py: ranks share devices {rank_devices}")
    else:
        rank_devices = [run.device_index]

    S = run.S
    for sl in run.slots:
        sl["all"] = run.new_gather_buffer(total) if world > 1 else run.records_tensor(sl)

    def step():
        sl = run.slots[run.step_no % run.n_slots]
        run.step_no += 1
        run.solve(sl)
        if world > 1:                                   # the single data-path collective
            with run.stream_ctx(sl):
                sl["all"] = sharding.gather_records(run.records_tensor(sl), total, out=sl["all"])

    def fence():
        run.synchronize()
        if world > 1:
            dist.barrier()
        run.synchronize()

    def timed(n_steps, n_warm):
        for _ in range(n_warm):
            step()
        fence()
        run.reset_timers()
        run.step_no = 0
        t0 = time.perf_counter()
        for _ in range(n_steps):
            step()
        fence()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=run.dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    elapsed = timed(args.steps, args.warmup)
    fill_us, order_us, kern_n = run.phase_times()

    if args.stats and rank == 0 and not args.stub:
        st = run.slots[0]["plan"].stats().astype(np.float64)
        names = ["setup_us", "p2_hist_quota_us", "p2_keep_p3_us", "p4_us", "p4_windows", "p4_steps",
                 "p5_rounds_or_queue_steps", "p2_ranked_tiles_wave0", "order_us", "solver_iterations", "solver_queue_rounds",
                 "solver_blocked", "stager_iterations", "stager_idle", "solver_queue_rows", "solver_rows_in_hand"]
        scale = [0.01, 0.01, 0.01, 0.01, 1, 1, 1, 1, 0.01, 1, 1, 1, 1, 1, 1, 1]
        summary = {n: {"mean": float(st[:, i].mean() * scale[i]), "max": float(st[:, i].max() * scale[i]),
                       "min": float(st[:, i].min() * scale[i])} for i, n in enumerate(names)}
        summary["fill_kernel_avg_us"] = fill_us
        summary["order_kernel_avg_us"] = order_us
        json.dump(summary, open(args.stats, "w"), indent=1)

    # ---- results of this rank (headline mix), and what the collective handed back ----------------
    sl0 = run.slots[0]
    sr = run.records_tensor(sl0).cpu().numpy().view(abi.SCENARIO_RESULT_DTYPE).copy()
    ok = int((sr["status"] == abi.KAS_OK).sum())
    gathered_ok = True
    if world > 1:
        all_sr = sl0["all"].cpu().numpy().view(abi.SCENARIO_RESULT_DTYPE)
        assert all_sr.shape[0] == total
        assert (all_sr[lo:hi] == sr).all(), "all-gather returned a different record for my own shard"
        if args.stub:
            gathered_ok = bool((all_sr == StubRun.expected_records(0, total)).all())
            assert gathered_ok, "all-gather returned wrong records for another rank's shard"

    # ---- the all-gather alone (reported separately) -----------------------------------
    allgather_us = None
    if world > 1:
        fence()
        reps = 20
        t0 = time.perf_counter()
        for i in range(reps):
            sl = run.slots[i % run.n_slots]
            with run.stream_ctx(sl):
                sharding.gather_records(run.records_tensor(sl), total, out=sl["all"])
        fence()
        t = torch.tensor([(time.perf_counter() - t0) / reps * 1e6], dtype=torch.float64, device=run.dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        allgather_us = float(t.item())

    # ---- one batch alone on the GPU: per-kernel durations free of other launches -----------------
    alone = None
    if not args.stub and not args.no_extras:
        keep, keep_n = run.slots, run.n_slots
        run.slots, run.n_slots = keep[:1], 1
        n_alone = max(3, min(10, args.steps))
        el1 = timed(n_alone, 1)
        f1, o1, n1 = run.phase_times()
        run.slots, run.n_slots = keep, keep_n
        alone = {"steps": n_alone, "ms_per_step": 1e3 * el1 / n_alone, "fill_kernel_us": f1, "order_kernel_us": o1,
                 "launches_timed": n1}

    # ---- the literal SURVEY 8(d) C3 action mix (with 'replace 1'), same cur tables ---------------
    literal = None
    # (the condition must not depend on the rank: every rank takes part in the timed collectives)
    if (not args.stub and not args.no_extras and not args.actions and args.scaling == "weak" and
            (args.scenarios, args.partitions, args.brokers, args.racks, args.rf) == C3_SHAPE):
        run.set_actions(G.ACTIONS)
        n_lit = max(run.n_slots, min(args.steps, 24))
        el2 = timed(n_lit, run.n_slots)
        sr2 = run.records_tensor(run.slots[0]).cpu().numpy().view(abi.SCENARIO_RESULT_DTYPE)
        literal = {"actions": list(G.ACTIONS), "steps": n_lit, "value": world * S * n_lit / el2,
                   "unit": "scenarios/s", "ms_per_step": 1e3 * el2 / n_lit,
                   "failed_scenarios_rank0": int((sr2["status"] != abi.KAS_OK).sum()),
                   "note": "a failed scenario (the reference's KAS:183-184 stranding at zero slack) skips P5, "
                           "which is why this mix is not the headline"}
        run.set_actions(action_mix)                      # back to the headline mix for the parity leg
        run.solve(run.slots[0]); run.synchronize()

    out_line = None
    if rank == 0:
        P, N, R, RF = args.partitions, args.brokers, args.racks, args.rf
        checked, checked_lists, cpu = 0, 0, None
        if not args.stub:
            checked, checked_lists, cpu = parity_and_cpu_baselines(args, run, sr, world)
        alg_bytes = run.algorithmic_bytes()
        value = total * args.steps / elapsed
        ms_per_step = 1e3 * elapsed / args.steps
        achieved = alg_bytes / (ms_per_step * 1e-3) / 1e9          # per GPU: rank 0's bytes per step time
        per_launch = alg_bytes / ((fill_us + order_us) * 1e-6) / 1e9 if (fill_us + order_us) > 0 else 0.0
        roof = {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
            "regime": f"whole-job rate per GPU: algorithmic bytes of one solve / ms_per_step, "
                      f"{run.n_slots} batches in flight",
            "kernel": run.describe(),
            "algorithmic_bytes_per_launch": alg_bytes,
            "in_flight_launch": {"fill_kernel_us": fill_us, "order_kernel_us": order_us, "launches_timed": kern_n,
                                 "achieved": per_launch, "frac": per_launch / HBM_PEAK_GBPS,
                                 "note": "HIP-event durations per launch while the batches in flight share the GPU "
                                         "(a launch then lasts longer than ms_per_step)"},
        }
        if alone:
            a_us = alone["fill_kernel_us"] + alone["order_kernel_us"]
            roof["one_batch_alone"] = dict(alone, achieved=alg_bytes / (a_us * 1e-6) / 1e9 if a_us > 0 else 0.0,
                                           frac=(alg_bytes / (a_us * 1e-6) / 1e9 / HBM_PEAK_GBPS) if a_us > 0 else 0.0,
                                           dominant_kernel="order" if alone["order_kernel_us"] >= alone["fill_kernel_us"] else "fill",
                                           dominant_kernel_achieved=alg_bytes / (max(alone["order_kernel_us"], alone["fill_kernel_us"]) * 1e-6) / 1e9
                                           if a_us > 0 else 0.0)
        shape_c3 = (S, P, N, R, RF) == C3_SHAPE
        out_line = {
            "metric": "assignment scenarios/sec at 100k partitions x 1k brokers RF=3",
            "value": value, "unit": "scenarios/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "int32",
            "data": "synthetic",
            "config": {
                "workload": f"{'BASELINE.json configs[2]' if shape_c3 else 'custom shape'}: "
                            f"batch of {S} independent scenarios per GPU, "
                            f"{P} partitions x {N} brokers x {R} racks, RF {RF}; per-scenario G(seed+s) "
                            f"current assignment + action in {{{', '.join(action_mix)}}} (remove <= 5, add <= 50)",
                "scenarios_per_gpu": sizes, "scenarios_total": total,
                "partitions": P, "brokers": N, "racks": R, "rf": RF,
                "world_size": world, "rank_devices": rank_devices, "device": run.device_name,
                "ok_scenarios_rank0": ok, "failed_scenarios_rank0": S - ok,
                "failed_note": "a failed scenario is the reference's own KAS:183-184 stranding, "
                               "reproduced bit-exactly (status + partition id)",
                "parity_checked_scenarios": checked, "parity_list_compared_scenarios": checked_lists,
                "collective": "all_gather of 32-byte result records per step" if world > 1 else "none (1 GPU)",
                "allgather_alone_us": allgather_us,
                "batches_in_flight": run.n_slots, "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                "setup_solves_per_slot": 0 if args.stub else 1,
                "literal_c3_mix": literal,
            },
            "roofline": roof,
            "cpu_baseline": cpu,
        }
        if args.stub:
            out_line["stub"] = True
            out_line["metric"] = "STUB harness self-test - not a measurement"
            out_line["config"]["gathered_records_ok"] = gathered_ok
        # PMC-measured HBM traffic of the same command, when a committed profile provides it
        try:
            prof = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            if prof.get("scenarios") == S and prof.get("partitions") == P and not args.stub:
                out_line["roofline"]["traffic"] = prof["hbm_bytes_per_launch"]
                out_line["roofline"]["traffic_source"] = prof.get("source")
        except Exception:
            pass
        print(json.dumps(out_line), flush=True)
    run.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def parity_and_cpu_baselines(args, run, sr, world):
    """Rank 0: compare the whole batch with the oracle (records of every scenario, and the lists of
    every scenario unless --check bounds them) and time the CPU baselines on the same scenarios."""
    from kafka_assigner_amd import abi
    from kafka_assigner_amd.flatten import node_set_batch
    from oracle_lib import cpu_fast_solve, host_threads, oracle_solve
    S, P, RF = run.S, args.partitions, args.rf
    n_check = S if args.check < 0 else max(0, min(args.check, S))
    if n_check == 0:
        return 0, 0, None
    pick = list(range(n_check))
    bad = np.nonzero(sr["status"] != abi.KAS_OK)[0]
    if len(bad) and int(bad[0]) not in pick:            # keep a failing scenario in a bounded sample
        pick[-1] = int(bad[0])
    h_cur = run.host_cur(pick)
    sub = node_set_batch([run.ids[s] for s in pick], [run.racks[s] for s in pick], P, RF, RF, cur=h_cur)
    cores = host_threads()

    def median_wall(solve, reps):
        """The whole batch on every host core inside one C call: median wall time of `reps` solves."""
        walls, res = [], None
        for _ in range(reps):
            t1 = time.perf_counter()
            res = solve(sub, threads=0)
            walls.append(time.perf_counter() - t1)
        return sorted(walls)[len(walls) // 2], res

    timed_cpu = not args.no_cpu and world == 1
    b1_all, want = median_wall(oracle_solve, 3 if timed_cpu else 1)     # B1 (and the checker's answers)
    b1_threads = want.threads_used
    ow = RF
    got_out = run.slots[0]["out"].cpu().numpy() if n_check == S else None
    for i, s in enumerate(pick):
        for f in ("status", "fail_topic", "fail_partition", "moved_replicas", "moved_partitions", "digest"):
            assert sr[f][s] == want.scenario_results[f][i], f"scenario {s}: {f} differs from the oracle"
        rows = got_out[s * P * ow:(s + 1) * P * ow] if got_out is not None else \
            run.slots[0]["out"][s * P * ow:(s + 1) * P * ow].cpu().numpy()
        assert (rows == want.out[i * P * ow:(i + 1) * P * ow]).all(), f"scenario {s}: lists differ from the oracle"
    checked = checked_lists = len(pick)

    cpu = None
    if timed_cpu:
        b2_all, fast = median_wall(cpu_fast_solve, 3)   # B2, scenario-parallel on every host core
        for f in ("status", "fail_partition", "moved_replicas", "moved_partitions", "digest"):
            assert (fast.scenario_results[f][:len(pick)] == want.scenario_results[f][:len(pick)]).all(), \
                f"cpu_fast {f} differs from the oracle"
        assert (fast.out[:len(pick) * P * ow] == want.out[:len(pick) * P * ow]).all(), "cpu_fast lists differ"

        def one_core(solve):
            done, spent, m = 0, 0.0, 4
            t_c0 = time.perf_counter()
            while True:
                idx = [(done + i) % len(pick) for i in range(m)]
                part = node_set_batch([run.ids[pick[j]] for j in idx], [run.racks[pick[j]] for j in idx], P, RF, RF,
                                      cur=h_cur[idx])
                t2 = time.perf_counter()
                solve(part, threads=1)
                spent += time.perf_counter() - t2
                done += m
                if time.perf_counter() - t_c0 > args.cpu_seconds or done >= len(pick):
                    return done, spent
                m = min(64, m * 2)

        d1, s1 = one_core(oracle_solve)
        d2, s2 = one_core(cpu_fast_solve)
        cpu = {
            "value": d1 / s1, "unit": "scenarios/s", "cores": 1, "kind": "port",
            "sample": f"B1 oracle/kas_oracle.c (C restatement of the reference Java, rescans order[0..] per orphan "
                      f"like KAS:175), 1 thread, {d1} scenarios of the same batch, {s1:.1f} s solve time; "
                      f"no JVM in this image",
            "host_hardware_threads": cores,
            "oracle_all_cores": {"value": len(pick) / b1_all, "unit": "scenarios/s", "cores": b1_threads,
                                 "sample": f"{len(pick)} scenarios (the whole batch), pthreads inside one C call, "
                                           f"{b1_all:.2f} s wall (median of 3)"},
            "cpu_fast": {"value": d2 / s2, "unit": "scenarios/s", "cores": 1, "kind": "port",
                         "sample": f"B2 oracle/kas_cpu_fast.c (flat arrays, full-node skipping, same results: "
                                   f"diffed against B1 on the whole batch), 1 thread, {d2} scenarios, {s2:.1f} s"},
            "cpu_fast_all_cores": {"value": len(pick) / b2_all, "unit": "scenarios/s", "cores": fast.threads_used,
                                   "sample": f"{len(pick)} scenarios (the whole batch), pthreads inside one C call, "
                                             f"{b2_all:.2f} s wall (median of 3)"},
        }
    return checked, checked_lists, cpu


def main() -> int:
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return spawn_ranks(args)
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
