#!/usr/bin/env python
"""bench.py — assignment scenarios/sec on MI355X (BASELINE.json metric).

One "step" = one solve of one batch of synthetic cluster scenarios by the HIP path, with every
bulk table already resident in HBM.  Eight batches are in flight on eight streams by default (round 3:
6 to 10 measure the same within noise, 12 and more are ~4 % slower — the order kernel's workgroups then
oversubscribe the CUs), and they are eight DIFFERENT batches: every slot has its own current-assignment
tables, its own broker-set draws, its own plan, scratch and outputs (8 x 2.4 GB of tables).  At N=1 the workload is
BASELINE.json configs[2] — the configuration the metric is quoted on: a batch of 1k independent
scenarios of 100k partitions x 1k brokers x 20 racks, RF 3, each with its own current assignment
G(seed+s) and its own broker-set perturbation drawn from {remove 1, remove k<=5, add k<=50,
remove k<=5 + add j<=50} ('replace 1' is swapped for the mixed action in the headline
because the reference itself throws on most such scenarios, see generator.BENCH_ACTIONS — the literal
SURVEY mix is measured too and reported as config.literal_c3_mix).

The timed region (exactly --steps steps between barriers) is run --repeats times inside one
invocation; `value` and `ms_per_step` are the MEDIAN repeat's, the spread is in `repeats`.

Multi-GPU: `python bench.py --gpus N` launches its own N ranks (one process per GPU, re-exec under
torch.distributed.run on 127.0.0.1) unless it already runs under a launcher (WORLD_SIZE set), and
aborts unless world size == N and N devices are visible.  --scaling weak: every rank solves
--scenarios scenarios per step; --scaling strong: --scenarios is the total, cut into contiguous
ranges (sharding.shard_range).  Each step ends with the ONE data-path collective of the design:
an RCCL all-gather of the 32-byte per-scenario result records; its time alone is reported too.

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline      — algorithmic HBM bytes per launch against the 8 TB/s HBM3E peak, at the whole-job
                  rate (bytes / ms_per_step) and per kernel with ONE batch on the GPU (HIP events
                  on the launch stream); kernel names come from the plan (kas_plan_describe);
                  `traffic` only from a committed PMC profile of the same kernels AND sources
  cpu_baseline  — B1 the CPU oracle (C restatement of the reference Java; no JVM exists here) and
                  B2 the flat-array CPU solver, each on one core (bounded sample) and scenario-
                  parallel on every host core inside one C call
  end_to_end    — through the host-buffer boundary JNI / the CLI call (kas_solve_host[_select]):
                  the what-if form (1000 broker sets over ONE snapshot, records of all, rows of one)
                  and the plain form (every scenario its own tables) with its PCIe rate
  other_configs — BASELINE.json configs[1] and configs[4] on this GPU with a CPU figure beside each
and checks, before reporting, the records (status, failing ids, movement, digest of every cell) of
EVERY slot's last timed solve against the CPU solvers and every list of slot 0 against the oracle.

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


SLOT_SEED_STRIDE = 1000003   # slot k draws its tables and broker sets from seed + k * this


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--repeats", type=int, default=5,
                    help="times the timed region (exactly --steps steps) is run; value = the median repeat")
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
                    help="scenarios of slot 0 list-compared against the oracle (-1 = all); the records of every "
                         "slot are compared either way unless this is 0")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="single-core CPU-baseline sample budget (each)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the one-batch-alone, literal-mix, end-to-end and other-config legs")
    ap.add_argument("--in-flight", type=int, default=12,
                    help="batches in flight: steps are issued round-robin on this many HIP streams, "
                         "each slot with its own tables, broker sets, plan scratch and outputs (round 5: 12 — with every "
                         "slot's stream on a hardware queue of its own 10-12 slots read 2-5 %% above 8 in the driver's "
                         "20-step region, whose slots start in phase; rounds 3-4: 8, when two pairs of slots shared a queue)")
    ap.add_argument("--same-batch", action="store_true",
                    help="every slot solves slot 0's tables and broker sets (the consistency stress of "
                         "scripts/stress_inflight.py; not a measurement mode)")
    ap.add_argument("--stats", default="", help="write the per-phase device counters (JSON) here")
    ap.add_argument("--waves", type=int, default=int(os.environ.get("KAS_BENCH_WAVES", "0")),
                    help="wavefronts per scenario workgroup (0 = the plan's choice)")
    ap.add_argument("--groups", type=int, default=int(os.environ.get("KAS_BENCH_GROUPS", "0")),
                    help="scenarios per wavefront of the ticket-form order kernel (0 = the plan's choice)")
    ap.add_argument("--plan-flags", type=int, default=0, help="KAS_PLAN_* switches (testing)")
    ap.add_argument("--cells", type=int, choices=(16, 32), default=32,
                    help="cells of the tables resident in HBM in the TIMED REGION: 32 = int32 broker ids in, broker ids out "
                         "(kas_plan_create; SURVEY 8(b)/(d)'s contract — the headline), 16 = uint16 node indices "
                         "(kas_plan_create16, ABI v5; lists up to 3 wide, else 32 is taken).  The other layout is measured "
                         "beside it on the same slots and reported as value_cells16 / value_int32_cells")
    ap.add_argument("--rccl-at-1", action="store_true",
                    help="run the data-path collective at ONE rank too (an RCCL communicator of one rank on this GPU: the library "
                         "loads, the communicator initialises on the device, the all-gather runs on the slots' own streams inside "
                         "the timed steps) - what a one-GPU box can exercise of the N > 1 path; never the default")
    ap.add_argument("--stub", action="store_true",
                    help="harness self-test on CPU (gloo, synthetic records): NOT a measurement")
    ap.add_argument("--config", type=int, choices=(2, 3, 4), default=0,
                    help="BASELINE.json configs[N] as a preset of the flags above (an explicit flag still wins): 2 = the headline "
                         "(1000 scenarios x 100k x 1k x 20 racks, RF 3, per GPU); 3 = 64k scenarios of that shape with brokers "
                         "1000-1049 added, cut over the GPUs (--scaling strong; 8000 per GPU at --gpus 8); 4 = 1M partitions x 5k "
                         "brokers x 40 racks, RF 5, remove every 50th broker + add 200, rack map on / off alternating, 8 variants "
                         "per GPU as replicas (--scaling weak)")
    args = ap.parse_args(argv)
    apply_config_preset(args, ap, argv if argv is not None else sys.argv[1:])
    return args


# BASELINE.json configs[2..4] as flag presets (DESIGN.md section 8 lists the command lines)
CONFIG_PRESETS = {
    2: dict(scenarios=1000, partitions=100000, brokers=1000, racks=20, rf=3, scaling="weak"),
    3: dict(scenarios=64000, partitions=100000, brokers=1000, racks=20, rf=3, scaling="strong", actions="add50", in_flight=2),
    4: dict(scenarios=8, partitions=1000000, brokers=5000, racks=40, rf=5, scaling="weak", actions="c5,c5_norack", in_flight=1,
            check=2),
}


def apply_config_preset(args, ap, argv):
    """--config N: the preset's values for every flag the command line does not name itself."""
    if not args.config:
        return
    named = {a.split("=")[0] for a in argv if a.startswith("--")}
    for key, val in CONFIG_PRESETS[args.config].items():
        if "--" + key.replace("_", "-") not in named:
            setattr(args, key, val)
    if args.config == 3 and "--in-flight" not in named and args.gpus < 2:
        args.in_flight = 1                          # (64k scenarios on ONE GPU: 154 GB of tables per batch in flight)


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


def sources_sha16() -> str:
    """Hash of the kernel sources (csrc/*.hip, *.h + the ABI header): ties a committed profile to the code it
    was taken from (the built .so is not tracked)."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "kafka-assigner_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode()); h.update(open(os.path.join(csrc, f), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "kas_abi.h"), "rb").read())
    return h.hexdigest()[:16]


def cells16_table(torch, cur_ids, ids, n_brokers, dev):
    """int32 broker ids [S, P, RF] -> int16 [S, P * RF] of node indices: every replica as the position of its broker in the
    scenario's ascending broker list `ids[s]` (0xFFFF: the broker left the set), by one table lookup per scenario on `dev`
    (what flatten.to_cells16 does on the host: tests/test_bench_harness.py holds the two to each other)."""
    S = len(ids)
    width = max(int(max(int(x.max()) for x in ids)) + 2, n_brokers + 2, int(cur_ids.max().item()) + 2)
    lut = np.full((S, width), 0xFFFF, dtype=np.uint16)
    for s, x in enumerate(ids):
        lut[s, x] = np.arange(len(x), dtype=np.uint16)
    d_lut = torch.from_numpy(lut.view(np.int16)).to(dev)
    out = torch.empty((S, cur_ids.shape[1] * cur_ids.shape[2]), dtype=torch.int16, device=dev)
    step = 64
    for lo in range(0, S, step):
        hi = min(S, lo + step)
        idx = cur_ids[lo:hi].reshape(hi - lo, -1).to(torch.int64)
        out[lo:hi] = torch.gather(d_lut[lo:hi], 1, idx)
    return out


# -------------------------------------------------------------------------------------------------
# the two runs behind one interface: HIP (the product) and the CPU stub (harness self-test)
# -------------------------------------------------------------------------------------------------
class HipRun:
    """Inputs in HBM; one set of tables + broker sets + plan + output tables + stream per in-flight slot."""

    backend = "nccl"

    def __init__(self, args, rank, world, local_rank, lo, hi, action_mix):
        import torch
        from kafka_assigner_amd import generator as G, native
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
        self.ctx = native.DeviceContext(local_rank)
        self.cells16 = getattr(args, "cells", 32) == 16 and RF <= 3
        self.n_slots = max(1, min(args.in_flight, args.steps))
        self.distinct = 1 if args.same_batch else self.n_slots
        # every slot its own current assignments (slot 0's are the ones rounds 1 and 2 measured)
        self.d_cur = []
        for k in range(self.distinct):
            gen = torch.Generator(device=self.dev)
            gen.manual_seed(args.seed + 7919 * rank + SLOT_SEED_STRIDE * k)
            self.d_cur.append(G.torch_random_assignment(gen, S, P, N, R, RF, self.dev))      # int32 [S, P, RF]
        self.slots = []
        self.set_actions(action_mix)

    def slot_seed(self, k):
        return self.args.seed + SLOT_SEED_STRIDE * (k if self.distinct > 1 else 0)

    def set_actions(self, action_mix):
        """(Re)build the per-scenario broker sets and the plans for an action mix."""
        torch = self.torch
        from kafka_assigner_amd import generator as G, native
        from kafka_assigner_amd.flatten import node_set_batch
        args = self.args
        for sl in self.slots:
            sl["plan"].close()
        S = self.S
        old = self.slots
        self.slots = []
        for i in range(self.n_slots):
            ids, racks, actions = [], [], []
            for s in range(S):
                act, bs = G.scenario_action(self.slot_seed(i), self.lo + s, args.brokers, args.racks, actions=action_mix)
                actions.append(act); ids.append(bs.node_id); racks.append(bs.node_rack)
            fb = node_set_batch(ids, racks, args.partitions, args.rf, args.rf)
            plan_ = native.Plan(self.ctx, fb, cells16=self.cells16)
            if args.waves or args.groups or args.plan_flags:
                plan_.set_flags((args.waves << 8) | (args.groups << 12) | args.plan_flags)
            if old:
                sl = old[i]
            else:
                # a dedicated HIP stream per slot, shared by its solver launches and its RCCL
                # all-gather (handle 0, torch's default stream, would select the library's own)
                sl = {"out": torch.empty(fb.out_len, dtype=torch.int16 if self.cells16 else torch.int32, device=self.dev),
                      "tr": torch.zeros(S * 16, dtype=torch.uint8, device=self.dev),
                      "sr": torch.zeros(S * 32, dtype=torch.uint8, device=self.dev),
                      "stream": torch.cuda.Stream(self.dev)}
                sl["stream"].wait_stream(torch.cuda.current_stream(self.dev))
            cur_ids = self.d_cur[i if self.distinct > 1 else 0]
            sl.update(plan=plan_, fb=fb, ids=ids, racks=racks, actions=actions, cur_ids=cur_ids,
                      cur=self.cells_of(cur_ids, ids) if self.cells16 else cur_ids)
            self.slots.append(sl)
        self.actions = self.slots[0]["actions"]
        # set-up, not warm-up: every slot's plan runs once so that no slot meets its first launch
        # (scratch first touched, kernels resident) inside the timed region when K is small
        self.synchronize()                  # (the tables' conversion above ran on torch's stream: not beside the first solves)
        for sl in self.slots:
            self.solve(sl)
        self.synchronize()
        self.step_no = 0

    def cells_of(self, cur_ids, ids):
        """The resident form of a slot's tables with 16-bit cells (cells16_table; set-up, not timed)."""
        return cells16_table(self.torch, cur_ids, ids, self.args.brokers, self.dev)

    def check_ids(self, sl):
        """broker ids of the slot's scenarios as the CHECKERS see them: with 16-bit cells node i has id i"""
        return [np.arange(len(x), dtype=np.int32) for x in sl["ids"]] if self.cells16 else sl["ids"]

    def check_cur(self, slot, idx=None):
        """the slot's cur tables as the checkers see them (int32 [n, P, RF]): broker ids, or node indices with -1 for 0xFFFF"""
        if not self.cells16:
            return self.host_cur(slot, idx)
        c = self.slots[slot]["cur"]
        if idx is not None and len(idx) != self.S:
            c = self.torch.stack([c[s] for s in idx])
        h = c.cpu().numpy().view(np.uint16).astype(np.int32)
        h[h == 0xFFFF] = -1
        return h.reshape(h.shape[0], self.args.partitions, self.args.rf)

    def check_out(self, sl, lo=None, hi=None):
        """out cells [lo, hi) as int32 (-1 = pad): broker ids, or node indices"""
        o = sl["out"] if lo is None else sl["out"][lo:hi]
        h = o.cpu().numpy()
        if not self.cells16:
            return h
        h = h.view(np.uint16).astype(np.int32)
        h[h == 0xFFFF] = -1
        return h

    def other_cells_rate(self, n_steps, n_regions=3):
        """The same slots (same scenarios, same broker sets, same streams) with the OTHER cell layout resident in HBM, for the
        line's second figure: with int32 broker ids timed as the headline, uint16 node-index cells (kas_plan_create16 /
        kas_solve_device16; the id -> index conversion of the tables is set-up here, outside the region: what a caller that
        keeps its tables in that form has done already); with --cells 16, int32 broker ids (kas_plan_create /
        kas_solve_device).  n_regions regions of n_steps steps round-robin, median.  Builds and drops its own plans and
        tables; returns (dict, records of slot 0's last solve)."""
        torch = self.torch
        from kafka_assigner_amd import abi, native
        args = self.args
        to16 = not self.cells16
        if to16 and args.rf > 3:
            return None, None
        extra = []
        for sl in self.slots:
            plan_ = native.Plan(self.ctx, sl["fb"], cells16=to16)
            if args.waves or args.groups or args.plan_flags:
                plan_.set_flags((args.waves << 8) | (args.groups << 12) | args.plan_flags)
            cur = self.cells_of(sl["cur_ids"], sl["ids"]) if to16 else sl["cur_ids"]
            extra.append((plan_, cur, torch.empty(sl["fb"].out_len, dtype=torch.int16 if to16 else torch.int32, device=self.dev)))
        self.synchronize()

        def go(k):
            sl, (plan_, cur, out) = self.slots[k % self.n_slots], extra[k % self.n_slots]
            plan_.solve_device(cur.data_ptr(), out.data_ptr(), sl["tr"].data_ptr(), sl["sr"].data_ptr(),
                               stream=sl["stream"].cuda_stream)
        for k in range(self.n_slots):
            go(k)
        self.synchronize()
        walls = []
        for _ in range(max(1, n_regions)):
            t0 = time.perf_counter()
            for k in range(n_steps):
                go(k)
            self.synchronize()
            walls.append(time.perf_counter() - t0)
        el = sorted(walls)[(len(walls) - 1) // 2]
        rec0 = self.slots[0]["sr"].cpu().numpy().view(abi.SCENARIO_RESULT_DTYPE).copy()
        what = extra[0][0].describe()
        alg = extra[0][0].algorithmic_bytes
        for plan_, _, _ in extra:
            plan_.close()
        del extra
        # (the slots' records were overwritten by these solves: one solve each puts the headline layout's back)
        for sl in self.slots:
            self.solve(sl)
        self.synchronize()
        ms = 1e3 * el / n_steps
        res = {"value": self.S * n_steps / el, "unit": "scenarios/s", "ms_per_step": ms, "steps": n_steps,
               "regions": [self.S * n_steps / w for w in walls], "kernel": what,
               "dtype": "uint16 node index" if to16 else "int32",
               "algorithmic_bytes_per_launch": alg,
               "achieved": alg / (ms * 1e-3) / 1e9, "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS}
        return res, rec0

    def solve(self, sl):
        sl["plan"].solve_device(sl["cur"].data_ptr(), sl["out"].data_ptr(), sl["tr"].data_ptr(),
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

    def host_cur(self, slot, idx=None):
        """the slot's tables as int32 broker ids (what the generator made: the host-boundary legs start from these)"""
        cur = self.slots[slot]["cur_ids"]
        if idx is None or len(idx) == self.S:
            return cur.cpu().numpy()
        return self.torch.stack([cur[s] for s in idx]).cpu().numpy()

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
        self.distinct = self.n_slots
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
    coll = world > 1 or bool(getattr(args, "rccl_at_1", False))     # the data-path collective runs (and its process group exists)
    if coll:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1 and "MASTER_PORT" not in os.environ:           # (--rccl-at-1 without a launcher)
            import socket
            with socket.socket() as so:
                so.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(so.getsockname()[1])
        if args.stub:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=run.dev)
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
        sl["all"] = run.new_gather_buffer(total) if coll else run.records_tensor(sl)
    if coll:
        # Pre-flight of the one data-path collective, before anything is timed: every slot's all-gather once, on the slot's
        # OWN stream (where the timed steps put it: a non-default HIP stream with a hardware queue of its own, which is where
        # an RCCL / IPC problem would first show), synchronised, the records of my own shard checked.  A failure ends the
        # run here with the collective's error text and no JSON line.  Environment the run needs:
        # HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC between the ranks' devices), GPU_MAX_HW_QUEUES >= slots + 2.
        try:
            for i, sl in enumerate(run.slots):
                with run.stream_ctx(sl):
                    sl["all"] = sharding.gather_records(run.records_tensor(sl), total, out=sl["all"])
            run.synchronize()
            dist.barrier()
            mine = run.records_tensor(run.slots[0])
            got = run.slots[0]["all"][lo * sharding.RECORD_BYTES:hi * sharding.RECORD_BYTES]
            if not bool((got == mine).all()):
                raise RuntimeError("the all-gather returned other bytes than this rank's own records for its shard")
        except Exception as e:
            print(f"bench.py: rank {rank}: pre-flight all-gather of the result records failed: {e!r}\n"
                  f"  (HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')}, "
                  f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')}, world {world}, device {run.device_index})",
                  file=sys.stderr, flush=True)
            raise SystemExit(3)

    def step():
        sl = run.slots[run.step_no % run.n_slots]
        run.step_no += 1
        run.solve(sl)
        if coll:                                   # the single data-path collective
            with run.stream_ctx(sl):
                sl["all"] = sharding.gather_records(run.records_tensor(sl), total, out=sl["all"])

    def fence():
        run.synchronize()
        if coll:
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
        if coll:
            t = torch.tensor([el], dtype=torch.float64, device=run.dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    # the timed region, --repeats times: W warm-up steps before the first, then exactly K steps each,
    # bracketed by barrier + synchronize on both sides, max over ranks; the median repeat is the figure
    n_rep = max(1, args.repeats)
    reps, rep_kernel_times = [], []
    for r in range(n_rep):
        reps.append(timed(args.steps, args.warmup if r == 0 else 0))
        rep_kernel_times.append(run.phase_times())
    order = sorted(range(n_rep), key=lambda r: reps[r])
    med = order[(n_rep - 1) // 2]
    elapsed = reps[med]
    fill_us, order_us, kern_n = rep_kernel_times[med]
    # the same region twice as long (three repeats): the ramp at both ends of a 20-step region is ~3 % of it, so a
    # kernel gain of that size would not show in `value`; reported beside it, never instead of it
    long_steps = 2 * args.steps
    long_reps = [] if args.stub or args.no_extras else [timed(long_steps, 0) for _ in range(3)]
    run.phase_times()
    # the same slots with the OTHER cell layout resident in HBM (HipRun.other_cells_rate), reported beside the headline
    other_cells, other_rec0 = None, None
    if not args.stub and not args.no_extras and world == 1:
        other_cells, other_rec0 = run.other_cells_rate(args.steps)

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

    # ---- results of this rank (headline mix): the records EVERY slot's last timed solve left ------
    slot_records = [run.records_tensor(sl).cpu().numpy().view(abi.SCENARIO_RESULT_DTYPE).copy() for sl in run.slots]
    sl0 = run.slots[0]
    sr = slot_records[0]
    ok = int((sr["status"] == abi.KAS_OK).sum())
    gathered_ok = True
    if coll:
        all_sr = sl0["all"].cpu().numpy().view(abi.SCENARIO_RESULT_DTYPE)
        assert all_sr.shape[0] == total
        assert (all_sr[lo:hi] == sr).all(), "all-gather returned a different record for my own shard"
        if args.stub:
            gathered_ok = bool((all_sr == StubRun.expected_records(0, total)).all())
            assert gathered_ok, "all-gather returned wrong records for another rank's shard"

    # ---- the all-gather alone (reported separately) -----------------------------------
    allgather_us = None
    if coll:
        fence()
        n_ag = 20
        t0 = time.perf_counter()
        for i in range(n_ag):
            sl = run.slots[i % run.n_slots]
            with run.stream_ctx(sl):
                sharding.gather_records(run.records_tensor(sl), total, out=sl["all"])
        fence()
        t = torch.tensor([(time.perf_counter() - t0) / n_ag * 1e6], dtype=torch.float64, device=run.dev)
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
        # ... and as a caller that KNOWS its batch has the GPU to itself would ask for it (what kas_solve_host's plans and batches of
        # fewer than 512 scenarios take by themselves): first fit as a wavefront of the order kernel's workgroup, double tiles
        try:
            from kafka_assigner_amd import abi as _abi
            base_flags = (args.waves << 8) | (args.groups << 12) | args.plan_flags
            run.slots, run.n_slots = keep[:1], 1
            keep[0]["plan"].set_flags(base_flags | _abi.KAS_PLAN_P4_WITH_ORDER | _abi.KAS_PLAN_RELAX_TILES_128)
            what_l = keep[0]["plan"].describe()
            el1l = timed(n_alone, 1)
            f1l, o1l, _ = run.phase_times()
            alone["with_latency_flags"] = {"ms_per_step": 1e3 * el1l / n_alone, "fill_kernel_us": f1l, "order_kernel_us": o1l,
                                           "plan_flags": "KAS_PLAN_P4_WITH_ORDER | KAS_PLAN_RELAX_TILES(2)", "kernel": what_l}
            # ... and over quad tiles (256 rows a step, round 6: where the plan has dword mid rows; elsewhere this is the leg above again)
            keep[0]["plan"].set_flags(base_flags | _abi.KAS_PLAN_P4_WITH_ORDER | _abi.KAS_PLAN_RELAX_TILES_64 | _abi.KAS_PLAN_RELAX_TILES_128)
            what_q = keep[0]["plan"].describe()
            el1q = timed(n_alone, 1)
            f1q, o1q, _ = run.phase_times()
            alone["with_latency_flags_quad_tiles"] = {"ms_per_step": 1e3 * el1q / n_alone, "fill_kernel_us": f1q, "order_kernel_us": o1q,
                                                      "plan_flags": "KAS_PLAN_P4_WITH_ORDER | KAS_PLAN_RELAX_TILES(3)", "kernel": what_q}
            keep[0]["plan"].set_flags(base_flags)
            run.solve(keep[0]); run.synchronize()                  # (slot 0's records and rows: the headline plan's again)
        except Exception as e:                                     # (a leg of its own: never costs the headline line)
            alone["with_latency_flags"] = {"error": repr(e)}
        finally:
            run.slots, run.n_slots = keep, keep_n

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
        # (set_actions solves every slot once: slot 0's lists are the headline mix's again)

    out_line = None
    if rank == 0:
        P, N, R, RF = args.partitions, args.brokers, args.racks, args.rf
        parity, cpu = {"records": 0, "lists": 0, "slots": 0}, None
        if not args.stub:
            parity, cpu = parity_and_cpu_baselines(args, run, slot_records, world)
        alg_bytes = run.algorithmic_bytes()
        value = total * args.steps / elapsed
        ms_per_step = 1e3 * elapsed / args.steps
        achieved = alg_bytes / (ms_per_step * 1e-3) / 1e9          # per GPU: rank 0's bytes per step time
        per_launch = alg_bytes / ((fill_us + order_us) * 1e-6) / 1e9 if (fill_us + order_us) > 0 else 0.0
        describe = run.describe()
        roof = {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
            "regime": f"whole-job rate per GPU: algorithmic bytes of one solve / ms_per_step, "
                      f"{run.n_slots} batches in flight",
            "kernel": describe,
            "kernel_sources_sha16": sources_sha16(),
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
        rates = [total * args.steps / e for e in reps]
        out_line = {
            "metric": "assignment scenarios/sec at 100k partitions x 1k brokers RF=3",
            "value": value, "unit": "scenarios/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "uint16 node index" if getattr(run, "cells16", False) else "int32",
            "data": "synthetic",
            "repeats": {"n": n_rep, "value_is": "median repeat", "values": rates, "min": min(rates), "max": max(rates),
                        "spread_pct": 100.0 * (max(rates) - min(rates)) / value if value > 0 else None,
                        "ms_per_step_each": [1e3 * e / args.steps for e in reps],
                        "note": f"each repeat = exactly {args.steps} steps between barrier + synchronize; "
                                f"{args.warmup} warm-up steps before the first"},
            "config": {
                "workload": f"{('BASELINE.json configs[%d]' % args.config) if args.config else ('BASELINE.json configs[2]' if shape_c3 else 'custom shape')}: "
                            f"batch of {S} independent scenarios per GPU, "
                            f"{P} partitions x {N} brokers x {R} racks, RF {RF}; per-scenario G(seed+s) "
                            f"current assignment + action in {{{', '.join(action_mix)}}} (remove <= 5, add <= 50)",
                "preset": args.config or None,
                "scenarios_per_gpu": sizes, "scenarios_total": total,
                "partitions": P, "brokers": N, "racks": R, "rf": RF,
                "world_size": world, "rank_devices": rank_devices, "device": run.device_name,
                "ok_scenarios_rank0": ok, "failed_scenarios_rank0": S - ok,
                "failed_note": "a failed scenario is the reference's own KAS:183-184 stranding, "
                               "reproduced bit-exactly (status + partition id)",
                "parity_checked_scenarios": parity["records"], "parity_list_compared_scenarios": parity["lists"],
                "parity_checked_slots": parity["slots"],
                "parity_note": "records (status, failing topic / partition, movement counts, digest of every emitted "
                               "cell) of every slot's last timed solve against the CPU solvers; every list of slot 0 "
                               "against the oracle",
                "collective": ("all_gather of 32-byte result records per step" + ("" if world > 1 else " (an RCCL communicator of ONE rank: --rccl-at-1)")) if coll else "none (1 GPU)",
                "allgather_alone_us": allgather_us,
                "batches_in_flight": run.n_slots, "distinct_batches_in_flight": run.distinct,
                "cells": ("uint16 node indices resident in HBM (kas_plan_create16 / kas_solve_device16, ABI v5): a replica is the "
                          "position of its broker in the scenario's ascending broker list; parity: the CPU solvers on the index "
                          "form of every batch" if getattr(run, "cells16", False) else
                          "int32 broker ids resident in HBM in, broker ids out (SURVEY 8(b)/(d): the id -> node lookup of KAS:118-119 "
                          "and the ids of KAS:231-236 are inside the timed solve)"),
                "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                "setup_solves_per_slot": 0 if args.stub else 1,
                "literal_c3_mix": literal,
            },
            "roofline": roof,
            "cpu_baseline": cpu,
        }
        # the other cell layout on the same slots: flat scalars (a record parser that drops nested objects keeps them) and the
        # whole leg under config; every frac is the layout's OWN bytes over its OWN time
        out_line["frac"] = roof["frac"]
        out_line["algorithmic_bytes_per_launch"] = alg_bytes
        if other_cells is not None:
            tag = "int32_cells" if getattr(run, "cells16", False) else "cells16"
            for f in ("status", "fail_topic", "fail_partition", "moved_replicas", "moved_partitions"):
                assert (other_rec0[f] == slot_records[0][f]).all(), f"{tag}: {f} of slot 0 differs between the two cell layouts"
            other_cells["records_checked"] = (f"status, failing topic / partition and movement counts of slot 0's {S} scenarios equal "
                                              "the headline layout's (whose records and lists are checked against the CPU solvers)")
            if tag == "cells16":
                other_cells["note"] = ("the same scenarios with the tables resident as uint16 node indices (ABI v5): the id -> index "
                                       "map of cur and the index -> id map of out are the CALLER's here (set-up, outside the region), "
                                       "which is why this is not the headline; frac = this layout's own bytes (2 P (cw + ow) + 4 N per "
                                       "scenario) over its own time")
            else:
                other_cells["note"] = "the same scenarios with int32 broker ids in and out (SURVEY 8(d)'s contract)"
            out_line["config"][tag] = other_cells
            out_line["value_" + tag] = other_cells["value"]
            out_line["ms_per_step_" + tag] = other_cells["ms_per_step"]
            out_line["frac_" + tag] = other_cells["frac"]
            out_line["dtype_" + tag] = other_cells["dtype"]
        if long_reps:
            lr = sorted(long_reps)[1]
            out_line["value_long_region"] = {"steps": long_steps, "value": total * long_steps / lr,
                                             "values": [total * long_steps / e for e in long_reps],
                                             "note": "the timed region twice as long (median of three): less of it is the ramp "
                                                     "at its two ends; `value` stays the figure of the driver's --steps"}
        if args.stub:
            out_line["stub"] = True
            out_line["metric"] = "STUB harness self-test - not a measurement"
        if coll or args.stub:                                 # (this rank's own shard came back byte for byte: asserted above; the stub checks every shard)
            out_line["config"]["gathered_records_ok"] = gathered_ok
        # PMC-measured HBM traffic: only from a committed profile of the SAME kernels (the plan's own
        # description) built from the SAME sources — anything else would be a stale number
        try:
            prof = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            if (not args.stub and prof.get("kernel") == describe and
                    prof.get("kernel_sources_sha16") == roof["kernel_sources_sha16"]):
                roof["traffic"] = prof["hbm_bytes_per_launch"]
                roof["traffic_source"] = prof.get("source")
            elif not args.stub:
                roof["traffic_note"] = ("profiles/pmc_traffic.json was taken from other kernels or sources "
                                        f"({prof.get('kernel_sources_sha16')}): not quoted")
        except Exception:
            pass
        # the measured traffic against what HBM DELIVERS to a plain streaming kernel with the same mix of reads and writes
        # (profiles/hbm_streaming.json: experiments/micro/mall_reads.hip on this GPU) — `frac` stays algorithmic bytes over the
        # nominal 8 TB/s; this says how much of the distance is traffic and nominal-versus-delivered, how much idle time
        if roof.get("traffic"):
            try:
                hs = json.load(open(os.path.join(ROOT, "profiles", "hbm_streaming.json")))
                mix = float(hs["two_streams_in_one_out_GBps"])
                t_stream_ms = roof["traffic"] / (mix * 1e9) * 1e3
                roof["delivered"] = {"streaming_GBps_at_this_mix": mix, "streaming_GBps_reads": hs["reads_16_byte_loads_GBps"],
                                     "traffic_at_streaming_rate_ms": t_stream_ms,
                                     "frac_of_streaming_rate_for_own_traffic": t_stream_ms / ms_per_step,
                                     "traffic_over_algorithmic": roof["traffic"] / alg_bytes,
                                     "source": hs["source"]}
            except Exception:
                pass
        # which pipe is nearest its ceiling: the per-pipe utilisation of the committed SQ-counter passes of this very
        # command line (scripts/pipe_table.py; eight batches in flight, whole job), HBM from the traffic above at this
        # run's rate.  Counters cannot be collected inside a timed run: the table is a committed measurement.
        if not args.stub and (S, P, N, R, RF) == C3_SHAPE:
            try:
                import csv
                pipes = {}
                side = json.load(open(os.path.join(ROOT, "profiles", "pipe_utilisation.json")))
                if (side.get("kernel") != describe or side.get("kernel_sources_sha16") != roof["kernel_sources_sha16"] or
                        side.get("batches_in_flight") != run.n_slots):
                    roof["pipes_note"] = ("profiles/pipe_utilisation.json was taken from other kernels or sources "
                                          f"({side.get('kernel_sources_sha16')}): not quoted")
                    raise LookupError("stale pipe table")
                with open(os.path.join(ROOT, side["csv"])) as f:
                    vals = {}
                    for row in csv.DictReader(f):
                        if row["batches_in_flight"] not in ("1", "batches_in_flight") and row["kernel"] == "fill+order":
                            vals[row["quantity"]] = float(row["value"])
                            if row["fraction_of_capacity"]:
                                pipes[row["quantity"]] = float(row["fraction_of_capacity"])
                table = {"valu_pipe_busy": pipes.get("VALU pipe busy, counter"),
                         "salu_issue_busy": pipes.get("SALU issue busy"),
                         "lds_array_busy": pipes.get("LDS array busy"),
                         "lds_array_busy_without_bank_conflicts": pipes.get("LDS array busy without bank-conflict cycles")}
                if roof.get("traffic"):
                    table["hbm_busy"] = roof["traffic"] / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS
                near = max(((k, v) for k, v in table.items() if v is not None and k != "lds_array_busy_without_bank_conflicts"),
                           key=lambda kv: kv[1])
                roof["pipes"] = dict(table, nearest_ceiling={"pipe": near[0], "frac": near[1]},
                                     waves_resident_per_simd=vals.get("waves resident per SIMD (average)"),
                                     source=side["csv"] + f" (rocprofv3 SQ counter passes of bench.py, {run.n_slots} batches in flight, "
                                            "fill + order kernels over the ms_per_step window; units in the file; same kernel "
                                            "sources as this run)")
            except Exception:
                pass
        if not args.stub and not args.no_extras and world == 1:
            try:
                out_line["end_to_end"] = end_to_end_leg(args, run)
            except Exception as e:                       # (a leg of its own: never costs the headline line)
                out_line["end_to_end"] = {"error": repr(e)}
            try:
                out_line["other_configs"] = other_configs_leg(args, run)
            except Exception as e:
                out_line["other_configs"] = {"error": repr(e)}
        print(json.dumps(out_line), flush=True)
    run.close()
    if coll:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def parity_and_cpu_baselines(args, run, slot_records, world):
    """Rank 0.  Every slot: the records its last timed solve left against the flat-array CPU solver
    (B2) on that slot's own tables and broker sets.  Slot 0 in addition: the whole batch with the oracle
    (B1), every list.  The CPU baselines are timed on the same solves."""
    from kafka_assigner_amd import abi
    from kafka_assigner_amd.flatten import node_set_batch
    from oracle_lib import cpu_fast_solve, host_threads, oracle_solve
    S, P, RF = run.S, args.partitions, args.rf
    n_check = S if args.check < 0 else max(0, min(args.check, S))
    if n_check == 0:
        return {"records": 0, "lists": 0, "slots": 0}, None
    sr = slot_records[0]
    pick = list(range(n_check))
    bad = np.nonzero(sr["status"] != abi.KAS_OK)[0]
    if len(bad) and int(bad[0]) not in pick:            # keep a failing scenario in a bounded sample
        pick[-1] = int(bad[0])
    sl0 = run.slots[0]
    # (with 16-bit cells resident in HBM the checkers solve the index form of the batch — node i has id i, cur holds node
    # indices — whose lists and digests are the device's cell for cell: run.check_*)
    h_cur = run.check_cur(0, pick)
    ids0 = run.check_ids(sl0)
    sub = node_set_batch([ids0[s] for s in pick], [sl0["racks"][s] for s in pick], P, RF, RF, cur=h_cur)
    cores = host_threads()
    fields = ("status", "fail_topic", "fail_partition", "moved_replicas", "moved_partitions", "digest")

    def median_wall(solve, batch, n):
        """The whole batch on every host core inside one C call: median wall time of `n` solves into the same
        host buffers, after one untimed solve (the GPU figures are warm figures too: no first touch of fresh
        pages, no NUMA placement of a new buffer, inside a timed call)."""
        from kafka_assigner_amd.flatten import host_tables
        into = host_tables(batch)
        walls, res = [], solve(batch, threads=0, into=into)
        for _ in range(n):
            t1 = time.perf_counter()
            res = solve(batch, threads=0, into=into)
            walls.append(time.perf_counter() - t1)
        return (sorted(walls)[len(walls) // 2] if walls else None), res

    timed_cpu = not args.no_cpu and world == 1
    b1_all, want = median_wall(oracle_solve, sub, 3 if timed_cpu else 0)      # B1 (and the checker's answers)
    b1_threads = want.threads_used
    ow = RF
    got_out = run.check_out(sl0) if n_check == S else None
    for i, s in enumerate(pick):
        for f in fields:
            assert sr[f][s] == want.scenario_results[f][i], f"slot 0 scenario {s}: {f} differs from the oracle"
        rows = got_out[s * P * ow:(s + 1) * P * ow] if got_out is not None else \
            run.check_out(sl0, s * P * ow, (s + 1) * P * ow)
        assert (rows == want.out[i * P * ow:(i + 1) * P * ow]).all(), f"slot 0 scenario {s}: lists differ from the oracle"
    parity = {"records": len(pick), "lists": len(pick), "slots": 1}

    # every other slot: its own tables, its own broker sets, the records of its last timed solve
    fast0 = None
    if n_check == S:
        for k in range(run.n_slots):
            sl = run.slots[k]
            if k > 0 and run.distinct == 1:
                assert (slot_records[k] == sr).all(), f"slot {k} (same inputs as slot 0) left different records"
                parity["records"] += S; parity["slots"] += 1
                continue
            batch = sub if k == 0 else node_set_batch(run.check_ids(sl), sl["racks"], P, RF, RF, cur=run.check_cur(k))
            fast = cpu_fast_solve(batch, threads=0)
            if k == 0:
                fast0 = fast
                for f in fields:
                    assert (fast.scenario_results[f][:S] == want.scenario_results[f][:S]).all(), f"cpu_fast {f} differs from the oracle"
                assert (fast.out[:S * P * ow] == want.out[:S * P * ow]).all(), "cpu_fast lists differ from the oracle"
            else:
                for f in fields:
                    diff = np.nonzero(slot_records[k][f] != fast.scenario_results[f][:S])[0]
                    assert diff.size == 0, f"slot {k} scenario {int(diff[0])}: {f} differs from the CPU solver"
                parity["records"] += S; parity["slots"] += 1
            del batch

    cpu = None
    if timed_cpu:
        b2_all, fast0 = median_wall(cpu_fast_solve, sub, 5)

        def one_core(solve):
            done, spent, m = 0, 0.0, 4
            t_c0 = time.perf_counter()
            while True:
                idx = [(done + i) % len(pick) for i in range(m)]
                part = node_set_batch([ids0[pick[j]] for j in idx], [sl0["racks"][pick[j]] for j in idx], P, RF, RF,
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
        r1, r2 = d1 / s1, d2 / s2
        from oracle_lib import delivered_parallelism
        par = delivered_parallelism()
        cpu = {
            "value": r1, "unit": "scenarios/s", "cores": 1, "kind": "port",
            "sample": f"B1 oracle/kas_oracle.c (C restatement of the reference Java, rescans order[0..] per orphan "
                      f"like KAS:175), 1 thread, {d1} scenarios of the same batch, {s1:.1f} s solve time; "
                      f"no JVM in this image",
            "host_hardware_threads": cores,
            "host_parallelism": dict(par, note="the same register-only loop on 1 and on every hardware thread: the cores this "
                                               "process is really given (a CPU quota on the container bounds every all-core "
                                               "figure below; scaling_efficiency_delivered is against that, not against the "
                                               "thread count)"),
            "oracle_all_cores": {"value": len(pick) / b1_all, "unit": "scenarios/s", "cores": b1_threads,
                                 "scaling_efficiency": (len(pick) / b1_all) / (r1 * b1_threads),
                                 "scaling_efficiency_delivered": (len(pick) / b1_all) / (r1 * par["cores_delivered"]) if par.get("cores_delivered") else None,
                                 "sample": f"{len(pick)} scenarios (the whole batch), pthreads inside one C call with "
                                           f"per-thread scratch arenas, {b1_all:.2f} s wall (median of 3, same host buffers)"},
            "cpu_fast": {"value": r2, "unit": "scenarios/s", "cores": 1, "kind": "port",
                         "sample": f"B2 oracle/kas_cpu_fast.c (flat arrays, full-node skipping, same results: "
                                   f"diffed against B1 on the whole batch), 1 thread, {d2} scenarios, {s2:.1f} s"},
            "cpu_fast_all_cores": {"value": len(pick) / b2_all, "unit": "scenarios/s", "cores": fast0.threads_used,
                                   "scaling_efficiency": (len(pick) / b2_all) / (r2 * fast0.threads_used),
                                   "scaling_efficiency_delivered": (len(pick) / b2_all) / (r2 * par["cores_delivered"]) if par.get("cores_delivered") else None,
                                   "sample": f"{len(pick)} scenarios per solve, pthreads inside one C call with per-thread "
                                             f"scratch arenas, {b2_all:.2f} s wall (median of 5, same host buffers)"},
        }
    return parity, cpu


# -------------------------------------------------------------------------------------------------
# end to end through the host-buffer boundary (what JNI, the CLI and the C++ mirror call)
# -------------------------------------------------------------------------------------------------
def end_to_end_leg(args, run):
    """SURVEY 8(d)'s second definition of the metric: tables in HOST memory in, results in HOST memory out.
    (a) what-if: S broker sets over ONE snapshot (one shared cur table), kas_solve_host_select — records of
    every variant, the rows of one; same batch again and fresh broker sets on every call.  (b) plain: every
    scenario its own tables, kas_solve_host, pageable and pinned caller buffers, with the PCIe rate."""
    import ctypes as C
    from kafka_assigner_amd import abi, generator as G, native
    from kafka_assigner_amd.flatten import batch_desc, host_tables, node_set_batch
    from oracle_lib import cpu_fast_solve, oracle_solve
    S, P, N, R, RF = run.S, args.partitions, args.brokers, args.racks, args.rf
    L = native.load()
    ctx = native.DeviceContext(run.device_index)
    fields = ("status", "fail_topic", "fail_partition", "moved_replicas", "moved_partitions", "digest")
    snapshot = run.host_cur(0, [0])[0]                                  # ONE current assignment [P, RF]

    def variants(seed):
        ids, racks = [], []
        for s in range(S):
            _, bs = G.scenario_action(seed, s, N, R, actions=G.BENCH_ACTIONS)
            ids.append(bs.node_id); racks.append(bs.node_rack)
        return node_set_batch(ids, racks, P, RF, RF, shared_cur=True, cur=snapshot)

    sets = [variants(args.seed + 17 + i) for i in range(3)]
    select = np.asarray([S // 2], dtype=np.int32)
    cells = P * RF
    calls = []
    for fb in sets:                                                     # buffers and descriptors outside the timed calls
        t, ho = host_tables(fb, out_len=native.selected_out_len(fb, select))
        calls.append((fb, batch_desc(fb), t, ho))

    def call(i):
        fb, bd, t, ho = calls[i]
        native._check(L.kas_solve_host_select(ctx._h, C.byref(bd), C.byref(t), select.ctypes.data_as(C.POINTER(C.c_int32)), 1))

    call(0)                                                             # first call: allocations, plan
    n = 6
    t0 = time.perf_counter()
    for _ in range(n):
        call(0)
    same = (time.perf_counter() - t0) / n
    call(1); call(2)
    t0 = time.perf_counter()
    for i in range(n):
        call(1 + i % 2)                                                 # other broker sets than the call before
    fresh = (time.perf_counter() - t0) / n
    hs = ctx.host_stats()
    # parity: every record of every variant set against the CPU solver, the selected variant's rows against the oracle
    for fb, _, _, ho in calls:
        want = cpu_fast_solve(fb, threads=0)
        for f in fields:
            assert (ho.scenario_results[f][:S] == want.scenario_results[f][:S]).all(), f"end_to_end what-if: {f} differs"
        one = node_set_batch([fb.node_id[int(fb.scen['node_off'][select[0]]):int(fb.scen['node_off'][select[0]]) + int(fb.scen['n_nodes'][select[0]])]],
                             [fb.node_rack[int(fb.scen['node_off'][select[0]]):int(fb.scen['node_off'][select[0]]) + int(fb.scen['n_nodes'][select[0]])]],
                             P, RF, RF, cur=snapshot)
        assert (ho.out[:cells] == oracle_solve(one).out[:cells]).all(), "end_to_end what-if: selected rows differ from the oracle"
    up = 4 * (snapshot.size + 2 * int(sets[0].node_id.size)) + 96 * S
    res = {
        "value": S / fresh, "unit": "scenarios/s",
        "what": f"kas_solve_host_select (the entry JNI's layout-3 payload and whatif.py call): {S} broker-set variants "
                f"of ONE {P} x {N} x RF {RF} snapshot per call, host buffers in and out, blocking; records of every "
                f"variant + the rows of one come back; value = fresh broker sets on every call",
        "what_if_fresh_broker_sets_each_call": {"value": S / fresh, "ms_per_call": 1e3 * fresh},
        "what_if_same_batch_again": {"value": S / same, "ms_per_call": 1e3 * same},
        "bytes_up_per_call": up, "bytes_down_per_call": 48 * S + 4 * cells,
        "parity": f"{3 * S} records against the CPU solver, the selected variant's {P} lists against the oracle",
        "host_path_counters": {"calls": hs[0], "plans_found_byte_for_byte": hs[1], "device_allocations": hs[2]},
    }
    # (b) every scenario its own tables: PCIe-bound
    m = min(S, 240)
    sl0 = run.slots[0]
    fbp = node_set_batch(sl0["ids"][:m], sl0["racks"][:m], P, RF, RF, cur=run.host_cur(0, list(range(m))))
    bdp = batch_desc(fbp)
    t_pg, ho_pg = host_tables(fbp)
    pin_cur, pin_out = native.PinnedArray(fbp.cur.size), native.PinnedArray(fbp.out_len)
    pin_cur.array[:] = fbp.cur
    pin_out.array[:] = 0                                                # (touch the pages: the first copies into fresh pinned memory fault them in)
    t_pin, ho_pin = host_tables(fbp)
    t_pin.cur = pin_cur.array.ctypes.data; t_pin.out = pin_out.array.ctypes.data
    plain = {}
    for name, t in (("pageable", t_pg), ("pinned", t_pin)):
        for _ in range(2):                                              # plans, device buffers, the caller's pages
            native._check(L.kas_solve_host(ctx._h, C.byref(bdp), C.byref(t)))
        t0 = time.perf_counter()
        for _ in range(4):
            native._check(L.kas_solve_host(ctx._h, C.byref(bdp), C.byref(t)))
        dt = (time.perf_counter() - t0) / 4
        plain[name] = {"value": m / dt, "unit": "scenarios/s", "ms_per_call": 1e3 * dt,
                       "pcie_gb_per_s": 4 * (fbp.cur.size + fbp.out_len) / dt / 1e9}
    want = cpu_fast_solve(fbp, threads=0)
    assert (ho_pg.out[:fbp.out_len] == want.out[:fbp.out_len]).all() and (pin_out.array == want.out[:fbp.out_len]).all(), \
        "end_to_end plain: lists differ from the CPU solver"
    for f in fields:
        assert (ho_pin.scenario_results[f][:m] == want.scenario_results[f][:m]).all()
    # the same call with 16-bit cells (kas_solve_host16, ABI v5): node indices, half the bytes over the link
    from kafka_assigner_amd.flatten import cells16_to_ids, host_tables16, to_cells16
    c16 = to_cells16(fbp)
    pin_cur16, pin_out16 = native.PinnedArray(c16.size, np.uint16), native.PinnedArray(fbp.out_len, np.uint16)
    pin_cur16.array[:] = c16
    pin_out16.array[:] = 0
    t16, ho16 = host_tables16(fbp, pin_cur16.array)
    t16.out = pin_out16.array.ctypes.data
    bdp16 = batch_desc(fbp)
    bdp16.node_id = None
    for _ in range(2):
        native._check(L.kas_solve_host16(ctx._h, C.byref(bdp16), C.byref(t16), None, -1))
    t0 = time.perf_counter()
    for _ in range(4):
        native._check(L.kas_solve_host16(ctx._h, C.byref(bdp16), C.byref(t16), None, -1))
    dt = (time.perf_counter() - t0) / 4
    plain["pinned_16_bit_cells"] = {"value": m / dt, "unit": "scenarios/s", "ms_per_call": 1e3 * dt,
                                    "pcie_gb_per_s": 2 * (c16.size + fbp.out_len) / dt / 1e9,
                                    "what": "kas_solve_host16: cur / out as uint16 node indices, widened / narrowed on the device"}
    assert (cells16_to_ids(fbp, pin_out16.array) == want.out[:fbp.out_len]).all(), "end_to_end plain, 16-bit cells: lists differ from the CPU solver"
    for f in fields:
        if f != "digest":                                               # (the digest covers the cells as emitted: node indices)
            assert (ho16.scenario_results[f][:m] == want.scenario_results[f][:m]).all()
    pin_cur16.close(); pin_out16.close()
    plain["what"] = (f"kas_solve_host: {m} scenarios with their own {P} x {RF} tables per call ({4 * fbp.cur.size / 1e6:.0f} MB up, "
                     f"{4 * fbp.out_len / 1e6:.0f} MB down), cut into scenario ranges whose upload / solve / download overlap; "
                     f"pinned = caller buffers from kas_host_alloc (DMA without staging)")
    res["plain_every_scenario_its_own_tables"] = plain
    pin_cur.close(); pin_out.close()
    # (c) the call the reference's adapter makes (KTA:70-71): ONE topic, its Context handed in and wanted back, host
    # buffers in and out, blocking — BASELINE configs[1]'s topic (10k partitions x 100 brokers x RF 3, decommission 1)
    import dataclasses
    from kafka_assigner_amd.flatten import uniform_batch
    cur1 = G.random_assignment(0, 10000, 100, 10, 3)
    bs1 = G.perturb_brokers(100, 10, remove=[0])
    fb1 = uniform_batch(cur1[None], bs1.node_id[None], bs1.node_rack[None], 3)
    scen1 = fb1.scen.copy(); scen1["ctx_width"] = 3; scen1["ctx_off"] = 0
    ctx_in = np.random.default_rng(1).integers(0, 300, size=int(fb1.scen["n_nodes"][0]) * 3).astype(np.int32)
    fb1 = dataclasses.replace(fb1, scen=scen1, ctx=ctx_in)
    bd1 = batch_desc(fb1)
    t1, ho1 = host_tables(fb1)
    native._check(L.kas_solve_host(ctx._h, C.byref(bd1), C.byref(t1)))
    want1 = cpu_fast_solve(fb1, threads=1)
    assert (ho1.out[:fb1.out_len] == want1.out[:fb1.out_len]).all() and (ho1.ctx == want1.ctx).all(), \
        "end_to_end per-topic call: differs from the CPU solver"
    n1 = 200
    t0 = time.perf_counter()
    for _ in range(n1):
        ho1.ctx[:] = ctx_in                                             # every call starts from the same Context
        native._check(L.kas_solve_host(ctx._h, C.byref(bd1), C.byref(t1)))
    per_call = (time.perf_counter() - t0) / n1
    tc = []
    for _ in range(9):
        t0 = time.perf_counter(); cpu_fast_solve(fb1, threads=1); tc.append(time.perf_counter() - t0)
    to = []
    for _ in range(5):
        t0 = time.perf_counter(); oracle_solve(fb1); to.append(time.perf_counter() - t0)
    c1 = to_cells16(fb1)
    t1h, ho1h = host_tables16(fb1, c1)
    bd1h = batch_desc(fb1)
    bd1h.node_id = None
    native._check(L.kas_solve_host16(ctx._h, C.byref(bd1h), C.byref(t1h), None, -1))
    assert (cells16_to_ids(fb1, ho1h.out)[:fb1.out_len] == want1.out[:fb1.out_len]).all() and (ho1h.ctx == want1.ctx).all(), \
        "end_to_end per-topic call, 16-bit cells: differs from the CPU solver"
    t0 = time.perf_counter()
    for _ in range(n1):
        ho1h.ctx[:] = ctx_in
        native._check(L.kas_solve_host16(ctx._h, C.byref(bd1h), C.byref(t1h), None, -1))
    per_call16 = (time.perf_counter() - t0) / n1
    res["per_topic_call_with_context"] = {
        "gpu_ms_per_call_16_bit_cells": 1e3 * per_call16,
        "what": "kas_solve_host, one 10k x 100 x RF 3 topic with the adapter's Context in and out (120 KB up, 120 KB down), "
                "blocking: the drop-in for one getRackAwareAssignment call",
        "gpu_ms_per_call": 1e3 * per_call,
        "cpu_fast_one_core_ms": 1e3 * sorted(tc)[len(tc) // 2],
        "oracle_one_core_ms": 1e3 * sorted(to)[len(to) // 2],
    }
    ctx.close()
    return res


# -------------------------------------------------------------------------------------------------
# BASELINE.json's other single-GPU configurations, each with a CPU figure measured in the same run
# -------------------------------------------------------------------------------------------------
def other_configs_leg(args, run):
    import torch
    from kafka_assigner_amd import abi, generator as G, native
    from kafka_assigner_amd.flatten import uniform_batch
    from oracle_lib import cpu_fast_solve, host_threads
    dev = run.dev
    ctx = run.ctx
    out = {}

    def gpu_ms(fb, n=20, flags=0):
        plan = native.Plan(ctx, fb)
        if flags:
            plan.set_flags(flags)
        d_ctx0 = torch.from_numpy(fb.ctx).to(dev) if fb.ctx is not None and fb.ctx.size else None
        d_ctx = d_ctx0.clone() if d_ctx0 is not None else None
        d_cur = torch.from_numpy(fb.cur).to(dev)
        d_out = torch.empty(fb.out_len, dtype=torch.int32, device=dev)
        d_tr = torch.zeros(fb.n_topics * 16, dtype=torch.uint8, device=dev)
        d_sr = torch.zeros(fb.n_scenarios * 32, dtype=torch.uint8, device=dev)
        st = torch.cuda.Stream(dev)
        st.wait_stream(torch.cuda.current_stream(dev))

        def go():
            if d_ctx is not None:
                with torch.cuda.stream(st):
                    d_ctx.copy_(d_ctx0, non_blocking=True)               # every solve starts from the same Context (a 1 KB copy)
            plan.solve_device(d_cur.data_ptr(), d_out.data_ptr(), d_tr.data_ptr(), d_sr.data_ptr(),
                              ctx=d_ctx.data_ptr() if d_ctx is not None else 0, stream=st.cuda_stream)
        go(); st.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            go()
        st.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / n
        f_us, o_us, _ = plan.phase_times_us()
        desc = plan.describe()
        sr = d_sr.cpu().numpy().view(abi.SCENARIO_RESULT_DTYPE).copy()
        rows = d_out.cpu().numpy()
        plan.close()
        if d_ctx is not None:
            return ms, f_us, o_us, desc, sr, rows, d_ctx.cpu().numpy()
        return ms, f_us, o_us, desc, sr, rows

    def cpu_ms(fb, threads, n=3):
        walls, res = [], None
        for _ in range(n):
            t0 = time.perf_counter()
            res = cpu_fast_solve(fb, threads=threads)
            walls.append(time.perf_counter() - t0)
        return 1e3 * sorted(walls)[len(walls) // 2], res

    def check(fb, sr, rows, want, what):
        for f in ("status", "fail_partition", "moved_replicas", "moved_partitions", "digest"):
            assert (sr[f] == want.scenario_results[f][:fb.n_scenarios]).all(), f"{what}: {f} differs from the CPU solver"
        assert (rows == want.out[:fb.out_len]).all(), f"{what}: lists differ from the CPU solver"

    # configs[1]: one scenario, 10k partitions x 100 brokers x 10 racks, RF 3, decommission 1 broker
    cur = G.random_assignment(0, 10000, 100, 10, 3)
    bs = G.perturb_brokers(100, 10, remove=[0])
    fb = uniform_batch(cur[None], bs.node_id[None], bs.node_rack[None], 3)
    ms, f_us, o_us, desc, sr, rows = gpu_ms(fb, n=200)
    c1, want = cpu_ms(fb, 1, n=9)
    check(fb, sr, rows, want, "configs[1]")
    # the same call as the reference's adapter makes it (KTA:19-23, 70-71): with the instance's Context handed in and
    # wanted back — counters as a few earlier topics would have left them; relaxation form and, asked for, ticket form
    import dataclasses
    scen_c = fb.scen.copy()
    scen_c["ctx_width"] = 3
    scen_c["ctx_off"] = 0
    ctx0 = np.random.default_rng(1).integers(0, 300, size=int(fb.scen["n_nodes"][0]) * 3).astype(np.int32)
    fbc = dataclasses.replace(fb, scen=scen_c, ctx=ctx0)
    want_c = cpu_fast_solve(fbc, threads=1)
    with_ctx = {}
    for name, flags in (("relaxation_form", 0), ("ticket_form", abi.KAS_PLAN_TICKET_ORDER)):
        ms_c, f_c, o_c, desc_c, sr_c, rows_c, ctx_c = gpu_ms(fbc, n=200, flags=flags)
        assert (rows_c == want_c.out[:fbc.out_len]).all() and (ctx_c == want_c.ctx).all(), "configs[1] with a Context: differs from the CPU solver"
        with_ctx[name] = {"gpu_ms_per_solve": ms_c, "gpu_order_kernel_us": o_c, "kernel": desc_c}
    out["configs[1]"] = {
        "workload": "one scenario, 10k partitions x 100 brokers x 10 racks, RF 3, decommission 1 broker",
        "gpu_ms_per_solve": ms, "gpu_fill_kernel_us": f_us, "gpu_order_kernel_us": o_us, "kernel": desc,
        "roofline_frac": fb.algorithmic_bytes() / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
        "with_the_adapters_context_in_and_out": with_ctx,
        "cpu_fast_one_core_ms": c1, "cpu_fast_all_cores_ms": c1,
        "note": "a single scenario has no scenario-level parallelism for the host (all-core = one core) and little for "
                "the GPU: ~1.6k dependent solver steps; one CPU core and the GPU take about the same time here",
    }
    # configs[4]: 1M partitions x 5k brokers x 40 racks, RF 5, remove every 50th broker + add 200; one scenario and a
    # what-if batch of 16 over the same snapshot (rack map on / off alternating)
    P5, N5, R5, RF5 = 1000000, 5000, 40, 5
    cur5 = G.random_assignment(7, P5, N5, R5, RF5)
    one = G.perturb_brokers(N5, R5, remove=list(range(0, N5, 50)), add=200, rack_aware=True)
    fb1 = uniform_batch(cur5[None], one.node_id[None], one.node_rack[None], RF5)
    ms1, f1, o1, desc1, sr1, rows1 = gpu_ms(fb1, n=5)
    c5_1, want1 = cpu_ms(fb1, 1, n=3)
    check(fb1, sr1, rows1, want1, "configs[4] x1")
    out["configs[4]"] = {
        "workload": "1M partitions x 5k brokers x 40 racks, RF 5, remove every 50th broker + add 200 (N = 5100, cap 981)",
        "one_scenario": {"gpu_ms_per_solve": ms1, "gpu_fill_us": f1, "gpu_order_kernel_us": o1, "kernel": desc1,
                         "roofline_frac": fb1.algorithmic_bytes() / (ms1 * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                         "cpu_fast_one_core_ms": c5_1, "moved_replicas": int(sr1["moved_replicas"][0])},
        "host_hardware_threads": host_threads(),
    }
    for nb in (16, 64):                                       # 64 = the config's whole what-if batch on ONE GPU
        sets = [G.perturb_brokers(N5, R5, remove=list(range(k % 50, N5, 50)), add=200, rack_aware=(k % 2 == 0)) for k in range(nb)]
        fbn = uniform_batch(cur5, np.stack([b.node_id for b in sets]), np.stack([b.node_rack for b in sets]), RF5, shared_cur=True)
        msn, fn_, on_, descn, srn, rowsn = gpu_ms(fbn, n=3)
        c5_n, wantn = cpu_ms(fbn, 0, n=3 if nb <= 16 else 1)
        check(fbn, srn, rowsn, wantn, f"configs[4] x{nb}")
        out["configs[4]"][f"batch_of_{nb}"] = {
            "what": f"{nb} broker-set variants of the same snapshot (rack map on / off alternating) in one batch",
            "gpu_ms_per_batch": msn, "gpu_scenarios_per_s": 1e3 * nb / msn, "gpu_fill_us": fn_,
            "gpu_order_kernel_us": on_, "kernel": descn,
            "roofline_frac": fbn.algorithmic_bytes() / (msn * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            "cpu_fast_all_cores_ms": c5_n, "cpu_fast_all_cores_scenarios_per_s": 1e3 * nb / c5_n,
            "cpu_threads": wantn.threads_used}
        del fbn, srn, rowsn, wantn
    # configs[3], one GPU's share: 8000 scenarios in one batch, the exact action "add brokers 1000-1049", two batches
    # in flight (the config names 64k scenarios over 8 GPUs)
    import copy
    a3 = copy.copy(args)
    a3.scenarios, a3.in_flight, a3.steps, a3.same_batch = 8000, 2, 4, False
    torch.cuda.empty_cache()
    run3 = HipRun(a3, 0, 1, run.device_index, 0, a3.scenarios, ("add50",))
    try:
        walls = []
        for _ in range(3):
            run3.synchronize()
            t0 = time.perf_counter()
            for i in range(a3.steps):
                run3.solve(run3.slots[i % run3.n_slots])
            run3.synchronize()
            walls.append(time.perf_counter() - t0)
        w3 = sorted(walls)[1]
        sl0 = run3.slots[0]
        sr3 = run3.records_tensor(sl0).cpu().numpy().view(abi.SCENARIO_RESULT_DTYPE).copy()
        # records of a sample against the flat-array CPU solver (all host threads), which is also the CPU figure
        n_cpu = 128
        from kafka_assigner_amd.flatten import node_set_batch
        sub = node_set_batch(run3.check_ids(sl0)[:n_cpu], sl0["racks"][:n_cpu], a3.partitions, a3.rf, a3.rf, cur=run3.check_cur(0, list(range(n_cpu))))
        t0 = time.perf_counter()
        want3 = cpu_fast_solve(sub, threads=0)
        c3 = time.perf_counter() - t0
        for f in ("status", "fail_partition", "moved_replicas", "moved_partitions", "digest"):
            assert (sr3[f][:n_cpu] == want3.scenario_results[f][:n_cpu]).all(), f"configs[3]: {f} differs from the CPU solver"
        out["configs[3]"] = {
            "workload": "one GPU's share of 64k scenarios: 8000 scenarios of 100k partitions x 1k brokers x 20 racks, RF 3, "
                        "action add brokers 1000-1049 (N = 1050, cap 286), one batch of 8000, 2 batches in flight",
            "gpu_scenarios_per_s": a3.scenarios * a3.steps / w3, "gpu_ms_per_batch": 1e3 * w3 / a3.steps,
            "roofline_frac": run3.algorithmic_bytes() / (w3 / a3.steps) / 1e9 / HBM_PEAK_GBPS,
            "kernel": run3.describe(), "ok_scenarios": int((sr3["status"] == abi.KAS_OK).sum()),
            "parity_checked_scenarios": n_cpu,
            "cpu_fast_all_cores_scenarios_per_s": n_cpu / c3, "cpu_threads": want3.threads_used,
        }
    finally:
        run3.close()
    return out


def main() -> int:
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return spawn_ranks(args)
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
