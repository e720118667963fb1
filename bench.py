#!/usr/bin/env python
"""bench.py — assignment scenarios/sec on MI355X (BASELINE.json metric).

One "step" = one solve of one batch of synthetic cluster scenarios by the HIP path, with every
bulk table already resident in HBM.  At N=1 the workload is BASELINE.json configs[2] — the
configuration the metric is quoted on: a batch of 1k independent scenarios of 100k partitions x
1k brokers x 20 racks, RF 3, each with its own current assignment G(seed+s) and its own broker-set
perturbation drawn from {remove 1, remove k<=5, add k<=50, remove k<=5 + add j<=50} (SURVEY.md
8d; 'replace 1' is excluded because the reference itself throws on most such scenarios, see
generator.BENCH_ACTIONS).  With
--gpus N every rank solves its own 1k scenarios (weak scaling) and each step ends with the ONE
data-path collective of the design: an RCCL all-gather of the 32-byte per-scenario result records.

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline     — algorithmic HBM bytes per launch / average kernel duration (HIP events on the
                 launch stream) against the 8 TB/s HBM3E peak
  cpu_baseline — the CPU oracle (C restatement of the reference Java; no JVM exists here) timed on
                 one host core over a bounded sample of the same scenarios
and checks a sample of the GPU results list-for-list against the oracle before reporting.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# The HIP runtime multiplexes streams onto 4 hardware queues by default; steps in flight on more
# streams than that would serialise in pairs.  Must be set before the runtime is loaded.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--scenarios", type=int, default=1000, help="scenarios per GPU per step")
    ap.add_argument("--partitions", type=int, default=100000)
    ap.add_argument("--brokers", type=int, default=1000)
    ap.add_argument("--racks", type=int, default=20)
    ap.add_argument("--rf", type=int, default=3)
    ap.add_argument("--seed", type=int, default=2026)
    ap.add_argument("--actions", default="", help="comma list overriding the per-scenario action mix "
                    "(remove1,remove_k,add_k,mixed,replace1); default generator.BENCH_ACTIONS")
    ap.add_argument("--check", type=int, default=8, help="scenarios list-compared against the oracle")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU-baseline sample budget")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--in-flight", type=int, default=8,
                    help="batches in flight: steps are issued round-robin on this many HIP streams, "
                         "each with its own plan scratch and output tables")
    ap.add_argument("--stats", default="", help="write the per-phase device counters (JSON) here")
    ap.add_argument("--waves", type=int, default=int(os.environ.get("KAS_BENCH_WAVES", "0")),
                    help="wavefronts per scenario workgroup (0 = the plan's choice)")
    ap.add_argument("--groups", type=int, default=int(os.environ.get("KAS_BENCH_GROUPS", "0")),
                    help="scenarios per wavefront of the ticket-form order kernel (0 = the plan's choice)")
    ap.add_argument("--plan-flags", type=int, default=0, help="KAS_PLAN_* switches (testing)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from kafka_assigner_amd import abi, generator as G, native, sharding
    from kafka_assigner_amd.flatten import node_set_batch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus or world == 1, f"WORLD_SIZE {world} != --gpus {args.gpus}"
    assert torch.cuda.is_available(), "bench.py needs a GPU: the product has no CPU path"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    S, P, N, R, RF = args.scenarios, args.partitions, args.brokers, args.racks, args.rf
    first = rank * S                                   # global index of this rank's first scenario

    # ---- synthetic inputs, generated straight into HBM -----------------------------------------
    gen = torch.Generator(device=dev)
    gen.manual_seed(args.seed + 7919 * rank)
    d_cur = G.torch_random_assignment(gen, S, P, N, R, RF, dev)          # int32 [S, P, RF]
    action_mix = tuple(a for a in args.actions.split(",") if a) or G.BENCH_ACTIONS
    actions, ids, racks = [], [], []
    for s in range(S):
        act, bs = G.scenario_action(args.seed, first + s, N, R, actions=action_mix)
        actions.append(act); ids.append(bs.node_id); racks.append(bs.node_rack)
    fb = node_set_batch(ids, racks, P, RF, RF)
    ctx = native.DeviceContext(local_rank)
    # Each in-flight slot owns a plan (accept-mask scratch), output tables and a dedicated HIP
    # stream shared by its solver launches and its RCCL all-gather (handle 0, torch's default
    # stream, would select the library's own stream instead).  Consecutive steps go to
    # consecutive slots, so independent batches overlap on the GPU; every step still solves the
    # whole batch and produces its own full outputs.
    n_slots = max(1, min(args.in_flight, args.steps))
    slots = []
    for _ in range(n_slots):
        plan_ = native.Plan(ctx, fb)
        if args.waves or args.groups or args.plan_flags:
            plan_.set_flags((args.waves << 8) | (args.groups << 12) | args.plan_flags)
        sl = {"plan": plan_,
              "out": torch.empty(fb.out_len, dtype=torch.int32, device=dev),
              "tr": torch.zeros(S * 16, dtype=torch.uint8, device=dev),
              "sr": torch.zeros(S * 32, dtype=torch.uint8, device=dev),
              "stream": torch.cuda.Stream(dev)}
        sl["all"] = torch.zeros(world * S * 32, dtype=torch.uint8, device=dev) if world > 1 else sl["sr"]
        sl["stream"].wait_stream(torch.cuda.current_stream(dev))
        slots.append(sl)
    plan, d_out, d_tr, d_sr = (slots[0][k] for k in ("plan", "out", "tr", "sr"))
    step_no = [0]

    def step():
        sl = slots[step_no[0] % n_slots]
        step_no[0] += 1
        sl["plan"].solve_device(d_cur.data_ptr(), sl["out"].data_ptr(), sl["tr"].data_ptr(),
                                sl["sr"].data_ptr(), stream=sl["stream"].cuda_stream)
        if world > 1:                                   # the single data-path collective
            with torch.cuda.stream(sl["stream"]):
                sharding.gather_records(sl["sr"], world * S, out=sl["all"])

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    for sl in slots:
        sl["plan"].kernel_time_us()                     # reset the kernel-event accumulators
    step_no[0] = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    fill_us, order_us, kern_n = 0.0, 0.0, 0
    for sl in slots:
        f, o, n = sl["plan"].phase_times_us()
        fill_us += f * n; order_us += o * n; kern_n += n
    fill_us = fill_us / kern_n if kern_n else 0.0
    order_us = order_us / kern_n if kern_n else 0.0
    kern_us = fill_us + order_us
    if args.stats and rank == 0:
        st = plan.stats().astype(np.float64)
        names = ["setup_us", "p2_hist_quota_us", "p2_keep_p3_us", "p4_us", "p4_windows", "p4_steps",
                 "p5_rounds_or_queue_steps", "p2_ranked_tiles_wave0", "order_us", "solver_iterations", "solver_queue_rounds",
                 "solver_blocked", "stager_iterations", "stager_idle", "solver_queue_rows", "solver_rows_in_hand"]
        scale = [0.01, 0.01, 0.01, 0.01, 1, 1, 1, 1, 0.01, 1, 1, 1, 1, 1, 1, 1]
        summary = {n: {"mean": float(st[:, i].mean() * scale[i]), "max": float(st[:, i].max() * scale[i]),
                       "min": float(st[:, i].min() * scale[i])} for i, n in enumerate(names)}
        summary["fill_kernel_avg_us"] = fill_us
        summary["order_kernel_avg_us"] = order_us
        json.dump(summary, open(args.stats, "w"), indent=1)

    # ---- results of this rank -------------------------------------------------------------------
    sr = d_sr.cpu().numpy().view(abi.SCENARIO_RESULT_DTYPE)
    ok = int((sr["status"] == abi.KAS_OK).sum())
    all_sr = slots[0]["all"].cpu().numpy().view(abi.SCENARIO_RESULT_DTYPE)
    if world > 1:
        assert (all_sr[first:first + S] == sr).all(), "all-gather returned a different record"

    out_line = None
    if rank == 0:
        from oracle_lib import oracle_solve
        # ---- parity: list-compare a sample against the oracle ------------------------------------
        n_check = max(0, min(args.check, S))
        pick = list(range(n_check))
        # make sure at least one failing scenario (if any) is in the sample
        bad = np.nonzero(sr["status"] != abi.KAS_OK)[0]
        if len(bad) and int(bad[0]) not in pick and n_check:
            pick[-1] = int(bad[0])
        checked = 0
        if pick:
            h_cur = torch.stack([d_cur[s] for s in pick]).cpu().numpy()
            sub = node_set_batch([ids[s] for s in pick], [racks[s] for s in pick], P, RF, RF, cur=h_cur)
            want = oracle_solve(sub)
            ow = RF
            for i, s in enumerate(pick):
                got_rows = d_out[s * P * ow:(s + 1) * P * ow].cpu().numpy()
                assert (got_rows == want.out[i * P * ow:(i + 1) * P * ow]).all(), f"scenario {s}: lists differ from the oracle"
                for f in ("status", "fail_partition", "moved_replicas", "moved_partitions", "digest"):
                    assert sr[f][s] == want.scenario_results[f][i], f"scenario {s}: {f} differs"
                checked += 1

        # ---- CPU baseline: the oracle on a bounded sample of the same workload -------------------
        cpu = None
        if not args.no_cpu and world == 1:
            m = 4
            t_c0 = time.perf_counter()
            done = 0
            cpu_time = 0.0
            while True:
                idx = [(done + i) % S for i in range(m)]
                h_cur = torch.stack([d_cur[s] for s in idx]).cpu().numpy()
                sub = node_set_batch([ids[s] for s in idx], [racks[s] for s in idx], P, RF, RF, cur=h_cur)
                t1 = time.perf_counter()
                oracle_solve(sub)
                dt = time.perf_counter() - t1
                done += m
                cpu_time += dt
                if time.perf_counter() - t_c0 > args.cpu_seconds or done >= S:
                    break
                m = min(64, m * 2)
            # the same port on many host cores (scenario-parallel, one thread per sub-batch; ctypes
            # releases the GIL inside the C solver): what a whole host does, for scale
            from concurrent.futures import ThreadPoolExecutor
            n_thr = max(1, min(os.cpu_count() or 1, 128))
            per = 8
            subs = []
            for t in range(n_thr):
                idx = [(t * per + i) % S for i in range(per)]
                h_cur = torch.stack([d_cur[s] for s in idx]).cpu().numpy()
                subs.append(node_set_batch([ids[s] for s in idx], [racks[s] for s in idx], P, RF, RF, cur=h_cur))
            t1 = time.perf_counter()
            with ThreadPoolExecutor(max_workers=n_thr) as ex:
                list(ex.map(oracle_solve, subs))
            thr_time = time.perf_counter() - t1
            cpu_threads = {"value": n_thr * per / thr_time, "unit": "scenarios/s", "cores": n_thr,
                           "sample": f"{n_thr * per} scenarios, {per} per thread, {thr_time:.1f} s wall"}
            cpu = {"value": done / cpu_time, "unit": "scenarios/s", "cores": 1, "kind": "port",
                   "many_cores": cpu_threads,
                   "sample": f"{done} scenarios of the same batch, oracle/kas_oracle.c (C restatement of "
                             f"the reference Java, rescans order[0..] per orphan like KAS:175), 1 thread, "
                             f"{cpu_time:.1f} s solve time; host has {os.cpu_count()} cores; no JVM in this image"}

        alg_bytes = plan.algorithmic_bytes
        achieved = alg_bytes / (kern_us * 1e-6) / 1e9 if kern_us > 0 else 0.0
        value = world * S * args.steps / elapsed
        out_line = {
            "metric": "assignment scenarios/sec at 100k partitions x 1k brokers RF=3",
            "value": value, "unit": "scenarios/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32",
            "data": "synthetic",
            "config": {
                "workload": f"{'BASELINE.json configs[2]' if (S, P, N, R, RF) == (1000, 100000, 1000, 20, 3) else 'custom shape'}: "
                            f"batch of {S} independent scenarios per GPU, "
                            f"{P} partitions x {N} brokers x {R} racks, RF {RF}; per-scenario G(seed+s) "
                            f"current assignment + action in {{{', '.join(action_mix)}}} (remove <= 5, add <= 50)",
                "scenarios_per_gpu": S, "partitions": P, "brokers": N, "racks": R, "rf": RF,
                "ok_scenarios_rank0": ok, "failed_scenarios_rank0": S - ok,
                "failed_note": "a failed scenario is the reference's own KAS:183-184 stranding, "
                               "reproduced bit-exactly (status + partition id)",
                "parity_checked_scenarios": checked,
                "collective": "all_gather of 32-byte result records per step" if world > 1 else "none (1 GPU)",
                "batches_in_flight": n_slots, "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                "kernel": "kas_fill_kernel<3,4> + kas_order_ticket_kernel<3,2,true> (one solve = both, same stream)",
                "kernel_avg_us": kern_us, "fill_kernel_avg_us": fill_us, "order_kernel_avg_us": order_us,
                "launches_timed": kern_n,
                "achieved_wall": value * alg_bytes / (world * S) / 1e9,
                "note": "durations are HIP-event times per launch while batches_in_flight solves share "
                        "the GPU; achieved = algorithmic bytes of one solve / (fill + order duration); "
                        "achieved_wall = algorithmic bytes per second at the measured whole-job rate",
                "algorithmic_bytes_per_launch": alg_bytes,
            },
            "cpu_baseline": cpu,
        }
        # PMC-measured HBM traffic of the same command, when a committed profile provides it
        try:
            prof = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            if prof.get("scenarios") == S and prof.get("partitions") == P:
                out_line["roofline"]["traffic"] = prof["hbm_bytes_per_launch"]
                out_line["roofline"]["traffic_source"] = prof.get("source")
        except Exception:
            pass
        print(json.dumps(out_line), flush=True)
    for sl in slots:
        sl["plan"].close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
