"""Randomised GPU-vs-oracle stress (not part of the pytest suites): random shapes (8..500 brokers,
300..20000 partitions, RF 2..3, every action mix, 1..8 scenarios per batch) for N seconds, each batch
solved with the default plan, one scenario per solver wavefront, 4 x uint16 counter rows, and the spread fill
forced, and compared bit for bit with the CPU oracle.  Usage: python scripts/stress_gpu.py SECONDS
(round 1: 21,494 batches x 3 plan variants in 150 s on an MI355X, all identical).
With --emu the same draws go through the CPU fiber emulator of the kernel source instead (no GPU; KAS_EMU_CHAOS=<seed> /
KAS_EMU_WAVE_DIV=w:k in the environment change the wave schedules): python scripts/stress_gpu.py SECONDS [SEED] --emu"""
import sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
from test_emu_parity import _batch
from oracle_lib import oracle_solve
from parity_util import assert_same_outputs
from kafka_assigner_amd import generator as G
from kafka_assigner_amd import abi
from kafka_assigner_amd.flatten import FlatBatch
EMU = "--emu" in sys.argv
argv = [a for a in sys.argv if a != "--emu"]
rng = np.random.default_rng(int(argv[2]) if len(argv) > 2 else 2026)   # python scripts/stress_gpu.py SECONDS [SEED]
if EMU:
    from emu_lib import emu_solve

    def solve(fb, flags=0):
        return emu_solve(fb, flags=flags)

    def solve16(fb, flags=0):
        from emu_lib import emu_solve16
        return emu_solve16(fb, flags=flags)
else:
    from kafka_assigner_amd import native

    def solve(fb, flags=0):
        return native.solve_host_with_flags(fb, flags) if flags else native.solve_host(fb)

    def solve16(fb, flags=0):
        return native.solve_device16_with_flags(fb, flags)


n16 = 0


def check16(fb, what, flag_sets=(0, 0x20000, 0x40000, 2)):
    """the same batch on 16-bit node-index cells (kas_plan_create16 / kas_solve_device16) against the oracle on its index
    form — lists up to 3 wide"""
    global n16
    from kafka_assigner_amd.flatten import index_form
    want = oracle_solve(index_form(fb))
    want.out = np.where(want.out < 0, 0xFFFF, want.out).astype(np.uint16)
    for flags in flag_sets:
        assert_same_outputs(fb, want, solve16(fb, flags), f"{what} 16-bit cells flags {flags}")
    n16 += 1


def with_context(fb, rng):
    """Every scenario of a single-topic batch hands a Context in and wants it back: a random width (narrower and wider
    than the lists), counters as earlier topics would have left them — now and then one that leaves the relaxation
    form's and the ticket form's 16-bit fields, or is negative: that scenario must go to the round form."""
    scen = fb.scen.copy()
    width = int(rng.choice([1, 2, 3, 4, 8]))
    ctx, off = [], 0
    for s in range(fb.n_scenarios):
        n = int(scen["n_nodes"][s])
        tab = rng.integers(0, int(rng.choice([1, 50, 400, 3000])), size=(n, width)).astype(np.int32)
        roll = rng.random()
        if roll < 0.15: tab[int(rng.integers(n)), int(rng.integers(width))] = int(rng.choice([4090, 5000, 65535, 70000, 1 << 29]))
        elif roll < 0.2: tab[int(rng.integers(n)), 0] = -int(rng.integers(1, 9))
        scen["ctx_width"][s] = width
        scen["ctx_off"][s] = off
        ctx.append(tab.reshape(-1)); off += n * width
    return FlatBatch(scen=scen, topics=fb.topics, node_id=fb.node_id, node_rack=fb.node_rack, cur=fb.cur, aux=fb.aux,
                     ctx=np.concatenate(ctx), out_len=fb.out_len)


def thin_wide_batch(rng):
    """Rows 4-5 wide holding 1-2 replicas over 2-4 brokers, 1,023 .. 2,039 rows per broker: the wide ticket form
    with its count fields checked at the end — rf 1 puts every row of a broker at list position 0, so its count
    passes 1,023 and the scenario is flagged and solved again (fill + round form); rf 2 stays inside the fields."""
    W = int(rng.choice([4, 5])); rf = int(rng.choice([1, 2])); N = int(rng.choice([2, 3, 4]))
    P = int(rng.integers(1023 * N // rf + 1, 2039 * N // rf))
    S = int(rng.choice([1, 2, 3]))
    scen = np.zeros(S, dtype=abi.SCENARIO_DESC_DTYPE); topics = np.zeros(S, dtype=abi.TOPIC_DESC_DTYPE)
    ids_all, racks_all, curs = [], [], []
    for s in range(S):
        ids = np.sort(rng.choice(np.arange(1, 60), size=N, replace=False)).astype(np.int32)
        gone = 77
        rows = np.arange(P)
        cur = np.stack([ids[(rows + k + (rows // N) % N) % N] for k in range(rf)], axis=1).astype(np.int32)
        hit = rng.random(P) < float(rng.choice([0.0, 0.05, 0.3] if rf == 1 else [0.0, 0.0, 0.004]))   # (rf 2 at zero slack strands easily)
        cur[hit, 0] = gone                                       # rows of a broker that is gone: orphans
        scen[s] = (N, s, 1, 0, s * N, -1)
        topics[s] = (int(rng.integers(1, 1 << 30)), P, rf, rf, W, 0, s * P * rf, s * P * W, -1, -1, -1)
        ids_all.append(ids); racks_all.append(np.arange(N, dtype=np.int32)); curs.append(cur.reshape(-1))
    return FlatBatch(scen=scen, topics=topics, node_id=np.concatenate(ids_all), node_rack=np.concatenate(racks_all),
                     cur=np.concatenate(curs).astype(np.int32), aux=np.zeros(0, np.int32), ctx=np.zeros(0, np.int32),
                     out_len=S * P * W), f"thin W{W} rf{rf} N{N} P{P} S{S}"


t0 = time.time(); n = 0; q_rows = 0; n_thin = 0; n_big = 0; n_ctx = 0; n_mid = 0
while time.time() - t0 < float(argv[1]):
    kind = rng.random()
    if kind < 0.12:                                              # the checked wide form and its second solve
        fb, what = thin_wide_batch(rng)
        want = oracle_solve(fb)
        for flags in (0, 2):
            got = solve(fb, flags)
            assert_same_outputs(fb, want, got, f"{what} flags {flags}")
        n += 1; n_thin += 1
        continue
    if kind < 0.17:                                              # lists 3 wide, 3,000 .. 7,400 brokers: one group of the ticket form;
        N = int(rng.choice([3000, 5000, 7400, 9000, 12000])); P = int(rng.choice([8000, 30000]))   # beyond 8,191: the relaxation form only
        seed = int(rng.integers(1 << 30)); S = int(rng.choice([1, 2, 3]))
        fb = _batch(seed, S, P, N, int(rng.choice([10, 25, 40])), int(rng.choice([2, 3])), ("add_k", "mixed", "remove_k"))
        want = oracle_solve(fb)
        for flags in ((0, 4, 0x20000) if N <= 8191 else (0, 0x20000)):
            got = solve(fb, flags)
            assert_same_outputs(fb, want, got, f"seed {seed} S{S} P{P} N{N} big-N flags {flags}")
        check16(fb, f"seed {seed} S{S} P{P} N{N} big-N", (0, 0x20000))
        n += 1; n_big += 1
        continue
    if kind < 0.24:                                              # lists 3 wide, 900 .. 2,060 brokers: node indices on either side of 1,024 and up to
        N = int(rng.choice([900, 1020, 1030, 1500, 2040, 2047, 2060])); P = int(rng.choice([6000, 20000]))   # the dword mid rows' 2,046 (beyond: 16-bit rows)
        seed = int(rng.integers(1 << 30)); S = int(rng.choice([1, 2, 4]))
        fb = _batch(seed, S, P, N, int(rng.choice([10, 25, 40])), 3, ("add_k", "mixed", "remove_k", "remove1"))
        want = oracle_solve(fb)
        for flags in (0, 0xC00000, 0x100000, 0xC00000 | 0x60000, 0x800000 | 0x40000):
            got = solve(fb, flags)
            assert_same_outputs(fb, want, got, f"seed {seed} S{S} P{P} N{N} mid-N flags {flags}")
        n += 1; n_mid += 1
        continue
    N = int(rng.choice([8, 12, 20, 33, 64, 100, 150, 300, 500]))
    R = int(rng.choice([2, 3, 5, 8, 10, 20])); R = min(R, N)
    RF = int(rng.choice([2, 3, 3, 3, 4, 5])); RF = min(RF, R)     # 4, 5: the wide ticket form
    P = int(rng.choice([300, 1000, 3000, 7000, 20000]))
    acts = [("add_k",), ("remove1",), ("remove_k", "mixed"), G.ACTIONS, ("mixed", "add_k")][int(rng.integers(5))]
    seed = int(rng.integers(1 << 30))
    S = int(rng.choice([1, 2, 3, 5, 8]))
    fb = _batch(seed, S, P, N, R, RF, acts)
    if rng.random() < 0.3:                                       # with the Context the reference's adapter always hands in
        fb = with_context(fb, rng)
        want = oracle_solve(fb)
        for flags in ((0, 1 << 16, 2, 0x20000, 0x40000) if RF <= 3 else (0, 2)):
            got = solve(fb, flags)
            assert_same_outputs(fb, want, got, f"seed {seed} S{S} P{P} N{N} R{R} RF{RF} {acts} Context width {int(fb.scen['ctx_width'][0])} flags {flags}")
        if RF <= 3:
            check16(fb, f"seed {seed} S{S} P{P} N{N} R{R} RF{RF} {acts} Context width {int(fb.scen['ctx_width'][0])}")
        n += 1; n_ctx += 1
        continue
    want = oracle_solve(fb)
    # (0 = the relaxation form of the order kernel for lists <= 3 wide, 1 << 12 / 4 = its ticket forms;
    # 32 = KAS_PLAN_SPREAD_FILL: the row scans over one-wavefront workgroups with their slim LDS layouts)
    # (round 6: 0x80 / 0x40 = index rows on / off, 0xC00000 = first fit beside the order kernel in one workgroup (with either tile
    # size), 0x400000 / 0x800000 = first fit in kas_p4_kernel / inside the fill workgroup, 16 = KAS_PLAN_FULL_FILL (no slim fill kernel in front); lists 4-5 wide: 0x20000 = the relaxation
    # form for wide lists)
    for flags in ((0, 1 << 12, 4, 32, 0x20000, 0x40000, 0x200000, 0x80, 0x40 | 0x400000, 0xC00000 | 0x20000, 0xC00000 | 0x40000 | 0x80, 0x800000, 0x400000, 0x400000 | 16,
                   0x100000, 0x100000 | 0xC00000, 0x100000 | 0x400000 | 0x20000, 0x60000, 0xC00000 | 0x60000)   # (0x100000 = KAS_PLAN_NO_MID32: the packed 16-bit mid rows where the default takes a dword a row)
                  if RF <= 3 else ((0, 2, 1, 32, 0x20000, 0x20000 | 0x800000) if RF <= 5 else (0, 2, 1, 32))):
        got = solve(fb, flags)
        assert_same_outputs(fb, want, got, f"seed {seed} S{S} P{P} N{N} R{R} RF{RF} {acts} flags {flags}")
    if RF <= 3:
        check16(fb, f"seed {seed} S{S} P{P} N{N} R{R} RF{RF} {acts}", (0, 0x20000, 0x40000, 2, 1, 0x200000, 0x400000, 0x800000, 0xC00000, 0xC00000 | 0x20000, 0x20000 | (255 << 24)))
    n += 1
print("emulator stress ok:" if EMU else "stress ok:", n, "random batches x 1-4 plan variants;", n_thin, "of them thin wide rows (checked wide form),", n_big, "with 3,000-12,000 brokers,", n_mid, "with 900-2,060 brokers (dword mid rows up to 2,047),", n_ctx, "with a Context in and out,", n16, "also on 16-bit cells (kas_solve_device16); seed", int(argv[2]) if len(argv) > 2 else 2026)
