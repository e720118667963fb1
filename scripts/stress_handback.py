"""scripts/stress_handback.py SECONDS [SEED]: random batches of MANY small scenarios of which a random share starts from rows that
are not rack-diverse — the slim fill kernel hands those back, kas_fill_kernel behind it deals them to its workgroups by rank and
the plan sizes that launch by the count the previous solve left in pinned host memory (kas_plan_back_grid).  Every batch is solved
three times on ONE plan (grid 256, then grown, then grown again or the same) and once more on a fresh plan with
KAS_PLAN_FULL_FILL, each time compared bit for bit with the CPU oracle; the describe string must show the grid the count implies.
MEASUREMENT / TEST TOOLING (GPU only; not part of the pytest suites)."""
import re
import sys
import time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import torch
from oracle_lib import oracle_solve
from parity_util import assert_same_outputs
from kafka_assigner_amd import abi, native
from kafka_assigner_amd import generator as G
from kafka_assigner_amd.flatten import uniform_batch, host_tables

T = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
ctx = native.default_context()
dev = torch.device("cuda", ctx.device)
st = torch.cuda.Stream(dev)
t_end = time.time() + T
n, n_grown, n_back_total, shares = 0, 0, 0, []
while time.time() < t_end:
    S = int(rng.integers(257, 1100))
    P = int(rng.integers(64, 700))
    R = int(rng.choice([4, 5, 6, 8]))
    N = R * int(rng.integers(4, 12))
    rf = int(rng.choice([2, 3, 3]))
    drop = int(rng.integers(0, 3))
    share = float(rng.choice([0.0, 0.02, 0.1, 0.3, 0.6, 1.0]))
    back = rng.random(S) < share
    racks = (np.arange(N) % R).astype(np.int32)
    ids = np.arange(N, dtype=np.int32)
    # (cyclic rows stepping by R land every holder of a row on ONE rack: not rack-diverse)
    curs = [((G.cyclic_assignment(P, N, rf, s) * R) % N) if back[s] else G.random_assignment(int(rng.integers(1 << 30)), P, N, R, rf) for s in range(S)]
    fb = uniform_batch(np.stack(curs).astype(np.int32), np.tile(ids, (S, 1))[:, :N - drop], np.tile(racks, (S, 1))[:, :N - drop], rf)
    want = oracle_solve(fb, threads=0)
    plan = native.Plan(ctx, fb)
    d0 = plan.describe()
    slim = d0.startswith("kas_fill_slim_kernel")
    d_cur = torch.from_numpy(fb.cur).to(dev)
    for turn in range(3):
        _, ho = host_tables(fb)
        d_out = torch.full((fb.out_len,), -2, dtype=torch.int32, device=dev)
        d_tr = torch.zeros(fb.n_topics * 16, dtype=torch.uint8, device=dev)
        d_sr = torch.zeros(S * 32, dtype=torch.uint8, device=dev)
        st.wait_stream(torch.cuda.current_stream(dev))
        plan.solve_device(d_cur.data_ptr(), d_out.data_ptr(), d_tr.data_ptr(), d_sr.data_ptr(), stream=st.cuda_stream)
        st.synchronize()
        ho.out = d_out.cpu().numpy()
        ho.topic_results = d_tr.cpu().numpy().view(abi.TOPIC_RESULT_DTYPE)
        ho.scenario_results = d_sr.cpu().numpy().view(abi.SCENARIO_RESULT_DTYPE)
        assert_same_outputs(fb, want, ho, f"batch {n}: S {S} P {P} N {N} R {R} rf {rf} share {share} solve {turn + 1}")
        if slim:
            nb = int(back.sum())
            g = 256
            if nb + nb // 4 > 256:
                g = min(((nb + nb // 4 + 63) // 64) * 64, S)
            m = re.search(r"kas_fill_kernel<\d,4>\[quota, chunk histograms\] grid=(\d+)x256", plan.describe())
            assert m and int(m.group(1)) == min(g, S), (nb, S, g, plan.describe())
            n_grown += 1 if (turn == 2 and g > 256) else 0
    plan.close()
    assert_same_outputs(fb, want, native.solve_host_with_flags(fb, abi.KAS_PLAN_FULL_FILL), f"batch {n}: the same without the slim kernel")
    n += 1; n_back_total += int(back.sum()); shares.append(share)
print(f"hand-back stress ok: {n} random batches of 257-1,099 scenarios x (3 solves on one plan + 1 with KAS_PLAN_FULL_FILL), "
      f"{n_back_total} scenarios handed back in all, {n_grown} batches whose launch behind the slim kernel grew past 256 workgroups; "
      f"seed {int(sys.argv[2]) if len(sys.argv) > 2 else 2026}")
