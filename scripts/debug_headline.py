"""Debug aid: the headline launch several times on fresh plans, statuses against the CPU solver."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from kafka_assigner_amd import abi, generator as G, native
from kafka_assigner_amd.flatten import node_set_batch
from oracle_lib import cpu_fast_solve
S, P, N, R, RF = 1000, 100000, 1000, 20, 3
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev); gen.manual_seed(2026)
d_cur = G.torch_random_assignment(gen, S, P, N, R, RF, dev)
ids, racks = [], []
for s in range(S):
    _, bs = G.scenario_action(2026, s, N, R, actions=G.BENCH_ACTIONS)
    ids.append(bs.node_id); racks.append(bs.node_rack)
fb = node_set_batch(ids, racks, P, RF, RF)
full = node_set_batch(ids, racks, P, RF, RF, cur=d_cur.cpu().numpy())
fast = cpu_fast_solve(full, threads=0)
ctx = native.default_context()
torch.cuda.synchronize()
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    flags = [0, abi.KAS_PLAN_RELAX_TILES_128, abi.KAS_PLAN_FILL_WITH_P4, 0, abi.KAS_PLAN_INDEX_ROWS, 0][rep % 6]
    plan = native.Plan(ctx, fb)
    if flags:
        plan.set_flags(flags)
    d_out = torch.full((fb.out_len,), -2, dtype=torch.int32, device=dev)
    d_tr = torch.zeros(S * 16, dtype=torch.uint8, device=dev)
    d_sr = torch.zeros(S * 32, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    st = torch.cuda.Stream(dev)
    plan.solve_device(d_cur.data_ptr(), d_out.data_ptr(), d_tr.data_ptr(), d_sr.data_ptr(), stream=st.cuda_stream)
    st.synchronize()
    sr = d_sr.cpu().numpy().view(abi.SCENARIO_RESULT_DTYPE)
    bad = np.nonzero(sr["status"] != fast.scenario_results["status"][:S])[0]
    badd = np.nonzero(sr["digest"] != fast.scenario_results["digest"][:S])[0]
    stt = plan.stats()
    print(f"rep {rep} flags {flags:#x}: {len(bad)} statuses differ, {len(badd)} digests differ", plan.describe()[-90:])
    if len(bad):
        print("   first", bad[:10], "got", sr["status"][bad[:10]], "want", fast.scenario_results["status"][bad[:10]])
        print("   stats of first bad: evals", stt[bad[0], 9], "tiles", stt[bad[0], 12], "slow", stt[bad[0], 13], "verified", stt[bad[0], 10], "unsound", stt[bad[0], 11])
        print("   stats of a good one: evals", stt[0, 9], "tiles", stt[0, 12], "slow", stt[0, 13])
    plan.close()
