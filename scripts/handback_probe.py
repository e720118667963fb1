"""scripts/handback_probe.py [S] [FRACTION]: what a batch costs whose scenarios the slim fill kernel hands back (rows that are not
rack-diverse: the general fill of kas_fill_kernel, launched behind the slim kernel on at most 256 workgroups).  The bench shape
(S x 100,000 x 1,000 x RF 3, int32 cells), current assignments that are rack-diverse over 20 racks; FRACTION of the scenarios get a
rack map of 10 racks instead (two brokers of a row then share a rack now and then: the whole scenario goes back).  Times the
plan's default against KAS_PLAN_FULL_FILL (kas_fill_kernel for every scenario on the full grid) and checks the records of the
two against each other.  MEASUREMENT TOOLING (GPU only)."""
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import torch
from kafka_assigner_amd import abi, native
from kafka_assigner_amd import generator as G
from kafka_assigner_amd.flatten import node_set_batch

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
FR = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
P, N, R, RF = 100000, 1000, 20, 3
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev); gen.manual_seed(7)
d_cur = G.torch_random_assignment(gen, S, P, N, R, RF, dev)
rng = np.random.default_rng(3)
ids, racks = [], []
for s in range(S):
    _, bs = G.scenario_action(2026, s, N, R, actions=("remove1", "add_k"))
    ids.append(bs.node_id)
    racks.append((bs.node_id % 10).astype(np.int32) if rng.random() < FR else bs.node_rack)
fb = node_set_batch(ids, racks, P, RF, RF)
ctx = native.default_context()
d_tr = torch.zeros(S * 16, dtype=torch.uint8, device=dev)
d_sr = torch.zeros(S * 32, dtype=torch.uint8, device=dev)
d_out = torch.empty((fb.out_len,), dtype=torch.int32, device=dev)
st = torch.cuda.Stream(dev)
recs = {}
for name, flags in (("default (slim kernel, full kernel behind it for what is handed back)", 0), ("KAS_PLAN_FULL_FILL", abi.KAS_PLAN_FULL_FILL)):
    plan = native.Plan(ctx, fb)
    if flags:
        plan.set_flags(flags)
    for _ in range(2):
        plan.solve_device(d_cur.data_ptr(), d_out.data_ptr(), d_tr.data_ptr(), d_sr.data_ptr(), stream=st.cuda_stream)
    st.synchronize()
    plan.phase_times_us()
    for _ in range(5):
        plan.solve_device(d_cur.data_ptr(), d_out.data_ptr(), d_tr.data_ptr(), d_sr.data_ptr(), stream=st.cuda_stream)
    st.synchronize()
    f, o, n = plan.phase_times_us()
    sr = d_sr.cpu().numpy().view(abi.SCENARIO_RESULT_DTYPE).copy()
    recs[name] = sr
    print(f"{name}: fill + first fit {f:.0f} us, order {o:.0f} us per batch of {S} ({n} launches); failed scenarios {int((sr['status'] != 0).sum())}")
    print("   ", plan.describe()[:230])
    plan.close()
a, b = list(recs.values())
for f in ("status", "fail_partition", "moved_replicas", "moved_partitions", "digest"):
    assert (a[f] == b[f]).all(), f
print("records equal")
