"""Planning decisions of csrc/kas_plan_math.h (kas_shape_batch), through the emulator build of the
same header: which order kernel serves a shape, and that a shape no kernel can serve is refused at
plan time (KAS_E_UNSUPPORTED) rather than at its first launch."""
import numpy as np

from emu_lib import plan_shape, spread_plan
from kafka_assigner_amd import abi
from kafka_assigner_amd.flatten import node_set_batch


def _shape(n_nodes, rf, P=1000, S=2):
    ids = [np.arange(n_nodes, dtype=np.int32)] * S
    racks = [(np.arange(n_nodes) % 20).astype(np.int32)] * S
    return plan_shape(node_set_batch(ids, racks, P, rf, rf))


def test_headline_shape_takes_the_packed_ticket_form_two_scenarios_per_wavefront():
    rc, sh, _ = _shape(1050, 3, P=100000)
    assert rc == 0 and sh["relax_ok"] == 1          # round 4: what the plan launches; the ticket form on request:
    assert rc == 0 and sh["tickets_ok"] == 1 and sh["packed_ok"] == 1 and sh["G"] == 2 and sh["NW"] == 4
    assert sh["fused_ok"] == 1 and sh["round_fits"] == 1


def test_groups_shrink_before_the_ticket_form_is_given_up():
    # 14 B of LDS per broker and scenario: two groups fit 64 KiB of 16-bit offsets up to ~2,340 brokers
    rc, sh, _ = _shape(3000, 3)
    assert rc == 0 and sh["tickets_ok"] == 1 and sh["G"] == 1
    # round 3: one group is limited by where its COUNTER ROWS end (8 B per broker below 64 KiB), not its whole region
    rc, sh, _ = _shape(5000, 3)                      # (round 2: round form from 4,680 brokers on)
    assert rc == 0 and sh["tickets_ok"] == 1 and sh["G"] == 1 and sh["round_fits"] == 1
    rc, sh, _ = _shape(8191, 3)                      # the last broker count whose padding row has a 16-bit offset
    assert rc == 0 and sh["tickets_ok"] == 1 and sh["G"] == 1 and sh["round_fits"] == 0
    # round 4: beyond that the relaxation form (4 B of LDS per broker, no 16-bit offsets) serves lists <= 3 wide
    # that hand no Context in; what limits the broker count then is the fill kernel's LDS
    rc, sh, _ = _shape(8192, 3)
    assert rc == 0 and sh["tickets_ok"] == 0 and sh["relax_ok"] == 1


def test_shape_that_no_order_kernel_serves_is_refused_at_plan_time():
    """at ~7,000 brokers x RF 3 the round form does not fit 160 KiB and the ticket form's
    counter rows leave the 16-bit offset range; the plan used to be created (tickets_ok was cleared
    after the fallback check) and failed at its first launch with KAS_E_HIP.
    (Round 3: the ticket form's limit moved to 8,191 brokers, so the shapes nothing serves start there.)"""
    rc, sh, _ = _shape(9000, 3)                      # (round 4: the relaxation form serves it)
    assert rc == 0 and sh["tickets_ok"] == 0 and sh["round_fits"] == 0 and sh["relax_ok"] == 1
    rc, _, err = _shape(9000, 4)                     # lists 4 wide: no wide ticket form beyond 8,191 brokers, no round form either
    assert rc == abi.KAS_E_UNSUPPORTED, (rc, err)
    assert "LDS" in err
    rc, sh, _ = _shape(7000, 3)                      # beyond the round form's limit, inside the ticket form's
    assert rc == 0 and sh["tickets_ok"] == 1 and sh["round_fits"] == 0
    rc, sh, _ = _shape(6800, 3)                      # just inside the round form's limit
    assert rc == 0 and sh["tickets_ok"] == 1 and sh["round_fits"] == 1


def test_wide_lists_take_the_wide_ticket_form_also_where_the_round_form_does_not_fit():
    rc, sh, _ = _shape(5100, 5, P=100000)
    assert rc == 0 and sh["wide_ok"] == 1 and sh["tickets_ok"] == 0


def test_wide_form_is_kept_up_to_2039_rows_per_broker_with_its_count_fields_checked():
    """Round 3: lists 4-5 wide.  Below 1,023 rows per broker the 10-bit count fields are safe a priori; from there up
    to 2,039 (the 11-bit commits field) the wide form runs with its check at the end; beyond that the round form."""
    rc, sh, _ = _shape(5000, 5, P=1000000)           # BASELINE configs[4]: 1,000 rows per broker
    assert rc == 0 and sh["wide_ok"] == 1 and sh["wide_checked"] == 0
    rc, sh, _ = _shape(5000, 5, P=1100000)           # 1,100
    assert rc == 0 and sh["wide_ok"] == 1 and sh["wide_checked"] == 1 and sh["round_fits"] == 1
    rc, sh, _ = _shape(100, 4, P=50000)              # 2,000
    assert rc == 0 and sh["wide_ok"] == 1 and sh["wide_checked"] == 1
    rc, sh, _ = _shape(100, 4, P=51000)              # 2,040: the round form
    assert rc == 0 and sh["wide_ok"] == 0 and sh["wide_checked"] == 0 and sh["round_fits"] == 1
    rc, sh, _ = _shape(6200, 5, P=1300000)           # 1,049 rows per broker, but no room for the round form to fall back on
    assert (rc != 0) or sh["wide_checked"] == 0


def test_spread_scan_kernels_fit_two_to_a_cu_at_5000_brokers_and_their_chunks_fit_uint16_cells():
    """Round 3: passes A and B of the spread fill carry only what they touch (71 / 61 KB instead of the
    one-workgroup layout's 145 KB at BASELINE configs[4]), and pass A counts in uint16 cells: a chunk stays
    below 1,023 tiles of 64 rows, whatever the batch size asks for."""
    ids = [np.arange(5100, dtype=np.int32)]
    racks = [(np.arange(5100) % 40).astype(np.int32)]
    for S in (1, 8, 64):
        rc, sp = spread_plan(node_set_batch(ids * S, racks * S, 1000000, 5, 5))
        assert rc == 0 and sp["chunks"] >= 4, (S, sp)
        assert (sp["tiles"] + sp["chunks"] - 1) // sp["chunks"] < 1023, (S, sp)
        assert 2 * sp["lds_a"] <= 160 * 1024 and 2 * sp["lds_b"] <= 160 * 1024 and sp["lds_full"] > 80 * 1024, sp
    assert spread_plan(node_set_batch(ids * 64, racks * 64, 1000000, 5, 5))[1]["chunks"] == 16
    # small scenarios and big batches stay with the one-workgroup kernel
    assert spread_plan(node_set_batch(ids * 2, racks * 2, 100000, 5, 5))[1]["chunks"] == 0
    assert spread_plan(node_set_batch(ids * 65, racks * 65, 1000000, 5, 5))[1]["chunks"] == 0
