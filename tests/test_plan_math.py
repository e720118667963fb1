"""Planning decisions of csrc/kas_plan_math.h (kas_shape_batch), through the emulator build of the
same header: which order kernel serves a shape, and that a shape no kernel can serve is refused at
plan time (KAS_E_UNSUPPORTED) rather than at its first launch."""
import numpy as np

from emu_lib import plan_shape
from kafka_assigner_amd import abi
from kafka_assigner_amd.flatten import node_set_batch


def _shape(n_nodes, rf, P=1000, S=2):
    ids = [np.arange(n_nodes, dtype=np.int32)] * S
    racks = [(np.arange(n_nodes) % 20).astype(np.int32)] * S
    return plan_shape(node_set_batch(ids, racks, P, rf, rf))


def test_headline_shape_takes_the_packed_ticket_form_two_scenarios_per_wavefront():
    rc, sh, _ = _shape(1050, 3, P=100000)
    assert rc == 0 and sh["tickets_ok"] == 1 and sh["packed_ok"] == 1 and sh["G"] == 2 and sh["NW"] == 4
    assert sh["fused_ok"] == 1 and sh["round_fits"] == 1


def test_groups_shrink_before_the_ticket_form_is_given_up():
    # 14 B of LDS per broker and scenario: two groups fit 64 KiB of 16-bit offsets up to ~2,340 brokers
    rc, sh, _ = _shape(3000, 3)
    assert rc == 0 and sh["tickets_ok"] == 1 and sh["G"] == 1
    rc, sh, _ = _shape(5000, 3)                      # one group no longer fits: round form (24 B per broker)
    assert rc == 0 and sh["tickets_ok"] == 0 and sh["round_fits"] == 1


def test_shape_that_no_order_kernel_serves_is_refused_at_plan_time():
    """at ~7,000 brokers x RF 3 the round form does not fit 160 KiB and the ticket form's
    counter rows leave the 16-bit offset range; the plan used to be created (tickets_ok was cleared
    after the fallback check) and failed at its first launch with KAS_E_HIP."""
    rc, _, err = _shape(7000, 3)
    assert rc == abi.KAS_E_UNSUPPORTED, (rc, err)
    assert "LDS" in err
    rc, sh, _ = _shape(6800, 3)                      # just inside the round form's limit
    assert rc == 0 and sh["tickets_ok"] == 0 and sh["round_fits"] == 1


def test_wide_lists_take_the_wide_ticket_form_also_where_the_round_form_does_not_fit():
    rc, sh, _ = _shape(5100, 5, P=100000)
    assert rc == 0 and sh["wide_ok"] == 1 and sh["tickets_ok"] == 0
