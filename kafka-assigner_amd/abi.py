"""ctypes mirror of include/kas_abi.h (structs, constants, digest function).

Pure declarations: nothing here loads a library.  Field order and types must match the header
byte for byte; tests/test_abi_layout.py checks sizes and offsets against a C probe.
"""
from __future__ import annotations

import ctypes as C

KAS_ABI_VERSION = 5
KAS_MAX_WIDTH = 8
KAS_CELL16_NONE = 0xFFFF      # 16-bit cells (kas_solve_host16): no such broker in cur, pad in out

KAS_E_OK = 0
KAS_E_INVALID_ARG = -1
KAS_E_HIP = -2
KAS_E_UNSUPPORTED = -3
KAS_E_NOMEM = -4

KAS_OK = 0
KAS_FAIL_UNASSIGNABLE = 1
KAS_FAIL_RF_NOT_POSITIVE = 2
KAS_FAIL_RF_GT_BROKERS = 3
KAS_FAIL_HASH_INDEX = 4
KAS_FAIL_RF_MISMATCH = 5
KAS_SKIPPED = 6
KAS_FAIL_BAD_NODES = 7
KAS_FAIL_WATCHDOG = 8

# plan flags (kas_plan_set_flags; include/kas_abi.h)
KAS_PLAN_GENERIC_FILL = 1
KAS_PLAN_ROUND_ORDER = 2
KAS_PLAN_WIDE_COUNTERS = 4
KAS_PLAN_TWO_PASS_HIST = 8
KAS_PLAN_SPREAD_FILL = 32
KAS_PLAN_FULL_FILL = 16
KAS_PLAN_NO_INDEX_ROWS = 64
KAS_PLAN_INDEX_ROWS = 128
KAS_PLAN_MID32 = 0x80000
KAS_PLAN_NO_MID32 = 0x100000
KAS_PLAN_TICKET_ORDER = 0x10000
KAS_PLAN_RELAX_TILES_64 = 0x20000     # KAS_PLAN_RELAX_TILES(1)
KAS_PLAN_RELAX_TILES_128 = 0x40000    # KAS_PLAN_RELAX_TILES(2)
KAS_PLAN_NO_RTN_QUOTA = 0x200000
KAS_PLAN_SPLIT_P4 = 0x400000          # first fit in kas_p4_kernel whatever the batch size (default: from 512 scenarios on)
KAS_PLAN_FILL_WITH_P4 = 0x800000      # first fit inside the fill workgroup whatever the batch size
KAS_PLAN_P4_WITH_ORDER = 0xC00000     # both: first fit as a second wavefront of the order kernel's workgroup (kas_p4_order_kernel)


def KAS_PLAN_VERIFY_SAMPLE(k: int) -> int:
    """relaxation form: k tiles per topic evaluated a second time row by row (include/kas_abi.h)"""
    return (k & 0xff) << 24

STATUS_NAMES = {
    KAS_OK: "OK",
    KAS_FAIL_UNASSIGNABLE: "FAIL_UNASSIGNABLE",
    KAS_FAIL_RF_NOT_POSITIVE: "FAIL_RF_NOT_POSITIVE",
    KAS_FAIL_RF_GT_BROKERS: "FAIL_RF_GT_BROKERS",
    KAS_FAIL_HASH_INDEX: "FAIL_HASH_INDEX",
    KAS_FAIL_RF_MISMATCH: "FAIL_RF_MISMATCH",
    KAS_SKIPPED: "SKIPPED",
    KAS_FAIL_BAD_NODES: "FAIL_BAD_NODES",
    KAS_FAIL_WATCHDOG: "FAIL_WATCHDOG",
}


class TopicDesc(C.Structure):
    _fields_ = [
        ("name_hash", C.c_int32),
        ("n_partitions", C.c_int32),
        ("cur_width", C.c_int32),
        ("rf", C.c_int32),
        ("out_width", C.c_int32),
        ("reserved", C.c_int32),
        ("cur_off", C.c_int64),
        ("out_off", C.c_int64),
        ("cur_len_off", C.c_int64),
        ("in_partitions_off", C.c_int64),
        ("part_id_off", C.c_int64),
    ]


class ScenarioDesc(C.Structure):
    _fields_ = [
        ("n_nodes", C.c_int32),
        ("topic_begin", C.c_int32),
        ("topic_count", C.c_int32),
        ("ctx_width", C.c_int32),
        ("node_off", C.c_int64),
        ("ctx_off", C.c_int64),
    ]


class TopicResult(C.Structure):
    _fields_ = [
        ("status", C.c_int32),
        ("fail_partition", C.c_int32),
        ("moved_replicas", C.c_int32),
        ("moved_partitions", C.c_int32),
    ]


class ScenarioResult(C.Structure):
    _fields_ = [
        ("status", C.c_int32),
        ("fail_topic", C.c_int32),
        ("fail_partition", C.c_int32),
        ("moved_replicas", C.c_int32),
        ("moved_partitions", C.c_int32),
        ("reserved", C.c_int32),
        ("digest", C.c_uint64),
    ]


class BatchDesc(C.Structure):
    _fields_ = [
        ("n_scenarios", C.c_int32),
        ("n_topics", C.c_int32),
        ("scenarios", C.POINTER(ScenarioDesc)),
        ("topics", C.POINTER(TopicDesc)),
        ("node_id", C.POINTER(C.c_int32)),
        ("node_rack", C.POINTER(C.c_int32)),
        ("node_pool_len", C.c_int64),
    ]


class Tables(C.Structure):
    _fields_ = [
        ("cur", C.c_void_p),
        ("out", C.c_void_p),
        ("aux", C.c_void_p),
        ("ctx", C.c_void_p),
        ("topic_results", C.c_void_p),
        ("scenario_results", C.c_void_p),
        ("cur_len", C.c_int64),
        ("out_len", C.c_int64),
        ("aux_len", C.c_int64),
        ("ctx_len", C.c_int64),
    ]


class RfResult(C.Structure):
    """kas_rf_result: what kas_resolve_replication_factor leaves (KTA:47-69 as data)."""
    _fields_ = [("status", C.c_int32), ("rf", C.c_int32), ("fail_partition", C.c_int32), ("fail_list_size", C.c_int32)]


# numpy structured dtypes with the same layout as TopicResult / ScenarioResult
import numpy as _np  # noqa: E402

TOPIC_RESULT_DTYPE = _np.dtype([
    ("status", "<i4"), ("fail_partition", "<i4"),
    ("moved_replicas", "<i4"), ("moved_partitions", "<i4")])
SCENARIO_RESULT_DTYPE = _np.dtype([
    ("status", "<i4"), ("fail_topic", "<i4"), ("fail_partition", "<i4"),
    ("moved_replicas", "<i4"), ("moved_partitions", "<i4"), ("reserved", "<i4"),
    ("digest", "<u8")])
TOPIC_DESC_DTYPE = _np.dtype([
    ("name_hash", "<i4"), ("n_partitions", "<i4"), ("cur_width", "<i4"), ("rf", "<i4"),
    ("out_width", "<i4"), ("reserved", "<i4"), ("cur_off", "<i8"), ("out_off", "<i8"),
    ("cur_len_off", "<i8"), ("in_partitions_off", "<i8"), ("part_id_off", "<i8")])
SCENARIO_DESC_DTYPE = _np.dtype([
    ("n_nodes", "<i4"), ("topic_begin", "<i4"), ("topic_count", "<i4"), ("ctx_width", "<i4"),
    ("node_off", "<i8"), ("ctx_off", "<i8")])

assert TOPIC_RESULT_DTYPE.itemsize == C.sizeof(TopicResult) == 16
assert SCENARIO_RESULT_DTYPE.itemsize == C.sizeof(ScenarioResult) == 32
assert TOPIC_DESC_DTYPE.itemsize == C.sizeof(TopicDesc) == 64
assert SCENARIO_DESC_DTYPE.itemsize == C.sizeof(ScenarioDesc) == 32

_M64 = (1 << 64) - 1


def digest_cell(topic: int, row: int, slot: int, broker: int) -> int:
    """Python twin of kas_digest_cell() in include/kas_abi.h."""
    m32 = 0xFFFFFFFF
    tlo = ((topic + 1) * 0x9E3779B1) & m32
    thi = ((topic + 1) * 0x85EBCA77) & m32
    b = ((broker & m32) ^ tlo ^ 0x7F4A7C15) & m32
    r = ((((row << 4) & m32) | ((slot & 7) << 1) | 1) ^ (thi & 0xFFFFFFFE)) & m32
    x = (b * r + ((r << 32) | b)) & _M64
    x ^= x >> 29
    return x
