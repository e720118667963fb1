"""The wide order kernel's LDS arithmetic (csrc/kas_plan_math.h: kas_order_wide_lds_core / has_front),
restated for tests that need to know which side of a limit a shape is on."""
RING_SLOTS, HOT, BULK = 8, 2, 2
LDS_LIMIT = 160 * 1024


def _a16(v):
    return (v + 15) & ~15


def wide_lds_core(n_max):
    n = max(n_max, 1)
    return _a16(2 * _a16(8 * (n + 1)) + _a16(2 * (n + 1)) + RING_SLOTS * 64 * 32 + 2 * RING_SLOTS * 64 * 2 + 16
                + 256 * HOT * (1 + BULK) + 16)


def wide_has_front(n_max):
    return wide_lds_core(n_max) + _a16(4 * (max(n_max, 1) + 1)) <= LDS_LIMIT
