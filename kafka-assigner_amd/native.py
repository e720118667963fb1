"""ctypes binding of the C ABI (include/kas_abi.h) implemented by csrc/libkas_hip.so.

There is no fallback: if the library has not been built, or no gfx950 device is visible, every
solve entry point raises.  The library is looked up in-tree only (kafka-assigner_amd/csrc/).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from . import abi
from .flatten import FlatBatch, HostOutputs, batch_desc, host_tables

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "csrc", "libkas_hip.so")

# every symbol include/kas_abi.h declares
SYMBOLS = [
    "kas_abi_version", "kas_strerror", "kas_status_string", "kas_last_error", "kas_device_count",
    "kas_ctx_create", "kas_ctx_destroy", "kas_ctx_synchronize", "kas_plan_create",
    "kas_plan_destroy", "kas_plan_algorithmic_bytes", "kas_solve_device", "kas_solve_host",
    "kas_plan_kernel_time_us", "kas_plan_phase_times_us", "kas_plan_stats", "kas_plan_set_flags",
    "kas_plan_describe", "kas_ctx_host_stats", "kas_solve_host_select", "kas_host_alloc", "kas_host_free",
    "kas_shard_range", "kas_batch_slice", "kas_solve_host_sharded", "kas_ctx_lds_lane_order", "kas_solve_host16",
    "kas_plan_create16", "kas_solve_device16", "kas_resolve_replication_factor", "kas_failure_text",
]

_LIB = None


class KasError(RuntimeError):
    def __init__(self, code: int, detail: str):
        super().__init__(f"kas error {code}: {detail}")
        self.code = code
        self.detail = detail


def load():
    """dlopen csrc/libkas_hip.so and declare prototypes.  Raises if it is not built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    lib_path = os.environ.get("KAS_HIP_LIB", LIB_PATH)     # tuning builds of the same library
    if not os.path.exists(lib_path):
        raise ImportError(
            f"{lib_path} is missing: build it with `python -m kafka_assigner_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    # PyTorch bundles its own libamdhip64.so.7 / libhsa-runtime64.so.1 with the same SONAMEs as
    # /opt/rocm's; whichever is loaded first serves the whole process, and mixing them (ours
    # first, torch second) leaves HIP without devices.  Torch owns device memory and streams in
    # this project, so its runtime must be the one: import it before the dlopen.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(lib_path)
    L.kas_abi_version.restype = C.c_int
    L.kas_strerror.restype = C.c_char_p; L.kas_strerror.argtypes = [C.c_int]
    L.kas_status_string.restype = C.c_char_p; L.kas_status_string.argtypes = [C.c_int]
    L.kas_last_error.restype = C.c_char_p
    L.kas_device_count.restype = C.c_int
    L.kas_ctx_create.restype = C.c_int; L.kas_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.kas_ctx_destroy.restype = None; L.kas_ctx_destroy.argtypes = [C.c_void_p]
    L.kas_ctx_synchronize.restype = C.c_int; L.kas_ctx_synchronize.argtypes = [C.c_void_p]
    L.kas_plan_create.restype = C.c_int
    L.kas_plan_create.argtypes = [C.c_void_p, C.POINTER(abi.BatchDesc), C.POINTER(C.c_void_p)]
    L.kas_plan_destroy.restype = None; L.kas_plan_destroy.argtypes = [C.c_void_p]
    L.kas_plan_algorithmic_bytes.restype = C.c_int64; L.kas_plan_algorithmic_bytes.argtypes = [C.c_void_p]
    L.kas_solve_device.restype = C.c_int
    L.kas_solve_device.argtypes = [C.c_void_p, C.POINTER(abi.Tables), C.c_void_p]
    L.kas_solve_host.restype = C.c_int
    L.kas_solve_host.argtypes = [C.c_void_p, C.POINTER(abi.BatchDesc), C.POINTER(abi.Tables)]
    L.kas_solve_host_select.restype = C.c_int
    L.kas_solve_host_select.argtypes = [C.c_void_p, C.POINTER(abi.BatchDesc), C.POINTER(abi.Tables),
                                        C.POINTER(C.c_int32), C.c_int32]
    L.kas_solve_host16.restype = C.c_int         # (kas_tables16 has kas_tables' layout: abi.Tables holds untyped pointers)
    L.kas_solve_host16.argtypes = [C.c_void_p, C.POINTER(abi.BatchDesc), C.POINTER(abi.Tables),
                                   C.POINTER(C.c_int32), C.c_int32]
    L.kas_plan_create16.restype = C.c_int
    L.kas_plan_create16.argtypes = [C.c_void_p, C.POINTER(abi.BatchDesc), C.POINTER(C.c_void_p)]
    L.kas_solve_device16.restype = C.c_int
    L.kas_solve_device16.argtypes = [C.c_void_p, C.POINTER(abi.Tables), C.c_void_p]
    L.kas_solve_host_sharded.restype = C.c_int
    L.kas_solve_host_sharded.argtypes = [C.POINTER(C.c_void_p), C.c_int32, C.POINTER(abi.BatchDesc), C.POINTER(abi.Tables)]
    L.kas_host_alloc.restype = C.c_int; L.kas_host_alloc.argtypes = [C.c_int64, C.POINTER(C.c_void_p)]
    L.kas_host_free.restype = None; L.kas_host_free.argtypes = [C.c_void_p]
    L.kas_shard_range.restype = None
    L.kas_shard_range.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.kas_batch_slice.restype = C.c_int
    L.kas_batch_slice.argtypes = [C.POINTER(abi.BatchDesc), C.c_int64, C.c_int64, C.POINTER(abi.ScenarioDesc),
                                  C.POINTER(abi.BatchDesc), C.POINTER(abi.Tables), C.POINTER(abi.Tables)]
    L.kas_plan_kernel_time_us.restype = C.c_int
    L.kas_plan_kernel_time_us.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    L.kas_plan_phase_times_us.restype = C.c_int
    L.kas_plan_phase_times_us.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                          C.POINTER(C.c_int)]
    L.kas_plan_set_flags.restype = C.c_int
    L.kas_plan_set_flags.argtypes = [C.c_void_p, C.c_uint32]
    L.kas_ctx_host_stats.restype = C.c_int
    L.kas_ctx_host_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.kas_plan_describe.restype = C.c_int
    L.kas_plan_describe.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    L.kas_plan_stats.restype = C.c_int
    L.kas_plan_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.c_int64]
    L.kas_resolve_replication_factor.restype = C.c_int
    L.kas_resolve_replication_factor.argtypes = [C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_int32,
                                                 C.POINTER(abi.RfResult)]
    L.kas_failure_text.restype = C.c_int
    L.kas_failure_text.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_char_p, C.c_int]
    if L.kas_abi_version() != abi.KAS_ABI_VERSION:
        raise ImportError("libkas_hip.so ABI version mismatch")
    _LIB = L
    return L


def _check(rc: int):
    if rc != 0:
        L = load()
        raise KasError(rc, (L.kas_last_error() or b"").decode() or L.kas_strerror(rc).decode())


def resolve_replication_factor(partition_ids, list_sizes, desired_rf: int, n_brokers: int) -> "abi.RfResult":
    """kas_resolve_replication_factor (KTA:47-69; host arithmetic, needs no device): entries in the caller's map order."""
    pid = np.ascontiguousarray(partition_ids, dtype=np.int32)
    ls = np.ascontiguousarray(list_sizes, dtype=np.int32)
    assert pid.shape == ls.shape and pid.ndim == 1
    res = abi.RfResult()
    _check(load().kas_resolve_replication_factor(pid.ctypes.data_as(C.POINTER(C.c_int32)), ls.ctypes.data_as(C.POINTER(C.c_int32)),
                                                 int(pid.size), int(desired_rf), int(n_brokers), C.byref(res)))
    return res


def failure_text(topic, status: int, fail_partition: int = -1, rf: int = -1, list_size: int = -1) -> str:
    """kas_failure_text: the message of the exception the reference throws for this status ('' where it has none)."""
    buf = C.create_string_buffer(1024)
    n = load().kas_failure_text(None if topic is None else str(topic).encode("utf-8"), int(status), int(fail_partition), int(rf),
                                int(list_size), buf, 1024)
    return buf.raw[:n].decode("utf-8")


class DeviceContext:
    """kas_ctx: one HIP device + stream."""

    def __init__(self, device: int = 0):
        self._lib = load()
        self._h = C.c_void_p()
        _check(self._lib.kas_ctx_create(device, C.byref(self._h)))
        self.device = device

    def synchronize(self):
        _check(self._lib.kas_ctx_synchronize(self._h))

    def lds_lane_order(self):
        """(state, lane-operations checked): kas_ctx_lds_lane_order — 1 the LDS served every checked atomic-with-return in
        lane order, 0 violated, -1 the self-test could not run, -2 switched off (KAS_NO_LANE_ORDER=1)."""
        n = C.c_int64()
        self._lib.kas_ctx_lds_lane_order.restype = C.c_int
        self._lib.kas_ctx_lds_lane_order.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        return int(self._lib.kas_ctx_lds_lane_order(self._h, C.byref(n))), n.value

    def host_stats(self):
        """(kas_solve_host calls, calls served by a cached plan, device allocations made)."""
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        _check(self._lib.kas_ctx_host_stats(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def close(self):
        if self._h:
            self._lib.kas_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Plan:
    """kas_plan: validated batch shape with descriptors and node tables resident in HBM."""

    def __init__(self, ctx: DeviceContext, fb: FlatBatch, cells16: bool = False):
        """cells16: kas_plan_create16 — the plan's solves read and write uint16 node-index cells (solve_device with
        pointers to uint16 pools; fb.node_id is not read)."""
        self._lib = load()
        self._ctx = ctx
        self._fb = fb                      # keeps the host descriptor arrays alive
        self.cells16 = bool(cells16)
        bd = batch_desc(fb)
        self._h = C.c_void_p()
        if cells16:
            bd.node_id = None
            _check(self._lib.kas_plan_create16(ctx._h, C.byref(bd), C.byref(self._h)))
        else:
            _check(self._lib.kas_plan_create(ctx._h, C.byref(bd), C.byref(self._h)))

    @property
    def algorithmic_bytes(self) -> int:
        return int(self._lib.kas_plan_algorithmic_bytes(self._h))

    def solve_device(self, cur: int, out: int, topic_results: int, scenario_results: int,
                     aux: int = 0, ctx: int = 0, stream: int = 0):
        """Enqueue one solve; every argument is a raw device pointer (int)."""
        t = abi.Tables()
        t.cur = cur or None; t.out = out or None; t.aux = aux or None; t.ctx = ctx or None
        t.topic_results = topic_results or None; t.scenario_results = scenario_results or None
        fn = self._lib.kas_solve_device16 if self.cells16 else self._lib.kas_solve_device
        _check(fn(self._h, C.byref(t), C.c_void_p(stream) if stream else None))

    def set_flags(self, flags: int):
        _check(self._lib.kas_plan_set_flags(self._h, flags))

    def describe(self) -> str:
        """The kernels a solve of this plan launches (template arguments, grids, LDS)."""
        buf = C.create_string_buffer(512)
        n = self._lib.kas_plan_describe(self._h, buf, 512)
        if n < 0:
            _check(n)
        return buf.value.decode()

    def stats(self) -> np.ndarray:
        """Per-scenario device counters of the last solve: int64 [S, 16] =
        (setup, P2, P3+P4, P5 time in 10 ns ticks; P4 windows, P4 node steps, P5 rounds,
        P2 overflow tiles)."""
        n = self._fb.n_scenarios
        a = np.zeros((n, 16), dtype=np.int64)
        _check(self._lib.kas_plan_stats(self._h, a.ctypes.data_as(C.POINTER(C.c_int64)), a.size))
        return a

    def kernel_time_us(self):
        avg = C.c_double(); n = C.c_int()
        _check(self._lib.kas_plan_kernel_time_us(self._h, C.byref(avg), C.byref(n)))
        return avg.value, n.value

    def phase_times_us(self):
        """(fill kernel avg us, order kernel avg us, launches) since the last call."""
        f = C.c_double(); o = C.c_double(); n = C.c_int()
        _check(self._lib.kas_plan_phase_times_us(self._h, C.byref(f), C.byref(o), C.byref(n)))
        return f.value, o.value, n.value

    def close(self):
        if self._h:
            self._lib.kas_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_DEFAULT_CTX: Optional[DeviceContext] = None


def default_context() -> DeviceContext:
    global _DEFAULT_CTX
    if _DEFAULT_CTX is None:
        _DEFAULT_CTX = DeviceContext(int(os.environ.get("LOCAL_RANK", "0")) if
                                     load().kas_device_count() > 1 else 0)
    return _DEFAULT_CTX


def solve_host(fb: FlatBatch, ctx: Optional[DeviceContext] = None) -> HostOutputs:
    """kas_solve_host: copy in, solve on the GPU, copy out (blocking)."""
    L = load()
    ctx = ctx or default_context()
    bd = batch_desc(fb)
    t, ho = host_tables(fb)
    _check(L.kas_solve_host(ctx._h, C.byref(bd), C.byref(t)))
    return ho


def selected_out_len(fb: FlatBatch, select) -> int:
    """int32 cells kas_solve_host_select returns for the scenarios in `select` (their rows, packed)."""
    n = 0
    for s in select:
        if not 0 <= int(s) < fb.n_scenarios:
            continue                       # (the library refuses the call: KAS_E_INVALID_ARG)
        sd = fb.scen[int(s)]
        t = fb.topics[int(sd["topic_begin"]): int(sd["topic_begin"]) + int(sd["topic_count"])]
        n += int((t["n_partitions"].astype(np.int64) * t["out_width"]).sum())
    return n


def solve_host_select(fb: FlatBatch, select, ctx: Optional[DeviceContext] = None, ho: Optional[HostOutputs] = None,
                      tables=None) -> HostOutputs:
    """kas_solve_host_select: every scenario is solved and reports its records, rows come back only
    for the scenarios in `select`, packed in that order (the what-if form: one assignment printed)."""
    L = load()
    ctx = ctx or default_context()
    bd = batch_desc(fb)
    sel = np.ascontiguousarray(select, dtype=np.int32)
    if tables is None:
        tables, ho = host_tables(fb, out_len=selected_out_len(fb, sel))
    _check(L.kas_solve_host_select(ctx._h, C.byref(bd), C.byref(tables),
                                   sel.ctypes.data_as(C.POINTER(C.c_int32)), int(sel.size)))
    return ho


def solve_host16(fb: FlatBatch, ctx: Optional[DeviceContext] = None, select=None, cur16=None, tables=None,
                 ho: Optional[HostOutputs] = None) -> HostOutputs:
    """kas_solve_host16: the host call with 16-bit cells (node indices; flatten.to_cells16 / cells16_to_ids).
    HostOutputs.out is the uint16 out pool; select = scenario indices whose rows come back packed (None: every row)."""
    from .flatten import host_tables16, to_cells16
    L = load()
    ctx = ctx or default_context()
    bd = batch_desc(fb)
    bd.node_id = None                      # (not read: node i has id i)
    sel = None if select is None else np.ascontiguousarray(select, dtype=np.int32)
    if tables is None:
        cur16 = to_cells16(fb) if cur16 is None else cur16
        tables, ho = host_tables16(fb, cur16, out_len=None if sel is None else selected_out_len(fb, sel))
    _check(L.kas_solve_host16(ctx._h, C.byref(bd), C.byref(tables),
                              None if sel is None else sel.ctypes.data_as(C.POINTER(C.c_int32)),
                              -1 if sel is None else int(sel.size)))
    return ho


def solve_host_sharded(fb: FlatBatch, ctxs) -> HostOutputs:
    """kas_solve_host_sharded: contiguous scenario ranges over several contexts (devices), one host
    thread each; the caller's host arrays are the gather."""
    L = load()
    bd = batch_desc(fb)
    t, ho = host_tables(fb)
    arr = (C.c_void_p * len(ctxs))(*[c._h for c in ctxs])
    _check(L.kas_solve_host_sharded(arr, len(ctxs), C.byref(bd), C.byref(t)))
    return ho


class PinnedArray:
    """int32 (or uint16: 16-bit cells) numpy array over kas_host_alloc memory (pinned: kas_solve_host moves it by DMA
    without staging)."""

    def __init__(self, n: int, dtype=np.int32):
        self._lib = load()
        self._p = C.c_void_p()
        item = np.dtype(dtype).itemsize
        _check(self._lib.kas_host_alloc(item * max(int(n), 1), C.byref(self._p)))
        buf = (C.c_uint8 * (item * max(int(n), 1))).from_address(self._p.value)
        self.array = np.frombuffer(buf, dtype=dtype)[:int(n)]

    def close(self):
        if self._p:
            self.array = None
            self._lib.kas_host_free(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def shard_range(total: int, rank: int, world: int):
    """kas_shard_range (the C mirror of sharding.shard_range)."""
    lo, hi = C.c_int64(), C.c_int64()
    load().kas_shard_range(total, rank, world, C.byref(lo), C.byref(hi))
    return lo.value, hi.value


def solve_host_with_flags(fb: FlatBatch, flags: int, ctx: Optional[DeviceContext] = None) -> HostOutputs:
    """Same as solve_host but through plan + device tables so plan flags can be set."""
    import torch
    ctx = ctx or default_context()
    dev = torch.device("cuda", ctx.device)
    plan = Plan(ctx, fb)
    plan.set_flags(flags)
    _, ho = host_tables(fb)
    d_cur = torch.from_numpy(fb.cur).to(dev)
    d_aux = torch.from_numpy(fb.aux).to(dev) if fb.aux.size else None
    d_ctx = torch.from_numpy(ho.ctx).to(dev) if ho.ctx.size else None
    d_out = torch.full((max(fb.out_len, 1),), -2, dtype=torch.int32, device=dev)
    d_tr = torch.zeros(max(fb.n_topics, 1) * 16, dtype=torch.uint8, device=dev)
    d_sr = torch.zeros(max(fb.n_scenarios, 1) * 32, dtype=torch.uint8, device=dev)
    st = torch.cuda.Stream(dev)
    st.wait_stream(torch.cuda.current_stream(dev))
    plan.solve_device(d_cur.data_ptr(), d_out.data_ptr(), d_tr.data_ptr(), d_sr.data_ptr(),
                      aux=d_aux.data_ptr() if d_aux is not None else 0,
                      ctx=d_ctx.data_ptr() if d_ctx is not None else 0, stream=st.cuda_stream)
    st.synchronize()
    ho.out = d_out.cpu().numpy()
    ho.topic_results = d_tr.cpu().numpy().view(abi.TOPIC_RESULT_DTYPE)
    ho.scenario_results = d_sr.cpu().numpy().view(abi.SCENARIO_RESULT_DTYPE)
    if d_ctx is not None:
        ho.ctx = d_ctx.cpu().numpy()
    plan.close()
    return ho


def solve_device16_with_flags(fb: FlatBatch, flags: int = 0, ctx: Optional[DeviceContext] = None, cur16=None) -> HostOutputs:
    """kas_plan_create16 + kas_solve_device16: the batch solved on uint16 node-index cells resident in HBM (HostOutputs.out is
    the uint16 out pool).  Raises KasError(KAS_E_UNSUPPORTED) for batches the 16-bit kernels do not take."""
    import torch
    from .flatten import to_cells16
    ctx = ctx or default_context()
    dev = torch.device("cuda", ctx.device)
    plan = Plan(ctx, fb, cells16=True)
    try:
        if flags:
            plan.set_flags(flags)
        _, ho = host_tables(fb)
        c16 = to_cells16(fb) if cur16 is None else cur16
        d_cur = torch.from_numpy(c16.view(np.int16)).to(dev)
        d_aux = torch.from_numpy(fb.aux).to(dev) if fb.aux.size else None
        d_ctx = torch.from_numpy(ho.ctx).to(dev) if ho.ctx.size else None
        d_out = torch.full((max(fb.out_len, 1),), -2, dtype=torch.int16, device=dev)
        d_tr = torch.zeros(max(fb.n_topics, 1) * 16, dtype=torch.uint8, device=dev)
        d_sr = torch.zeros(max(fb.n_scenarios, 1) * 32, dtype=torch.uint8, device=dev)
        st = torch.cuda.Stream(dev)
        st.wait_stream(torch.cuda.current_stream(dev))
        plan.solve_device(d_cur.data_ptr(), d_out.data_ptr(), d_tr.data_ptr(), d_sr.data_ptr(),
                          aux=d_aux.data_ptr() if d_aux is not None else 0,
                          ctx=d_ctx.data_ptr() if d_ctx is not None else 0, stream=st.cuda_stream)
        st.synchronize()
        ho.out = d_out.cpu().numpy().view(np.uint16)
        ho.topic_results = d_tr.cpu().numpy().view(abi.TOPIC_RESULT_DTYPE)
        ho.scenario_results = d_sr.cpu().numpy().view(abi.SCENARIO_RESULT_DTYPE)
        if d_ctx is not None:
            ho.ctx = d_ctx.cpu().numpy()
        ho.describe = plan.describe()
    finally:
        plan.close()
    return ho
