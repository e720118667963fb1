"""ctypes loader for the CPU oracle (oracle/kas_oracle.c).  Test infrastructure only."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from kafka_assigner_amd import abi
from kafka_assigner_amd.flatten import FlatBatch, HostOutputs, batch_desc, host_tables

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_LIB = None
_FAST = None


def _build(name: str, src_name: str, force: bool = False) -> str:
    so = os.path.join(ORACLE_DIR, name)
    deps = [os.path.join(ORACLE_DIR, src_name), os.path.join(ORACLE_DIR, "kas_batch_loop.h"),
            os.path.join(ROOT, "include", "kas_abi.h")]
    stale = not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(d) for d in deps)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "-B", name])
    return so


def build_oracle(force: bool = False) -> str:
    return _build("libkas_oracle.so", "kas_oracle.c", force)


def build_cpu_fast(force: bool = False) -> str:
    return _build("libkas_cpu_fast.so", "kas_cpu_fast.c", force)


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build_oracle())
        L.kas_oracle_solve_batch.restype = C.c_int
        L.kas_oracle_solve_batch.argtypes = [C.POINTER(abi.BatchDesc), C.POINTER(abi.Tables)]
        L.kas_oracle_solve_batch_mt.restype = C.c_int
        L.kas_oracle_solve_batch_mt.argtypes = [C.POINTER(abi.BatchDesc), C.POINTER(abi.Tables), C.c_int]
        L.kas_oracle_host_threads.restype = C.c_int
        L.kas_oracle_abi_version.restype = C.c_int
        assert L.kas_oracle_abi_version() == abi.KAS_ABI_VERSION
        _LIB = L
    return _LIB


def fast_lib():
    """oracle/kas_cpu_fast.c: the flat-array CPU baseline (B2); same results as the oracle."""
    global _FAST
    if _FAST is None:
        L = C.CDLL(build_cpu_fast())
        L.kas_cpu_fast_solve_batch.restype = C.c_int
        L.kas_cpu_fast_solve_batch.argtypes = [C.POINTER(abi.BatchDesc), C.POINTER(abi.Tables)]
        L.kas_cpu_fast_solve_batch_mt.restype = C.c_int
        L.kas_cpu_fast_solve_batch_mt.argtypes = [C.POINTER(abi.BatchDesc), C.POINTER(abi.Tables), C.c_int]
        L.kas_cpu_fast_host_threads.restype = C.c_int
        L.kas_cpu_fast_abi_version.restype = C.c_int
        assert L.kas_cpu_fast_abi_version() == abi.KAS_ABI_VERSION
        _FAST = L
    return _FAST


def host_threads() -> int:
    """Hardware threads of this host (== std::thread::hardware_concurrency())."""
    return int(lib().kas_oracle_host_threads())


def _solve(fn_st, fn_mt, fb: FlatBatch, threads: int, what: str, into=None) -> HostOutputs:
    bd = batch_desc(fb)
    t, ho = into if into is not None else host_tables(fb)
    if threads == 1:
        rc = fn_st(C.byref(bd), C.byref(t))
        if rc != 0:
            raise RuntimeError(f"{what} returned {rc}")
        ho.threads_used = 1
    else:
        rc = fn_mt(C.byref(bd), C.byref(t), int(threads))
        if rc < 0:
            raise RuntimeError(f"{what} (threaded) returned {rc}")
        ho.threads_used = rc
    return ho


def oracle_solve(fb: FlatBatch, threads: int = 1, into=None) -> HostOutputs:
    """Solve a flattened batch with the CPU oracle; same semantics as kas_solve_host.
    threads != 1: scenario-parallel inside the one C call (0 = every hardware thread).
    into = (kas_tables, HostOutputs) from flatten.host_tables: solve into these buffers again (timing
    loops: no allocation and no first touch of fresh pages inside the timed call)."""
    L = lib()
    return _solve(L.kas_oracle_solve_batch, L.kas_oracle_solve_batch_mt, fb, threads, "kas_oracle_solve_batch", into)


def cpu_fast_solve(fb: FlatBatch, threads: int = 1, into=None) -> HostOutputs:
    """The same batch by the flat-array CPU baseline (oracle/kas_cpu_fast.c)."""
    L = fast_lib()
    return _solve(L.kas_cpu_fast_solve_batch, L.kas_cpu_fast_solve_batch_mt, fb, threads, "kas_cpu_fast_solve_batch", into)


def delivered_parallelism(n_threads: int = 0, iters: int = 200_000_000) -> dict:
    """What the host really gives this process: the same register-only loop on 1 thread and on n_threads
    (0 = every hardware thread); cores = t1 * n / tn.  A container with a CPU quota reports all hardware
    threads of the machine and delivers a fraction of them."""
    import os
    L = fast_lib()
    L.kas_cpu_fast_parallelism_probe.restype = C.c_double
    L.kas_cpu_fast_parallelism_probe.argtypes = [C.c_int, C.c_uint64]
    n = n_threads if n_threads > 0 else host_threads()
    t1 = min(L.kas_cpu_fast_parallelism_probe(1, iters) for _ in range(3))      # (best of three each: noisy neighbours)
    tn = min(L.kas_cpu_fast_parallelism_probe(n, iters) for _ in range(3))
    info = {"threads": n, "seconds_1_thread": t1, "seconds_n_threads": tn,
            "cores_delivered": (t1 * n / tn) if tn > 0 else None}
    try:
        info["sched_affinity_cpus"] = len(os.sched_getaffinity(0))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            info["cgroup_" + os.path.basename(path)] = open(path).read().strip()
        except Exception:
            pass
    return info
