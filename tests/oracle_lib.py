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


def build_oracle(force: bool = False) -> str:
    so = os.path.join(ORACLE_DIR, "libkas_oracle.so")
    src = os.path.join(ORACLE_DIR, "kas_oracle.c")
    hdr = os.path.join(ROOT, "include", "kas_abi.h")
    stale = (not os.path.exists(so)
             or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)))
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "-B", "libkas_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build_oracle())
        L.kas_oracle_solve_batch.restype = C.c_int
        L.kas_oracle_solve_batch.argtypes = [C.POINTER(abi.BatchDesc), C.POINTER(abi.Tables)]
        L.kas_oracle_abi_version.restype = C.c_int
        assert L.kas_oracle_abi_version() == abi.KAS_ABI_VERSION
        _LIB = L
    return _LIB


def oracle_solve(fb: FlatBatch) -> HostOutputs:
    """Solve a flattened batch with the CPU oracle; same semantics as kas_solve_host."""
    bd = batch_desc(fb)
    t, ho = host_tables(fb)
    rc = lib().kas_oracle_solve_batch(C.byref(bd), C.byref(t))
    if rc != 0:
        raise RuntimeError(f"kas_oracle_solve_batch returned {rc}")
    return ho
