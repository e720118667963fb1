"""Loader for the CPU-fiber emulation of the kernel source (tests/emu).  Test infrastructure."""
from __future__ import annotations

import ctypes as C
import os
import shlex
import subprocess
import time

from kafka_assigner_amd import abi
from kafka_assigner_amd.flatten import FlatBatch, HostOutputs, batch_desc, host_tables

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "emu")
# KAS_EMU_CSRC=<dir>: the emulator suite against a patched copy of the kernel sources (experiments/*.patch applied to a
# scratch tree); its shared objects then go to KAS_EMU_OUT (default <dir>/emu_build), never next to the product's
CSRC = os.environ.get("KAS_EMU_CSRC") or os.path.join(ROOT, "kafka-assigner_amd", "csrc")
OUT_DIR = (os.environ.get("KAS_EMU_OUT") or os.path.join(CSRC, "emu_build")) if os.environ.get("KAS_EMU_CSRC") else EMU_DIR
_LIB = None


# Variants the CPU suite loads (tests/test_emu_watchdog.py): registered here so that the first emulator
# build of a fresh checkout compiles all of them side by side (one ~35 s compile each; in turn they
# were a third of the suite's wall clock).
TEST_VARIANTS = {
    "bounded": ["-DKAS_SPIN_BOUND=200000"],
    "stalled": ["-DKAS_SPIN_BOUND=1500", "-DKAS_TEST_STALL_AFTER=2"],
    "stalled_sparse": ["-DKAS_SPIN_BOUND=65536", "-DKAS_SPIN_CHECK=4096", "-DKAS_TEST_STALL_AFTER=2"],
    "rtn_descending": ["-DKAS_EMU_RTN_DESCENDING"],
}


def _deps():
    return [os.path.join(EMU_DIR, "emu_driver.cpp"), os.path.join(EMU_DIR, "kas_wave.h"),
            os.path.join(CSRC, "kas_solver_body.h"), os.path.join(CSRC, "kas_order_wide.h"),
            os.path.join(CSRC, "kas_order_relax.h"),
            os.path.join(CSRC, "kas_plan_math.h"), os.path.join(ROOT, "include", "kas_abi.h")]


def _stale(so: str) -> bool:
    return not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(d) for d in _deps())


def _compile_command(so: str, flags, warn=False):
    return ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", *(["-Wall", "-Wextra"] if warn else []),
            "-Wno-unused-parameter", "-Wno-unknown-pragmas", *flags, "-I" + os.path.join(ROOT, "tests"),
            "-I" + CSRC, "-I" + os.path.join(ROOT, "include"), "-o", so, _deps()[0]]


def _builder_gone(marker: str) -> bool:
    """The compile that wrote `marker` is no longer running (a killed test run leaves markers behind)."""
    try:
        txt = open(marker).read().strip()
        age = time.time() - os.path.getmtime(marker)
    except OSError:
        return False                                        # the marker has just gone: the build finished
    if not txt:
        return age > 30                                     # created, pid not written yet
    try:
        os.kill(int(txt), 0)
        return False
    except (ProcessLookupError, ValueError):
        return True
    except PermissionError:
        return False


def _start_build(so: str, flags, warn=False) -> None:
    """Start the compile of a stale shared object in a process of its own and return.  It writes to a
    temporary name and renames (a half-written file is never loaded); `so.building` (holding the
    compiling shell's pid) exists while it runs, so that other processes (the tests that spawn
    interpreters) wait for it instead of compiling the same file again."""
    if not _stale(so):
        return
    marker = so + ".building"
    try:
        os.close(os.open(marker, os.O_CREAT | os.O_EXCL | os.O_WRONLY))
    except FileExistsError:
        if not _builder_gone(marker):
            return                                          # somebody is on it
        try:
            os.remove(marker)                               # left behind by a killed run: take over
        except OSError:
            pass
        return _start_build(so, flags, warn)
    tmp = so + ".tmp%d" % os.getpid()
    cmd = " ".join(shlex.quote(c) for c in _compile_command(tmp, flags, warn))
    proc = subprocess.Popen(["/bin/sh", "-c", f"{cmd} && mv -f {shlex.quote(tmp)} {shlex.quote(so)}; rm -f {shlex.quote(marker)}"])
    try:
        with open(marker, "r+") as f:
            f.write(str(proc.pid))
    except OSError:
        pass                                                # (already finished and removed)


def _finish_build(so: str, flags=(), warn=False) -> str:
    marker = so + ".building"
    t0 = time.time()
    while os.path.exists(marker):
        if _builder_gone(marker):
            _start_build(so, list(flags), warn)             # takes the marker over
        if time.time() - t0 > 900:
            raise TimeoutError(f"{marker} has been there for 15 minutes")
        time.sleep(0.2)
    if _stale(so):
        raise RuntimeError("the emulator did not compile: " + " ".join(_compile_command(so, list(flags))))
    return so


def _variant_path(name: str) -> str:
    return os.path.join(OUT_DIR, f"libkas_emu_{name}.so")


def _start_all_stale() -> None:
    os.makedirs(OUT_DIR, exist_ok=True)
    _start_build(os.path.join(OUT_DIR, "libkas_emu.so"), [], warn=True)
    for name, flags in TEST_VARIANTS.items():
        _start_build(_variant_path(name), flags)


def build_emu() -> str:
    _start_all_stale()
    return _finish_build(os.path.join(OUT_DIR, "libkas_emu.so"), [], warn=True)


def build_emu_variant(name: str, flags) -> str:
    """The emulator compiled with extra -D flags (debug-build macros of the kernel source)."""
    assert name not in TEST_VARIANTS or list(flags) == TEST_VARIANTS[name], "a registered variant with other flags"
    if name in TEST_VARIANTS:
        _start_all_stale()
    so = _variant_path(name)
    _start_build(so, list(flags))
    return _finish_build(so, flags)


def variant_solver(name: str, flags):
    """emu_solve bound to an emulator variant (own shared object, loaded side by side)."""
    L = C.CDLL(build_emu_variant(name, flags))
    L.kas_emu_solve_batch.restype = C.c_int
    L.kas_emu_solve_batch.argtypes = [C.POINTER(abi.BatchDesc), C.POINTER(abi.Tables),
                                      C.c_uint, C.c_char_p, C.c_int]

    def solve(fb: FlatBatch, flags: int = 0, p4_by_batch_size: bool = False) -> HostOutputs:
        bd = batch_desc(fb)
        t, ho = host_tables(fb)
        err = C.create_string_buffer(512)
        rc = L.kas_emu_solve_batch(C.byref(bd), C.byref(t), _p4_form(flags, p4_by_batch_size), err, 512)
        if rc != 0:
            raise RuntimeError(f"kas_emu_solve_batch rc={rc}: {err.value.decode()}")
        return ho
    return solve


def variant_solver16(name: str, flags):
    """emu_solve16 bound to an emulator variant: (fb, plan flags) -> HostOutputs with the uint16 out pool"""
    from kafka_assigner_amd.flatten import host_tables16, to_cells16
    L = C.CDLL(build_emu_variant(name, flags))
    L.kas_emu_solve_batch16.restype = C.c_int
    L.kas_emu_solve_batch16.argtypes = [C.POINTER(abi.BatchDesc), C.POINTER(abi.Tables), C.c_uint, C.c_char_p, C.c_int]

    def solve(fb: FlatBatch, flags: int = 0) -> HostOutputs:
        bd = batch_desc(fb)
        bd.node_id = None
        c16 = to_cells16(fb)                                # (kept alive: t.cur is its raw address)
        t, ho = host_tables16(fb, c16)
        err = C.create_string_buffer(512)
        rc = L.kas_emu_solve_batch16(C.byref(bd), C.byref(t), _p4_form(flags, False), err, 512)
        del c16
        if rc != 0:
            raise RuntimeError(f"kas_emu_solve_batch16 rc={rc}: {err.value.decode()}")
        return ho
    return solve


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build_emu())
        L.kas_emu_solve_batch.restype = C.c_int
        L.kas_emu_solve_batch.argtypes = [C.POINTER(abi.BatchDesc), C.POINTER(abi.Tables),
                                          C.c_uint, C.c_char_p, C.c_int]
        L.kas_emu_last_queue_rows.restype = C.c_long
        _LIB = L
    return _LIB


def last_fused() -> bool:
    """The last emu_solve ran the rack-diverse fill with per-chunk histograms (no chunk-count pass)."""
    L = lib()
    L.kas_emu_last_fused.restype = C.c_int
    return bool(L.kas_emu_last_fused())


def last_queue_rows() -> int:
    """Rows the ticket-form solver decided inside queues during the last emu_solve."""
    return int(lib().kas_emu_last_queue_rows())


def _p4_form(flags: int, by_batch_size: bool) -> int:
    """The product runs first fit in kas_p4_kernel from 512 scenarios a batch on (kas_split_p4) — the headline's form, and
    the emulator's batches are small: unless a test names a form (FILL_WITH_P4 / SPLIT_P4) or asks for the product's own
    choice (p4_by_batch_size=True), the emulator runs kas_p4_kernel; the form inside the fill workgroup is covered where
    FILL_WITH_P4 is passed (test_emu_parity.py)."""
    if by_batch_size or (flags & (FILL_WITH_P4 | SPLIT_P4)):
        return flags
    return flags | SPLIT_P4


def emu_solve(fb: FlatBatch, flags: int = 0, p4_by_batch_size: bool = False) -> HostOutputs:
    bd = batch_desc(fb)
    t, ho = host_tables(fb)
    err = C.create_string_buffer(512)
    rc = lib().kas_emu_solve_batch(C.byref(bd), C.byref(t), _p4_form(flags, p4_by_batch_size), err, 512)
    if rc != 0:
        raise RuntimeError(f"kas_emu_solve_batch rc={rc}: {err.value.decode()}")
    return ho


def emu_solve16(fb: FlatBatch, flags: int = 0, p4_by_batch_size: bool = False, cur16=None) -> HostOutputs:
    """kas_plan_create16 + kas_solve_device16 on the emulator: cur / out as uint16 node-index pools (HostOutputs.out is the
    uint16 out pool).  Raises RuntimeError with the library's code for batches the 16-bit kernels do not take."""
    from kafka_assigner_amd.flatten import host_tables16, to_cells16
    L = lib()
    L.kas_emu_solve_batch16.restype = C.c_int
    L.kas_emu_solve_batch16.argtypes = [C.POINTER(abi.BatchDesc), C.POINTER(abi.Tables), C.c_uint, C.c_char_p, C.c_int]
    bd = batch_desc(fb)
    bd.node_id = None
    c16 = to_cells16(fb) if cur16 is None else cur16
    t, ho = host_tables16(fb, c16)
    err = C.create_string_buffer(512)
    rc = L.kas_emu_solve_batch16(C.byref(bd), C.byref(t), _p4_form(flags, p4_by_batch_size), err, 512)
    if rc != 0:
        raise RuntimeError(f"kas_emu_solve_batch16 rc={rc}: {err.value.decode()}")
    return ho


def last_index_rows() -> int:
    """Topics whose fill took the index rows in the last emu_solve (KAS_FLAG_INDEX_ROWS: pass A leaves the rows' node indices
    where the mid rows go, pass B streams those)."""
    L = lib()
    L.kas_emu_last_index_rows.restype = C.c_long
    return int(L.kas_emu_last_index_rows())


def last_mid32() -> int:
    """1 when the last emu_solve moved its mid rows as one dword each (KAS_FLAG_MID32: holders sorted, 11 bits each)."""
    L = lib()
    L.kas_emu_last_mid32.restype = C.c_long
    return int(L.kas_emu_last_mid32())


def last_relax_quad() -> int:
    """1 when the last emu_solve ran the relaxation form over quad tiles (KAS_PLAN_RELAX_TILES(3) on dword mid rows)."""
    L = lib()
    L.kas_emu_last_relax_quad.restype = C.c_long
    return int(L.kas_emu_last_relax_quad())


def last_slim_fill() -> int:
    """Scenarios the slim fill kernel (kas_fill_slim_kernel) solved itself in the last emu_solve: 0 when it was not launched,
    fewer than the batch's scenarios when it handed some back to kas_fill_kernel."""
    L = lib()
    L.kas_emu_last_slim_fill.restype = C.c_long
    return int(L.kas_emu_last_slim_fill())


def last_handback() -> int:
    """What the last emu_solve's by-rank launch of kas_fill_kernel (behind the slim kernel / the spread fill) left in
    KasLaunch::handback: the number of scenarios handed back; -1 when the solve had no such launch."""
    L = lib()
    L.kas_emu_last_handback.restype = C.c_int
    return int(L.kas_emu_last_handback())


def last_p4_order() -> int:
    """1: the last emu_solve ran first fit inside the order kernel's workgroup (kas_p4_order_kernel: KAS_PLAN_SPLIT_P4 | KAS_PLAN_FILL_WITH_P4)"""
    L = lib()
    L.kas_emu_last_p4_order.restype = C.c_int
    return int(L.kas_emu_last_p4_order())


P4_WITH_ORDER = 0x400000 | 0x800000   # KAS_PLAN_P4_WITH_ORDER: both first-fit switches = first fit as a second wavefront of the order kernel's workgroup


def last_relax_idl() -> int:
    """1: the relaxation form of the last emu_solve read the final rows' broker ids from the LDS (the IDL instances)"""
    L = lib()
    L.kas_emu_last_relax_idl.restype = C.c_int
    return int(L.kas_emu_last_relax_idl())


FULL_FILL = 16             # KAS_PLAN_FULL_FILL: kas_fill_kernel for every scenario (no kas_fill_slim_kernel in front)
NO_INDEX_ROWS = 64         # KAS_PLAN_NO_INDEX_ROWS: the fill reads `cur` in both of its row scans
RELAX_TILES_256 = 0x20000 | 0x40000   # KAS_PLAN_RELAX_TILES(3): quad tiles (256 rows a step) where the instances apply (dword mid rows), else double tiles
MID32 = 0x80000            # KAS_PLAN_MID32: dword mid rows (holders sorted, 11 bits each) where they apply
NO_MID32 = 0x100000        # KAS_PLAN_NO_MID32: the packed 16-bit mid rows
INDEX_ROWS = 128           # KAS_PLAN_INDEX_ROWS: the fill's first scan leaves node indices where the mid rows go, the second streams those


def last_spread() -> int:
    """Scenarios the spread fill solved itself (not handed back to the one-workgroup kernel) in the last emu_solve."""
    L = lib()
    L.kas_emu_last_spread.restype = C.c_int
    return int(L.kas_emu_last_spread())


def plan_shape(fb: FlatBatch):
    """kas_shape_batch's verdict on a batch shape (the product's planning code, nothing is run):
    (return code, {tickets_ok, wide_ok, round_fits, G, NW, with_x, packed_ok, fused_ok, wide_checked}, error text)."""
    L = lib()
    L.kas_emu_shape.restype = C.c_int
    L.kas_emu_shape.argtypes = [C.POINTER(abi.BatchDesc), C.POINTER(C.c_int32), C.c_char_p, C.c_int]
    bd = batch_desc(fb)
    out = (C.c_int32 * 10)()
    err = C.create_string_buffer(512)
    rc = L.kas_emu_shape(C.byref(bd), out, err, 512)
    names = ("tickets_ok", "wide_ok", "round_fits", "G", "NW", "with_x", "packed_ok", "fused_ok", "wide_checked", "relax_ok")
    return rc, dict(zip(names, list(out))), err.value.decode()


def last_flagged() -> int:
    """Scenarios a ticket form left to the round form in the last emu_solve (Context counters beyond its count fields;
    wide form: counts that outgrew the fields, found by its check at the end)."""
    L = lib()
    L.kas_emu_last_flagged.restype = C.c_int
    return int(L.kas_emu_last_flagged())


def last_order_form() -> int:
    """Order kernel of the last emu_solve: 1 ticket form (lists <= 3 wide), 2 wide ticket form, 3 relaxation form,
    0 round form."""
    L = lib()
    L.kas_emu_last_order_form.restype = C.c_int
    return int(L.kas_emu_last_order_form())


def last_split_p4() -> int:
    """1: the last emu_solve ran its first fit (P4) in kas_p4_kernel behind the fill kernel (KAS_FLAG_SPLIT_P4)"""
    L = lib()
    L.kas_emu_last_split_p4.restype = C.c_int
    return int(L.kas_emu_last_split_p4())


FILL_WITH_P4 = 0x800000    # KAS_PLAN_FILL_WITH_P4: first fit inside the fill workgroup whatever the batch size
SPLIT_P4 = 0x400000        # KAS_PLAN_SPLIT_P4: first fit in kas_p4_kernel whatever the batch size
TICKET_ORDER = 0x10000     # KAS_PLAN_TICKET_ORDER: the ticket form where the relaxation form would run
RELAX_TILES_64 = 0x20000   # KAS_PLAN_RELAX_TILES(1): relaxation form over tiles of 64 rows whatever the batch size
RELAX_TILES_128 = 0x40000  # KAS_PLAN_RELAX_TILES(2): double tiles whatever the batch size
NO_RTN_QUOTA = 0x200000    # KAS_PLAN_NO_RTN_QUOTA: the fill draws its quota without the atomic-with-return


def VERIFY_SAMPLE(k: int) -> int:
    """KAS_PLAN_VERIFY_SAMPLE(k): the relaxation form evaluates k tiles per topic again, one row at a time"""
    return (k & 0xff) << 24


def last_relax_stats():
    """Relaxation form of the last emu_solve: (tiles, evaluations, tiles off the straight-line path)."""
    L = lib()
    L.kas_emu_last_relax_stats.restype = None
    out = (C.c_long * 3)()
    L.kas_emu_last_relax_stats(out)
    return tuple(int(v) for v in out)


def spread_plan(fb: FlatBatch):
    """Spread-fill planning for a batch shape (nothing is run): chunks per scenario (0 = one-workgroup fill), LDS
    bytes of the scan kernels (pass A, pass B) and of the one-workgroup layout they used to carry, tiles per scenario."""
    L = lib()
    L.kas_emu_spread_plan.restype = C.c_int
    L.kas_emu_spread_plan.argtypes = [C.POINTER(abi.BatchDesc), C.POINTER(C.c_int32)]
    bd = batch_desc(fb)
    out = (C.c_int32 * 5)()
    rc = L.kas_emu_spread_plan(C.byref(bd), out)
    return rc, dict(zip(("chunks", "lds_a", "lds_b", "lds_full", "tiles"), list(out)))
