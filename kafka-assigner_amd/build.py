"""Build recipe for the C-ABI library (hipcc cross-compiles gfx950 without a GPU).

  python -m kafka_assigner_amd.build        # builds kafka-assigner_amd/csrc/libkas_hip.so in-tree
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libkas_hip.so")
SOURCES = ["kas_hip.hip"]
HEADERS = ["kas_solver_body.h", "kas_order_wide.h", "kas_order_relax.h", "kas_plan_math.h", "kas_wave.h"]


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.join(ROOT, "include", "kas_abi.h")]
    return os.path.getmtime(LIB) < max(os.path.getmtime(d) for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 into csrc/libkas_hip.so (no-op when up to date)."""
    if not force and not is_stale():
        return LIB
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-fvisibility=hidden", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
           "-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


HOST = os.path.join(HERE, "host")
CLI = os.path.join(HOST, "kafka-assignment-generator")


def build_host(force: bool = False, verbose: bool = False) -> str:
    """Compile the C++ host mirror's CLI (host/kas_cli.cpp) against csrc/libkas_hip.so."""
    build(verbose=verbose)
    deps = [os.path.join(HOST, f) for f in ("kas_cli.cpp", "kafka_assigner.hpp", "mini_json.hpp")]
    deps += [LIB, os.path.join(ROOT, "include", "kas_abi.h")]
    if not force and os.path.exists(CLI) and os.path.getmtime(CLI) >= max(os.path.getmtime(d) for d in deps):
        return CLI
    cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-I" + os.path.join(ROOT, "include"), "-I" + HOST,
           "-o", CLI, os.path.join(HOST, "kas_cli.cpp"), "-L" + CSRC, "-lkas_hip",
           "-Wl,-rpath,$ORIGIN/../csrc", "-Wl,--allow-shlib-undefined"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return CLI


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
