"""The implementations the parity tests run side by side.

  literal : oracle/literal_ref.py  (line-by-line Python restatement of the Java)
  oracle  : oracle/kas_oracle.c    (C restatement, via ctypes)
  emu     : the product's kernel SOURCE (csrc/kas_solver_body.h, kas_order_wide.h) stepped on the CPU fiber
            emulator (tests/emu) behind the product's host mirror — test infrastructure, no GPU
  hip     : the product — kafka_assigner_amd.KafkaTopicAssigner over the C ABI and HIP kernels
            (GPU only; tests using it are marked @pytest.mark.gpu); it calls kas_solve_host16 (16-bit node-index cells),
            hip32 the same mirror through kas_solve_host (int32 broker ids)

Each is exposed behind the reference's own interface: an object with
generate_assignment(topic, current_assignment, brokers, rack_assignment, desired_rf)
that raises on the reference's error paths.
"""
from __future__ import annotations

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import literal_ref  # noqa: E402
from kafka_assigner_amd import assigner as A  # noqa: E402
from oracle_lib import oracle_solve  # noqa: E402


class LiteralAssigner:
    def __init__(self):
        self._impl = literal_ref.KafkaTopicAssigner()

    @property
    def context(self):
        return {n: {r: c for r, c in m.items() if c} for n, m in
                self._impl.assignment_context.counter.items() if any(m.values())}

    def generate_assignment(self, topic, cur, brokers, racks, desired_rf):
        try:
            return self._impl.generate_assignment(topic, cur, set(brokers), dict(racks), desired_rf)
        except literal_ref.IllegalStateException as e:
            raise A.IllegalStateException(str(e))
        except literal_ref.ArrayIndexOutOfBoundsException as e:
            raise A.ArrayIndexOutOfBoundsException(str(e))


class OracleAssigner:
    """KTA:42-72 host logic from the product mirror, solve by the C oracle."""

    def __init__(self):
        self.assignment_context = A.Context()

    @property
    def context(self):
        return {n: dict(m) for n, m in self.assignment_context.counter.items() if m}

    def generate_assignment(self, topic, cur, brokers, racks, desired_rf):
        rf = A.resolve_replication_factor(topic, cur, len(set(brokers)), desired_rf)
        result, _ = A._solve_one(oracle_solve, topic, cur, racks, set(brokers),
                                 set(cur.keys()), rf, self.assignment_context)
        return result


class EmuAssigner(OracleAssigner):
    """KTA:42-72 host logic from the product mirror, solve by the kernel source on the emulator."""

    def generate_assignment(self, topic, cur, brokers, racks, desired_rf):
        from emu_lib import emu_solve
        rf = A.resolve_replication_factor(topic, cur, len(set(brokers)), desired_rf)
        result, _ = A._solve_one(emu_solve, topic, cur, racks, set(brokers),
                                 set(cur.keys()), rf, self.assignment_context)
        return result


class HipAssigner:
    def __init__(self):
        self._impl = A.KafkaTopicAssigner()

    @property
    def context(self):
        return {n: dict(m) for n, m in self._impl.assignment_context.counter.items() if m}

    def generate_assignment(self, topic, cur, brokers, racks, desired_rf):
        return self._impl.generate_assignment(topic, cur, set(brokers), dict(racks), desired_rf)


class HipAssignerInt32Cells(HipAssigner):
    """the same mirror with KAS_CELLS32=1: kas_solve_host (int32 broker ids) instead of kas_solve_host16 (node indices)"""

    def generate_assignment(self, topic, cur, brokers, racks, desired_rf):
        old = os.environ.get("KAS_CELLS32")
        os.environ["KAS_CELLS32"] = "1"
        try:
            return super().generate_assignment(topic, cur, brokers, racks, desired_rf)
        finally:
            if old is None:
                del os.environ["KAS_CELLS32"]
            else:
                os.environ["KAS_CELLS32"] = old


IMPLS = {
    "literal": LiteralAssigner,
    "oracle": OracleAssigner,
    "emu": EmuAssigner,
    "hip": HipAssigner,
    "hip32": HipAssignerInt32Cells,
}

# parametrisation helper: CPU implementations always, the product only on the GPU box
ALL_IMPLS = [
    pytest.param("literal", id="literal"),
    pytest.param("oracle", id="oracle"),
    pytest.param("emu", id="emu"),
    pytest.param("hip", id="hip", marks=pytest.mark.gpu),
    pytest.param("hip32", id="hip-int32-cells", marks=pytest.mark.gpu),
]
