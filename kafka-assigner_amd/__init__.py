"""kafka-assigner on MI355X: batch solver for the reference's minimal-movement, rack-aware
rebalance path (KafkaAssignmentStrategy.getRackAwareAssignment, KafkaAssignmentStrategy.java:40-63).

Layout:
  csrc/      HIP kernels (gfx950) + the C-ABI shared library declared in include/kas_abi.h
  host/      C++ mirror of KafkaTopicAssigner / KafkaAssignmentStrategy over the C ABI, the
             kafka-assignment-generator CLI over a cluster snapshot, JNI shim
  abi.py     ctypes mirror of the ABI structs
  flatten.py reference-shaped arguments <-> flat int32 tables
  native.py  loader/wrapper of the C-ABI library (fails loudly when it is not built / no GPU)
  assigner.py  Python mirror of KafkaTopicAssigner.generateAssignment over the native path
  generator.py synthetic cluster scenarios (BASELINE.json configs)
  sharding.py  scenario sharding across ranks + the result-record all-gather
  whatif.py  one snapshot x many broker-set variants -> one batch over a shared cur table
  build.py   hipcc build recipe
"""
from . import abi  # noqa: F401
from .flatten import (FlatBatch, Scenario, Topic, flatten, uniform_batch,  # noqa: F401
                      unflatten_topic, unflatten_context, java_string_hashcode)

__all__ = ["abi", "FlatBatch", "Scenario", "Topic", "flatten", "uniform_batch",
           "unflatten_topic", "unflatten_context", "java_string_hashcode"]
