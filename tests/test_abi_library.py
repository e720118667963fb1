"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, exports
every symbol include/kas_abi.h declares, and its structs have the layout the ctypes mirror
assumes.  No compute calls (there is no GPU here) — but the entry points must FAIL LOUDLY
without a device instead of falling back to anything."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from kafka_assigner_amd import abi, build, native
from kafka_assigner_amd.flatten import Scenario, Topic, flatten

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "kas_abi.h")


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(kas_[a-z_0-9]+)\s*\(", src)) - {"kas_digest_cell"})


def test_library_builds_and_exports_every_declared_symbol():
    so = build.build()
    assert os.path.exists(so)
    L = native.load()
    declared = _declared_functions()
    assert len(declared) >= 12
    for name in declared:
        assert hasattr(L, name), f"{name} declared in kas_abi.h but not exported"
    assert sorted(native.SYMBOLS) == declared
    assert L.kas_abi_version() == abi.KAS_ABI_VERSION
    assert L.kas_strerror(abi.KAS_E_HIP)
    assert b"KAS:183" in L.kas_status_string(abi.KAS_FAIL_UNASSIGNABLE)


def test_library_contains_gfx950_code_object():
    so = build.build()
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--list", "--type=o",
                          f"--input={so}"], capture_output=True, text=True)
    blob = open(so, "rb").read()
    assert b"gfx950" in blob, out.stdout + out.stderr
    assert b"kas_fill_kernel" in blob and b"kas_order_ticket_kernel" in blob and b"kas_order_round_kernel" in blob


def test_struct_layout_matches_c(tmp_path):
    probe = tmp_path / "probe.c"
    probe.write_text(r'''
#include <stdio.h>
#include <stddef.h>
#include "kas_abi.h"
#define S(t) printf(#t " %zu\n", sizeof(t))
#define O(t, f) printf(#t "." #f " %zu\n", offsetof(t, f))
int main(void) {
  S(kas_topic_desc); O(kas_topic_desc, cur_off); O(kas_topic_desc, part_id_off);
  S(kas_scenario_desc); O(kas_scenario_desc, node_off); O(kas_scenario_desc, ctx_off);
  S(kas_topic_result); S(kas_scenario_result); O(kas_scenario_result, digest);
  S(kas_batch_desc); O(kas_batch_desc, node_pool_len);
  S(kas_tables); O(kas_tables, cur_len); O(kas_tables, ctx_len);
  S(kas_rf_result); O(kas_rf_result, fail_list_size);
  printf("digest %llu\n", (unsigned long long)kas_digest_cell(2, 77, 1, 1005));
  return 0;
}''')
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-std=c11", "-I" + os.path.join(ROOT, "include"), str(probe), "-o", str(exe)])
    got = dict(line.rsplit(" ", 1) for line in subprocess.check_output([str(exe)], text=True).splitlines())
    assert int(got["kas_topic_desc"]) == C.sizeof(abi.TopicDesc)
    assert int(got["kas_topic_desc.cur_off"]) == abi.TopicDesc.cur_off.offset
    assert int(got["kas_topic_desc.part_id_off"]) == abi.TopicDesc.part_id_off.offset
    assert int(got["kas_scenario_desc"]) == C.sizeof(abi.ScenarioDesc)
    assert int(got["kas_scenario_desc.node_off"]) == abi.ScenarioDesc.node_off.offset
    assert int(got["kas_scenario_desc.ctx_off"]) == abi.ScenarioDesc.ctx_off.offset
    assert int(got["kas_topic_result"]) == C.sizeof(abi.TopicResult)
    assert int(got["kas_scenario_result"]) == C.sizeof(abi.ScenarioResult)
    assert int(got["kas_scenario_result.digest"]) == abi.ScenarioResult.digest.offset
    assert int(got["kas_batch_desc"]) == C.sizeof(abi.BatchDesc)
    assert int(got["kas_batch_desc.node_pool_len"]) == abi.BatchDesc.node_pool_len.offset
    assert int(got["kas_tables"]) == C.sizeof(abi.Tables)
    assert int(got["kas_tables.cur_len"]) == abi.Tables.cur_len.offset
    assert int(got["kas_tables.ctx_len"]) == abi.Tables.ctx_len.offset
    assert int(got["kas_rf_result"]) == C.sizeof(abi.RfResult)
    assert int(got["kas_rf_result.fail_list_size"]) == abi.RfResult.fail_list_size.offset
    assert int(got["digest"]) == abi.digest_cell(2, 77, 1, 1005)


def _gpu_present():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_gpu_present(), reason="this check is for GPU-less machines")
def test_no_cpu_fallback_without_a_device():
    L = native.load()
    assert L.kas_device_count() == 0
    with pytest.raises(native.KasError) as e:
        native.DeviceContext(0)
    assert e.value.code == abi.KAS_E_HIP
    assert "no CPU path" in e.value.detail
    from kafka_assigner_amd.assigner import KafkaTopicAssigner
    with pytest.raises(native.KasError):
        KafkaTopicAssigner().generate_assignment("test", {0: [1, 2]}, {1, 2, 3}, {}, -1)


def test_product_never_references_the_oracle():
    """The oracle is test infrastructure: nothing under the product package may import, link,
    dlopen or mention it."""
    pkg = os.path.join(ROOT, "kafka-assigner_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".so", ".o", ".pyc")):
                continue
            text = open(os.path.join(dirpath, f), errors="ignore").read()
            for needle in ("kas_oracle", "literal_ref", "oracle_lib", "libkas_emu", "emu_lib"):
                assert needle not in text, f"{f} mentions {needle}"
    blob = open(native.LIB_PATH, "rb").read()
    assert b"kas_oracle" not in blob
