"""The relaxation-form order kernel issues its global loads as inline assembly (kasw::gload_*_async) and waits for them
once per step (kasw::wait_loads): the compiler does not know those registers are in flight.  This test compiles every
instance of the kernel to gfx950 assembly (hipcc cross-compiles without a GPU) and proves, by data flow over the basic
blocks of what the compiler actually emitted, that nothing touches an in-flight register before the wait
(tools/check_async_loads.py).  It also pins what the change was for: ONE s_waitcnt vmcnt in the hot loop's tail, not a
compiler-placed vmcnt(0) behind the requests of the same iteration."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_no_instruction_touches_a_register_with_a_load_in_flight(tmp_path):
    asm = tmp_path / "relax.s"
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "kafka-assigner_amd", "csrc"),
           "-o", str(asm), os.path.join(ROOT, "tests", "asm", "relax_instances.hip")]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_async_loads.py"), str(asm)],
                       capture_output=True, text=True)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("_Z")]
    assert len(lines) == 42, r.stdout                       # 33 instances of the relaxation form (6 gathering ids, 12 with ids in the LDS, 3 on dword mid rows — one of them over quad tiles —, 12 for 16-bit cells) + 9 of kas_p4_order_kernel (3 on dword mid rows)
    assert all(" 0 problems" in l for l in lines)


def test_the_checker_sees_a_copy_of_an_in_flight_register(tmp_path):
    """negative control: the hazard the checker exists for, hand-written"""
    bad = tmp_path / "bad.s"
    bad.write_text("""_Zbad:
	v_mov_b32_e32 v3, -1
	;;#ASMSTART
	global_load_dword v3, v2, s[4:5] offset:0
	;;#ASMEND
	s_cbranch_scc1 .LBB0_2
	v_mov_b32_e32 v7, v3
.LBB0_2:
	;;#ASMSTART
	s_waitcnt vmcnt(0)
	;;#ASMEND
	v_mov_b32_e32 v8, v3
	s_endpgm
.Lfunc_end0:
""")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_async_loads.py"), str(bad)],
                       capture_output=True, text=True)
    assert r.returncode == 1 and "v_mov_b32_e32 v7, v3" in r.stdout and "v8" not in r.stdout, r.stdout
