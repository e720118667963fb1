"""The parallel P4 of the fill kernel (first fit, windows of 64 orphans handed from wavefront to
wavefront) as a unit on the CPU emulator, against a sequential first fit — with one wavefront made
slow, so that a window which may not wait for every earlier window overtakes.

Round 2's bug: window w waited for window w - 1 only.  Here window 0 (on the slow wave) has one
orphan whose racks rule out the first four nodes, so it walks on to the second position group;
window 1's orphans all fit the first group, so it finishes as soon as window 0 is done with that
group; window 2 also has an orphan for the second group, where the first node has ONE slot left.
Sequentially that slot is window 0's; a window 2 that only looks at window 1 takes it."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NONE = 0xffff


def first_fit(load, rack, cap, live, orphans, mid):
    """KAS:162-186 over the non-full nodes in processing order (the reference's rescan from the start
    finds the same node: a full node stays full, a refused rack stays refused)."""
    load = load.copy(); mid = mid.copy()
    for p in orphans:
        hold = [int(v) for v in mid[p][:3] if v != NONE]
        for j in live:
            if len(hold) == 3:
                break
            if load[j] < cap and rack[j] not in [rack[h] for h in hold]:
                mid[p][len(hold)] = j; hold.append(int(j)); load[j] += 1
        assert len(hold) == 3
    return load, mid


def build_case():
    # nodes 0..3: racks 0,0,1,1, plenty of room; node 4: rack 2, ONE slot; node 5: rack 3, plenty;
    # nodes 6, 7: the holders the orphan rows already have (full, not in the list)
    cap = 1000
    load = np.array([0, 0, 0, 0, cap - 1, 0, cap, cap, cap, cap], dtype=np.int32)
    rack = np.array([0, 0, 1, 1, 2, 3, 4, 5, 0, 1], dtype=np.int32)
    live = np.array([0, 1, 2, 3, 4, 5], dtype=np.int32)
    n_orph = 3 * 64
    P = n_orph
    mid = np.full((P, 3), NONE, dtype=np.uint16)
    mid[:, 0] = 6; mid[:, 1] = 7                       # two holders on racks 4, 5: any listed node is fine
    for p in (5, 2 * 64 + 9):                          # one orphan of window 0 and one of window 2: holders on racks 0, 1
        mid[p, 0] = 8; mid[p, 1] = 9
    return cap, load, rack, live, np.arange(n_orph, dtype=np.int32), mid


def run_case(div):
    code = (
        "import sys, ctypes as C; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "from emu_lib import lib\n"
        "from test_emu_p4_windows import build_case, first_fit\n"
        "cap, load, rack, live, orph, mid = build_case()\n"
        "want_load, want_mid = first_fit(load, rack, cap, live, orph, mid)\n"
        "L = lib(); L.kas_emu_p4_unit.restype = C.c_int\n"
        "got_mid = mid.copy(); got_load = np.zeros_like(load); o = orph.copy()\n"
        "p = lambda a: a.ctypes.data_as(C.c_void_p)\n"
        "rc = L.kas_emu_p4_unit(C.c_int(len(load)), p(load), p(rack), C.c_int(cap), C.c_int(len(live)), p(live),\n"
        "                       C.c_int(len(o)), p(o), C.c_int(mid.shape[0]), p(got_mid), p(got_load))\n"
        "assert rc == 0, rc\n"
        "assert (got_load == want_load).all(), (got_load.tolist(), want_load.tolist())\n"
        "bad = np.nonzero((got_mid != want_mid).any(axis=1))[0]\n"
        "assert len(bad) == 0, [(int(b), got_mid[b].tolist(), want_mid[b].tolist()) for b in bad[:4]]\n"
        "print('ok')\n") % (os.path.join(ROOT, "tests"), ROOT)
    env = dict(os.environ)
    env.pop("KAS_EMU_CHAOS", None)
    if div:
        env["KAS_EMU_WAVE_DIV"] = div
    else:
        env.pop("KAS_EMU_WAVE_DIV", None)
    return subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, env=env, timeout=600)


@pytest.mark.parametrize("div", ["", "0:40", "0:15,3:7"])     # 0:40 is the one the predecessor-only wait fails
def test_p4_windows_equal_sequential_first_fit_whatever_the_wave_speeds(div):
    r = run_case(div)
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout[-800:], r.stderr[-1500:])


def random_case(rng):
    """A list of nodes on three racks with little room each (the last slots of nodes are contested all
    the time), orphan rows of two kinds: most have their holders on other racks and take the first
    node with room; one in seventy holds two of the three racks already and walks the list for a node of
    the third — the orphan that an overtaking window would rob."""
    n_live = int(rng.integers(12, 30))
    hold_racks = np.array([3, 4, 5, 6, 0, 1, 2], dtype=np.int32)      # full nodes the rows already sit on
    n = n_live + len(hold_racks)
    rack = np.concatenate([rng.integers(0, 3, size=n_live), hold_racks]).astype(np.int32)
    n_orph = int(rng.integers(300, 700))
    mid = np.full((n_orph, 3), NONE, dtype=np.uint16)
    need_total = 0
    for p in range(n_orph):
        if rng.random() < 0.015:
            hs = (n_live + 4 + rng.permutation(3)[:2]).tolist()       # holders on two of the racks 0, 1, 2
        else:
            k = int(rng.integers(1, 3))
            hs = (n_live + rng.permutation(4)[:k]).tolist()           # holders on racks 3..6
        mid[p, :len(hs)] = hs
        need_total += 3 - len(hs)
    cap = need_total + 10
    room = rng.multinomial(need_total, np.ones(n_live) / n_live).astype(np.int32)
    load = np.full(n, cap, dtype=np.int32)
    load[:n_live] = cap - room
    # three roomy nodes at the end of the list, one per rack: every row finds a place
    rack = np.concatenate([rack, np.array([0, 1, 2], dtype=np.int32)])
    load = np.concatenate([load, np.zeros(3, dtype=np.int32)])
    live = np.concatenate([rng.permutation(n_live), n + np.arange(3)]).astype(np.int32)
    return cap, load, rack, live, np.arange(n_orph, dtype=np.int32), mid


@pytest.mark.parametrize("env", [{"KAS_EMU_WAVE_DIV": "0:25"}, {"KAS_EMU_WAVE_DIV": "3:25,2:4"}, {"KAS_EMU_CHAOS": "5"}])
def test_p4_windows_random_cases_under_skewed_wave_speeds(env):
    code = (
        "import sys, ctypes as C; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "from emu_lib import lib\n"
        "from test_emu_p4_windows import random_case, first_fit\n"
        "L = lib(); L.kas_emu_p4_unit.restype = C.c_int\n"
        "p = lambda a: a.ctypes.data_as(C.c_void_p)\n"
        "rng = np.random.default_rng(20260923)\n"
        "for case in range(60):\n"
        "    cap, load, rack, live, orph, mid = random_case(rng)\n"
        "    want_load, want_mid = first_fit(load, rack, cap, live, orph, mid)\n"
        "    got_mid = mid.copy(); got_load = np.zeros_like(load); o = orph.copy()\n"
        "    rc = L.kas_emu_p4_unit(C.c_int(len(load)), p(load), p(rack), C.c_int(cap), C.c_int(len(live)), p(live),\n"
        "                           C.c_int(len(o)), p(o), C.c_int(mid.shape[0]), p(got_mid), p(got_load))\n"
        "    assert rc == 0, (case, rc)\n"
        "    bad = np.nonzero((got_mid != want_mid).any(axis=1))[0]\n"
        "    assert len(bad) == 0 and (got_load == want_load).all(), (case, [(int(b), got_mid[b].tolist(), want_mid[b].tolist()) for b in bad[:3]])\n"
        "print('ok')\n") % (os.path.join(ROOT, "tests"), ROOT)
    e = dict(os.environ)
    e.pop("KAS_EMU_CHAOS", None); e.pop("KAS_EMU_WAVE_DIV", None)
    e.update(env)
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, env=e, timeout=900)
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout[-800:], r.stderr[-1500:])
