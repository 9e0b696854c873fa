"""-m gpu: message-passing litmus of the hand-offs the one-launch YZ stage ships with (tools/zy_litmus.hip, built as lib/zy_litmus):
cross-XCD producer / consumer pairs with the instruction pattern of csrc/dfft_zy.hip -- 16-byte sc1 stores, s_waitcnt vmcnt(0), relaxed
agent-scope increment | relaxed poll, sc1 loads -- and the part counter of the overlapped pipeline (sc1 nt stores + count | wait kernel and
a plain-load reader kernel on another stream).  The shipped patterns must show no stale read in ~10^6 hand-offs; the negative controls
(plain loads on the consumer, plain stores on the producer) must show some -- if they stop failing the test no longer proves anything."""
import re
import subprocess
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
EXE = Path(__file__).resolve().parent.parent / "distributedfft_amd" / "lib" / "zy_litmus"


def _run(iters, mode, kib=16):
    assert EXE.exists(), f"{EXE} missing: python -m distributedfft_amd.build"
    r = subprocess.run([str(EXE), str(iters), str(mode), str(kib)], capture_output=True, text=True, timeout=600)
    m = re.search(r"(\d+) hand-offs of \d+ KiB, (\d+) stale 16-byte reads, (\d+) time-outs, (\d+) of (\d+) pairs across XCDs", r.stdout)
    assert m, r.stdout + r.stderr
    return r.returncode, [int(v) for v in m.groups()], r.stdout.strip()


def test_shipped_hand_off_shows_no_stale_read(gpu):
    rc, (handoffs, stale, timeouts, cross, pairs), out = _run(4000, 0)   # 256 workgroups x 4000 iterations = 1.02e6 hand-offs
    assert rc == 0 and stale == 0 and timeouts == 0 and handoffs >= 10 ** 6, out
    assert cross == pairs, out   # every pair really sat on two XCDs


def test_part_counter_of_the_overlapped_pipeline_shows_no_stale_read(gpu):
    rc, (handoffs, stale, timeouts, _, _), out = _run(2000, 3, 64)
    assert rc == 0 and stale == 0 and timeouts == 0 and handoffs >= 2000 * 16 * 32, out


@pytest.mark.parametrize("mode", [1, 2])
def test_negative_controls_fail(gpu, mode):
    rc, (handoffs, stale, timeouts, cross, pairs), out = _run(500, mode)
    assert rc == 0 and stale > 0 and timeouts == 0, out   # rc 0 = "the control fails as it must"
