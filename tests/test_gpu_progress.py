"""Forward progress of the many-workgroup assignment tail (k_assign_label + k_assign_solve; reference: SortVoting::winners,
src/trackers/sort/voting.rs:30-100) does not depend on how many of its workgroups the device holds at once.

k_assign_solve's row workgroups never wait; the helper workgroups launched behind them wait for "every row workgroup has reported",
and on every XCD a helper is dispatched after the row workgroups it waits for.  The test takes the device away: a child process started
under a CU mask of TWO compute units (ROC_GLOBAL_CU_MASK for the HIP runtime's queues, HSA_CU_MASK for ROCr) runs crowd frames and a
frame of 16 row workgroups — more than two CUs hold — and must return the oracle's assignment, not SA_ERR_HIP from a wait that ran out.
Whether the mask took effect is read from a compute-bound launch timed in both children."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_child(env_extra):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "cu_mask_child.py")], env=env, capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("CU-MASK-CHILD ")][-1]
    return json.loads(line[len("CU-MASK-CHILD "):])


def check(d):
    assert len(d["frames"]) == 3
    for f in d["frames"]:
        assert f["same_total_gain"] and f["votes_match"] and f["repeat_matches"], f
        assert f["matched"] > 100, f
    assert d["frames"][2]["ids_match"], d["frames"][2]   # (isolated pairs: a unique optimum)


@pytest.mark.gpu
def test_assignment_tail_on_a_device_masked_down_to_two_compute_units():
    full = run_child({})
    check(full)
    masked = run_child({"ROC_GLOBAL_CU_MASK": "0x3", "HSA_CU_MASK": "0:0-1"})
    check(masked)
    slowdown = masked["gemm_ms"] / full["gemm_ms"]
    print(f"contraction under the mask: {masked['gemm_ms']:.3f} ms against {full['gemm_ms']:.3f} ms (x{slowdown:.1f}); frames: "
          + ", ".join(f"{a['frame']} {b['two_frames_ms']:.1f} ms (unmasked {a['two_frames_ms']:.1f})" for a, b in zip(full["frames"], masked["frames"])))
    # (the mask is honoured by this stack when the contraction slows down by an order of magnitude; if a future runtime ignores both
    # variables the frames above still ran and matched — the test then says so instead of claiming what it did not exercise)
    if slowdown < 4.0:
        pytest.skip(f"the CU mask was not honoured by this runtime (contraction x{slowdown:.2f}): frames matched the oracle on the full device only")
