"""The walk kernels' index arithmetic, checked by construction (a GPU memory fault in round 2's fuzz run was
never traced to its access -- the faulting tree was not kept).  `make EXTRA=-DTLM_CHECKED` builds the library with every
global index of the walk kernels compared with its array's size BEFORE the access (list entries of 8-byte and compact rows,
row records, {q_start, t_start}, the wave's path / mark-log / visited-map scratch, the arena, the unit records); an index
outside is recorded and skipped instead of dereferenced, and the walk call fails with TLM_EINTERNAL.  This test runs the
walk parity tests -- small graphs whose last rows end at the arrays' ends, every lane width with last lanes wholly in row
padding, walks whose band slides up to the last word of the visited map, path-buffer and arena overflow re-runs --
against that library in a child process: one offending index anywhere fails it.  The limits are the arrays' EXACT
sizes (no allowance for the 256 entries of slack behind the last row)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHECKED = os.path.join(ROOT, "telomeri_amd", "lib_checked", "libtelomeri_hip.so")

pytestmark = pytest.mark.gpu

SELECTION = ("fuzz_small_graphs or fuzz_deterministic_dense or walks_across_lane_widths or walk_mc_per_walk_seed or test_walk_det or "
             "walks_on_adopted_csr or mark_log or overflow_reruns or walks_quirks_all_metrics or two_metric_launch or "
             "accounting or rng_stream")


def test_walk_parity_suite_on_the_checked_build():
    if not os.path.exists(CHECKED):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "telomeri_amd", "csrc"), "OUT=" + os.path.dirname(CHECKED),
                        "EXTRA=-DTLM_CHECKED"], check=True)
    res = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-q", "-x", "-m", "gpu",
                          "-k", SELECTION, "-p", "no:cacheprovider"], capture_output=True, text=True, timeout=1500,
                         env=dict(os.environ, TLM_LIB=CHECKED))
    tail = res.stdout[-3000:] + res.stderr[-1000:]
    assert res.returncode == 0, tail
    assert "checked build" not in res.stdout, tail
    passed = [ln for ln in res.stdout.splitlines() if " passed" in ln]
    assert passed and int(passed[-1].split(" passed")[0].split()[-1]) >= 100, tail


def test_checked_build_reports_an_offending_index():
    """The check itself: with the limit of the list arrays lowered by the host (TLM_CHECK_SHRINK, checked build only) the last
    rows lie outside, the walk call fails and names the site -- nothing faults."""
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from telomeri_amd import capi, synth\nimport helpers as H\n"
            "ctx = capi.Context(0)\n"
            "dg = H.device_graph(ctx, synth.make_config('tiny'), capi.FilterParams.default())\n"
            "try:\n    dg.walk_mc(capi.METRIC_ES, capi.WalkParams.default(rb_att=30))\n    print('NO ERROR')\n"
            "except capi.TlmError as e:\n    print('ERR', e)\n" % (ROOT, os.path.join(ROOT, "tests")))
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, TLM_LIB=CHECKED, TLM_CHECK_SHRINK="0.5"))
    assert "ERR" in res.stdout and "checked build" in res.stdout and "formed an index outside" in res.stdout, res.stdout + res.stderr
