"""The PMC figures bench.py quotes (profiles/traffic_latest.json: HBM bytes per launch, matrix-pipe busy fraction) were collected
from a build of particular kernel sources; the file records their hash (`csrc_sha256`, dgps_with_iwvi_amd.kernel_resources.csrc_hash)
and bench.py says in its line whether the tree it runs from still matches (`roofline.pmc_profile_is_of_these_sources`).

A profile can only be refreshed on the GPU box AFTER a kernel change exists, so a stale profile is reported (xfail + warning), not a
failure of the CPU suite; the file's own consistency is asserted."""
import json
import os
import warnings

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pmc_profile_is_of_the_current_kernel_sources():
    from dgps_with_iwvi_amd.kernel_resources import csrc_hash
    tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
    for k in ("kernel", "hbm_bytes", "mfma_busy_frac", "kernel_avg_us_rocprof", "source"):
        assert k in tj, k
    assert "k_dgp_forward" in tj["kernel"] and tj["hbm_bytes"] > 0 and 0.0 < tj["mfma_busy_frac"] < 1.0
    assert os.path.exists(os.path.join(ROOT, tj["source"].split(" ")[0])), tj["source"]
    got, now = tj.get("csrc_sha256"), csrc_hash()
    if got != now:
        msg = ("profiles/traffic_latest.json was collected from kernel sources %s (commit %s); the tree is at %s: re-run "
               "scripts/profile_round.sh + scripts/summarise_profile.py <tag> --latest on the GPU box" % (got, tj.get("pmc_profile_of_commit"), now))
        warnings.warn(msg)
        pytest.xfail(msg)
