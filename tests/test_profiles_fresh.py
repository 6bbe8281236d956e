"""The PMC figures bench.py quotes (profiles/traffic_latest.json: HBM bytes per launch, matrix-pipe busy fraction) were collected
at a commit; they describe the kernels only while no later commit has touched dgps_with_iwvi_amd/csrc.  Fails when they are stale
(re-run scripts/profile_round.sh + scripts/summarise_profile.py <tag> --latest on the GPU box after the last kernel change)."""
import json
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pmc_profile_is_of_the_kernels_last_commit():
    if shutil.which("git") is None or not os.path.isdir(os.path.join(ROOT, ".git")):
        pytest.skip("no git history here (a snapshot of the tree)")
    try:
        last = subprocess.check_output(["git", "log", "-1", "--format=%h", "--", "dgps_with_iwvi_amd/csrc"], cwd=ROOT,
                                       stderr=subprocess.DEVNULL).decode().strip()
        dirty = subprocess.check_output(["git", "status", "--porcelain", "--", "dgps_with_iwvi_amd/csrc"], cwd=ROOT,
                                        stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        pytest.skip("git not usable here")
    tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
    got = tj.get("pmc_profile_of_commit")
    assert got, "profiles/traffic_latest.json carries no commit"
    # (abbreviations may differ in length: compare as prefixes)
    assert last.startswith(got) or got.startswith(last), (
        "profiles/traffic_latest.json was collected at %s, the kernels were last changed at %s: re-profile" % (got, last))
    assert not [l for l in dirty.splitlines() if l and not l.endswith((".o", ".so"))], "uncommitted kernel changes: the profile is of another tree"
