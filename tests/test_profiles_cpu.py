"""CPU: the committed profile of this round (profiles/r06_manifest.json, written by tools/refresh_profiles.py from a tools/profile_run.sh
run on the GPU box) must have been measured on the kernel sources that are in the tree: a kernel change after the last profiling run
makes the rocprofv3 numbers, the PMC traffic and the measured parity table stale -- refresh them (or this test says so)."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAN = os.path.join(ROOT, "profiles", "r06_manifest.json")


@pytest.mark.skipif(not os.path.isfile(MAN), reason="no profile committed for this round yet")
def test_committed_profile_was_measured_on_these_kernel_sources():
    from lightzero_amd.build import csrc_digest
    m = json.load(open(MAN))
    assert m["csrc_sha256"] == csrc_digest(), \
        "lightzero_amd/csrc or include/lz_mi355.h changed after profiles/r06_* were measured: re-run tools/profile_run.sh + tools/refresh_profiles.py"
    for f in m["files"]:
        assert os.path.isfile(os.path.join(ROOT, "profiles", f)), f
    par = json.load(open(os.path.join(ROOT, "profiles", "r06_parity.json")))
    for k, v in par["worst_over_everything"].items():
        assert v < par["bounds"][k], (k, v)
