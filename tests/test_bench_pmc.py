"""bench.py only uses a PMC summary collected on the kernel sources of this tree (profiles/traffic.json carries
their hash); anything else must come out as null, never as a stale number."""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stale_pmc_summary_is_refused(hmrm, tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    sha = hmrm.kernel_src_sha()
    assert len(sha) == 16
    # the committed summary: used iff its hash is this tree's
    entry, prov = bench._pmc_from_profiles("C3", sha)
    committed = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["C3"]
    if committed["kernel_src_sha"] == sha:
        assert entry is not None and prov["status"].startswith("ok") and entry["hbm_bytes_per_launch"] > 3e7
        v = entry["valu"]
        assert 0 < v["busy_cycles_weighted"] <= v["busy_cycles_upper"]
    else:
        assert entry is None and prov["status"].startswith("stale")
    # any other hash, or an unknown workload: nothing
    entry, prov = bench._pmc_from_profiles("C3", "0" * 16)
    assert entry is None and prov["status"].startswith("stale") and prov["kernel_src_sha"] == committed["kernel_src_sha"]
    entry, prov = bench._pmc_from_profiles("no such workload", sha)
    assert entry is None and "no PMC summary" in prov["status"]
