"""profiles/pmc_traffic.json: the PMC counters bench.py quotes in `roofline.traffic` / `roofline_issue` are
look-ups of committed counter passes. Every entry is stamped with the hash of the kernel sources it was
collected on (tools/pmc_table_update.py), and the line says whether that is the tree it runs on."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def test_source_hash_and_stamp_semantics():
    sha = bench.kernel_source_sha16()
    assert re.fullmatch(r"[0-9a-f]{16}", sha) and sha == bench.kernel_source_sha16()
    assert bench.counters_current({"csrc_sha16": sha}) is True
    assert bench.counters_current({"csrc_sha16": "0" * 16}) is False
    assert bench.counters_current({"hbm_bytes": 1.0}) is None and bench.counters_current(None) is None


def test_table_entries_of_the_bench_workloads_are_stamped_in_one_session():
    table = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    keys = ["eight_schools:4096:1000:16", "sv:2048:1000:64", "logistic:8192:1000:16", "radon:1024:1000:64",
            "multi_step:eight_schools:262144:32:1"]
    stamps = set()
    for k in keys:
        e = table[k]
        assert re.fullmatch(r"[0-9a-f]{16}", e["csrc_sha16"]), k
        assert e["hbm_bytes"] > 0 and os.path.exists(os.path.join(ROOT, e["source"].split(",")[0])), k
        if not k.startswith("multi_step:"):
            assert e["leapfrogs"] > 0 and e["sq"]["SQ_INSTS_VALU"] > 0, k
        stamps.add(e["csrc_sha16"])
    assert len(stamps) == 1   # one tree, one session (tools/r5_pmc_all.sh)
    # the look-up key of the driver's default line
    assert bench.traffic_entry("eight_schools", 4096, 1000, 16) == table[keys[0]]
