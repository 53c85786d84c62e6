#!/usr/bin/env python3
"""Put the counters of one PMC session into profiles/pmc_traffic.json, stamped with the hash of the
kernel sources of THIS tree (bench.kernel_source_sha16) -- run it on the tree the session was collected on:

    python tools/pmc_table_update.py <key> <profiles/dir with pmc_<kernel>.json [pmc_sq_<kernel>.json]> [--kernel nuts_kernel]

key: model:chains_per_gpu:draws:lanes for nuts_kernel, multi_step:model:chains:steps:lanes for
multi_step_kernel (profiles/README.md). bench.py reports `traffic_on_these_sources` /
`counters_on_these_sources` = (stamp == hash of the sources it runs on)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    key, src = sys.argv[1], sys.argv[2]
    kernel = sys.argv[sys.argv.index("--kernel") + 1] if "--kernel" in sys.argv else "nuts_kernel"
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    table = json.load(open(path))
    e = table.get(key, {})
    hb = json.load(open(os.path.join(src, "pmc_%s.json" % kernel)))
    e["hbm_bytes"] = hb["hbm_bytes_timed_launch"]
    sqf = os.path.join(src, "pmc_sq_%s.json" % kernel)
    if os.path.exists(sqf) and not key.startswith("multi_step:"):
        sq = json.load(open(sqf))
        e["sq"] = sq["counters"]
        e["leapfrogs"] = sq["leapfrogs_per_launch"]
    rel = os.path.relpath(src, ROOT)
    e["source"] = "%s/pmc_%s.json%s" % (rel, kernel, ", pmc_sq_%s.json" % kernel if "sq" in e and os.path.exists(sqf) else "")
    e["csrc_sha16"] = bench.kernel_source_sha16()
    table[key] = e
    json.dump(table, open(path, "w"), indent=1)
    print(key, "hbm_bytes %.4g" % e["hbm_bytes"], "sha", e["csrc_sha16"])


if __name__ == "__main__":
    main()
