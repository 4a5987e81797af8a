"""Merge the per-kernel sums of two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; tools/rocpd_pmc.py
output) into the per-launch fabric traffic table bench.py reads (profiles/rNN_pmc_hbm_traffic_<mode>.json).
Units (MI355X_MICROARCH.md, HBM / rocprofv3): FETCH_SIZE and WRITE_SIZE count KiB at the L2's fabric side;
on gfx950 FETCH_SIZE reports half of a wide coalesced read stream, so it is doubled.
Usage: python tools/pmc_merge.py fetch.json write.json > traffic.json"""
import json, sys
fetch, write = json.load(open(sys.argv[1])), json.load(open(sys.argv[2]))
out = {}
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, {"launches": 0, "sum": 0.0}), write.get(k, {"launches": 0, "sum": 0.0})
    n = max(f["launches"], w["launches"])
    if not n:
        continue
    fb, wb = 2.0 * f["sum"] * 1024.0, w["sum"] * 1024.0
    out[k] = {"launches": n, "fetch_GB": round(fb / 1e9, 3), "write_GB": round(wb / 1e9, 3),
              "per_launch_MB": round((fb + wb) / n / 1e6, 2)}
json.dump(out, sys.stdout, indent=1)
