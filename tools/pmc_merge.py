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
# provenance: digest of the kernel sources these counters were taken on (bench.py refuses a file whose digest is not
# that of the sources it runs with), steps covered, per-step total
import glob, hashlib, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
h = hashlib.sha256()
for f in sorted(glob.glob(os.path.join(root, "applecider_amd", "csrc", "*.hip")) + glob.glob(os.path.join(root, "applecider_amd", "csrc", "*.h")) +
                [os.path.join(root, "include", "applecider_hip.h")]):
    h.update(os.path.basename(f).encode())
    h.update(open(f, "rb").read())
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
out["_csrc_sha16"] = h.hexdigest()[:16]
out["_how"] = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `bench.py --steps 1 "
               "--warmup 1 --math bf16x3 --no-graph --no-branch-streams --no-h2d --no-ceilings --no-fast-mode "
               "--no-cpu-baseline`; FETCH_SIZE doubled for gfx950; %d steps incl. the roofline pass" % steps)
out["_total_GB_per_step"] = round(sum(v["fetch_GB"] + v["write_GB"] for v in out.values() if isinstance(v, dict)) / steps, 1)
json.dump(out, sys.stdout, indent=1)
