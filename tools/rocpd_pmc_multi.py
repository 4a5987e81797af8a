"""Per-kernel counter table from one or more rocprofv3 rocpd databases (one per counter pass).
Usage: python tools/rocpd_pmc_multi.py out.json pass1.db pass2.db ... [--match substr,substr]
Per kernel name: launches, total / average duration (ns, from the same pass) and the SUM of every counter over
its dispatches (rocprofv3 reports one value per dispatch and counter, already summed over XCDs / SEs)."""
import collections
import json
import re
import sqlite3
import sys

args = [a for a in sys.argv[1:] if not a.startswith("--match")]
match = None
for i, a in enumerate(sys.argv):
    if a == "--match":
        match = sys.argv[i + 1].split(",")
        args = [x for x in args if x != sys.argv[i + 1]]
out_path, dbs = args[0], args[1:]
table = collections.defaultdict(lambda: {"launches": 0, "ns": 0, "counters": collections.defaultdict(float),
                                          "counter_launches": collections.defaultdict(int)})
for path in dbs:
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    tb = lambda key: [t for t in tabs if key in t][0]
    kd, ks, pe, pi = tb('kernel_dispatch'), tb('kernel_symbol'), tb('rocpd_pmc_event'), tb('rocpd_info_pmc')
    names = {r[0]: r[1] for r in cur.execute(f"select id, kernel_name from {ks}")}
    cname = {r[0]: r[1] for r in cur.execute(f"select id, name from {pi}")}
    disp = {}
    for ev, kid, s, e in cur.execute(f"select event_id, kernel_id, start, end from {kd}"):
        n = re.sub(r"\(anonymous namespace\)::", "", names.get(kid, str(kid)))
        n = n.split("(")[0] if "<" not in n else n[:n.index(">") + 1]
        if match and not any(m in n for m in match):
            continue
        disp[ev] = n
        t = table[n]
        t["launches"] += 1
        t["ns"] += e - s
    seen = collections.defaultdict(set)
    for ev, pid, val in cur.execute(f"select event_id, pmc_id, value from {pe}"):
        n = disp.get(ev)
        if n is None:
            continue
        c = cname.get(pid, str(pid))
        table[n]["counters"][c] += val
        seen[(n, c)].add(ev)
    for (n, c), evs in seen.items():
        table[n]["counter_launches"][c] += len(evs)
    db.close()
res = {}
for n, t in table.items():
    res[n] = {"launches_all_passes": t["launches"], "avg_ns": t["ns"] / max(t["launches"], 1),
              "counters": {c: {"sum": v, "launches": t["counter_launches"][c], "per_launch": v / max(t["counter_launches"][c], 1)}
                           for c, v in sorted(t["counters"].items())}}
json.dump(res, open(out_path, "w"), indent=1, sort_keys=True)
print(json.dumps({k: {"avg_us": round(v["avg_ns"] / 1e3, 1), **{c: round(x["per_launch"]) for c, x in v["counters"].items()}}
                  for k, v in res.items()}, indent=1))
