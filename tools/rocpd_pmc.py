"""Per-kernel sums of one PMC counter from a rocprofv3 rocpd database.
Usage: python tools/rocpd_pmc.py run_results.db COUNTER  ->  JSON {kernel: {launches, sum}}"""
import sqlite3, sys, json, collections, re

db = sqlite3.connect(sys.argv[1])
want = sys.argv[2]
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
tb = lambda key: [t for t in tabs if key in t][0]
kd, ks, pe, pi = tb('kernel_dispatch'), tb('kernel_symbol'), tb('rocpd_pmc_event'), tb('rocpd_info_pmc')
pcols = [r[1] for r in cur.execute(f"pragma table_info({pe})")]
ids = [r[0] for r in cur.execute(f"select id from {pi} where name=?", (want,))]
names = {r[0]: r[1] for r in cur.execute(f"select id, kernel_name from {ks}")}
ev2k = {r[0]: r[1] for r in cur.execute(f"select event_id, kernel_id from {kd}")}
agg = collections.defaultdict(lambda: [set(), 0.0])
q = f"select event_id, pmc_id, value from {pe}"
for ev, pid, val in cur.execute(q):
    if pid in ids and ev in ev2k:
        a = agg[names[ev2k[ev]]]
        a[0].add(ev)
        a[1] += val
out = {}
for n, (evs, tot) in agg.items():
    short = re.sub(r"\(anonymous namespace\)::", "", n)
    out[short] = {"launches": len(evs), "sum": tot}
print(json.dumps(out))
