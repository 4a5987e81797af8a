"""Kernel statistics from a rocprofv3 rocpd database (the default output format of ROCm 7.2):
per-kernel calls / total / average / share, as `--stats` prints them, written as CSV.
Usage: python tools/rocpd_stats.py run_results.db [steps] > kernel_stats.csv"""
import sqlite3, sys, collections, re

db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
names = {r[0]: r[1] for r in cur.execute(f"select id, kernel_name from {ks}")}
agg = collections.defaultdict(lambda: [0, 0, 10**18, 0])
for kid, s, e in cur.execute(f"select kernel_id, start, end from {kd}"):
    a = agg[names.get(kid, str(kid))]
    d = e - s
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
tot = sum(a[1] for a in agg.values())
print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","MsPerStep"')
for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    short = re.sub(r"\(anonymous namespace\)::", "", n)
    short = short.split("(")[0] if "<" not in short else short[:short.index(">") + 1]
    print(f'"{short}",{a[0]},{a[1]},{a[1] / a[0]:.0f},{100 * a[1] / tot:.2f},{a[2]},{a[3]},{a[1] / 1e6 / steps:.3f}')
print(f'"TOTAL",{sum(a[0] for a in agg.values())},{tot},,100,,,{tot / 1e6 / steps:.3f}')
