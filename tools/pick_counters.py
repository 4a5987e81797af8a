"""Filters a wish list of PMC counter names against `rocprofv3 -L` output (a pass with an unknown name fails whole).
Usage: python tools/pick_counters.py counters_list.txt NAME NAME ...  -> prints the available ones, space separated"""
import re
import sys
txt = open(sys.argv[1]).read()
have = set(re.findall(r"\b([A-Z][A-Za-z0-9_]+)\b", txt))
print(" ".join(n for n in sys.argv[2:] if n in have))
