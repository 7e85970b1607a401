#!/usr/bin/env python
"""rocprofv3 *_kernel_stats.csv -> markdown table (kernel names shortened).  usage: stats_to_md.py <csv> [max_rows]"""
import csv
import re
import sys


def short(name: str) -> str:
    name = re.sub(r"lsr::\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"rocprim::[^<]*detail::trampoline_kernel<rocprim::\w+::detail::wrapped_(\w+)_config", name)
    if m:
        tail = re.findall(r"lambda\(auto:1\)#(\d)", name)
        return f"rocPRIM {m.group(1)}" + (f" #{tail[-1]}" if tail else "")
    m = re.match(r"rocprim::\w+::detail::(\w+)<", name)
    if m:
        return f"rocPRIM {m.group(1)}"
    return name.split("(")[0]


rows = list(csv.DictReader(open(sys.argv[1])))
cap = int(sys.argv[2]) if len(sys.argv) > 2 else 24
print("| kernel | calls | total ms | avg us | % | min us | max us |")
print("|---|---|---|---|---|---|---|")
for r in rows[:cap]:
    print(f"| `{short(r['Name'])}` | {r['Calls']} | {int(r['TotalDurationNs']) / 1e6:.3f} | {float(r['AverageNs']) / 1e3:.2f} | "
          f"{float(r['Percentage']):.2f} | {int(r['MinNs']) / 1e3:.2f} | {int(r['MaxNs']) / 1e3:.2f} |")
