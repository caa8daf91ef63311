"""Summarise a rocprofv3 --kernel-trace --stats CSV per train step:  python tools/prof_summary.py <kernel_stats.csv> <steps>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / steps / 1e6:.2f} ms/step, {sum(int(r['Calls']) for r in rows) / steps:.0f} launches/step")
for r in rows[:top]:
    print(f"{r['Name'][:64]:64s} calls/step {int(r['Calls']) / steps:7.1f}  avg {float(r['AverageNs']) / 1e3:9.1f} us  {int(r['TotalDurationNs']) / steps / 1e6:8.2f} ms/step  {float(r['Percentage']):5.2f}%")
