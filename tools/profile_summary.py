"""rocprofv3 run_kernel_stats.csv -> the per-step text summary kept under profiles/.
usage: profile_summary.py <run_kernel_stats.csv> <steps in the profiled run> "<header line>" > summary.txt"""
import csv, sys

path, steps, header = sys.argv[1], float(sys.argv[2]), sys.argv[3]
rows = list(csv.DictReader(open(path)))
total_ns = sum(float(r["TotalDurationNs"]) for r in rows)
calls = sum(int(r["Calls"]) for r in rows)
print("# " + header)
print(f"total kernel time {total_ns / steps / 1e6:.2f} ms/step, {calls / steps:.0f} launches/step")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    t = float(r["TotalDurationNs"])
    print(f"{r['Name'][:64]:64s} calls/step {int(r['Calls']) / steps:7.1f}  avg {float(r['AverageNs']) / 1e3:9.1f} us  "
          f"{t / steps / 1e6:8.2f} ms/step  {100 * t / total_ns:5.2f}%")
