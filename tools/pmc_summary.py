"""Per-kernel means of a rocprofv3 --pmc counter_collection.csv:  python tools/pmc_summary.py <csv> [name-substring]"""
import csv, re, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(.*$", "", r["Kernel_Name"]).replace("void ", "").replace("mstg::", "").strip()
    if len(sys.argv) > 2 and sys.argv[2] not in name:
        continue
    a = acc[name][r["Counter_Name"]]
    a[0] += 1
    a[1] += float(r["Counter_Value"])
for name, cs in acc.items():
    m = {k: v[1] / v[0] for k, v in cs.items()}
    n = next(iter(cs.values()))[0]
    print(f"{name}  (dispatches {n})")
    wc = m.get("SQ_WAVE_CYCLES", 0.0)
    for k, v in sorted(m.items()):
        extra = f"  {100 * v / wc:6.1f}% of WAVE_CYCLES" if wc and k.startswith("SQ_") and k != "SQ_WAVE_CYCLES" and "MFMA_BUSY" not in k and "LDS_" not in k and "WAVES" not in k else ""
        print(f"    {k:28s} {v:16.0f}{extra}")
