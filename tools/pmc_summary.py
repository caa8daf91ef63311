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
    if m.get("SQ_BUSY_CYCLES") and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
        # SQ_BUSY_CYCLES is summed over the 32 shader engines, SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs (cycles, not quad-cycles:
        # MI355X_MICROARCH.md) -> share of SIMD-cycles with the matrix pipe busy while the kernel runs
        print(f"    {'MFMA utilisation (derived)':28s} {100 * m['SQ_VALU_MFMA_BUSY_CYCLES'] / (32 * m['SQ_BUSY_CYCLES']):15.1f}%  = MFMA_BUSY / (1024 SIMDs x SQ_BUSY/32)")
    if m.get("SQ_LDS_IDX_ACTIVE"):
        print(f"    {'LDS bank-conflict share':28s} {100 * m.get('SQ_LDS_BANK_CONFLICT', 0.0) / m['SQ_LDS_IDX_ACTIVE']:15.1f}%  of LDS-active cycles")
    for k, v in sorted(m.items()):
        extra = f"  {100 * v / wc:6.1f}% of WAVE_CYCLES" if wc and k.startswith("SQ_") and k != "SQ_WAVE_CYCLES" and "MFMA_BUSY" not in k and "LDS_" not in k and "WAVES" not in k else ""
        print(f"    {k:28s} {v:16.0f}{extra}")
