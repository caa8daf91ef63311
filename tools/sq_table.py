"""Compact per-kernel table from a gpu_pmc_bench.sh counter dump (two rocprofv3 --pmc passes of SQ counters).
    python tools/sq_table.py gpurun_out/<tag>_sq.txt > profiles/<tag>_mfma_utilisation.txt
MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), kernel cycles = SQ_BUSY_CYCLES / 32 (the counter is
summed over the 32 shader engines; SQ_VALU_MFMA_BUSY_CYCLES counts cycles over all SIMDs: MI355X_MICROARCH.md)."""
import re, sys
d = {}
for block in re.split(r"\n(?=\S)", open(sys.argv[1]).read()):
    lines = block.strip().split("\n")
    if not lines or not lines[0].strip():
        continue
    name = lines[0].split("  (")[0]
    m = d.setdefault(name, {})
    mm = re.search(r"dispatches (\d+)", lines[0])
    if mm:
        m["dispatches"] = int(mm.group(1))
    for l in lines[1:]:
        p = l.split()
        if len(p) >= 2:
            try:
                m[p[0]] = float(p[1])
            except ValueError:
                pass
rows = []
for k, m in d.items():
    if not m.get("SQ_BUSY_CYCLES") or "SQ_INSTS_VALU" not in m:
        continue
    cyc = m["SQ_BUSY_CYCLES"] / 32.0
    mf = m.get("SQ_INSTS_MFMA", 0.0)
    rows.append((cyc * m.get("dispatches", 1), k, m.get("dispatches", 0), cyc, 100 * m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * cyc),
                 100 * m.get("SQ_LDS_BANK_CONFLICT", 0) / max(m.get("SQ_LDS_IDX_ACTIVE", 0), 1), 100 * m.get("SQ_LDS_IDX_ACTIVE", 0) / (256 * cyc),
                 m["SQ_INSTS_VALU"] / mf if mf else float("nan"), 100 * m.get("SQ_WAIT_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1),
                 100 * m.get("SQ_WAIT_INST_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1)))
rows.sort(reverse=True)
print(f"# {sys.argv[1]}: per dispatch means; sorted by total cycles.  MFMA% = matrix-pipe busy share of all SIMD-cycles; LDSbusy% = LDS-active")
print("# share of CU-cycles; confl% = bank-conflict share of LDS-active cycles; wait% / stall% = SQ_WAIT_ANY / SQ_WAIT_INST_ANY of wave-cycles")
print(f"{'kernel':78s} {'disp':>5s} {'kcycles':>9s} {'MFMA%':>6s} {'LDSbusy%':>8s} {'confl%':>6s} {'VALU/MFMA':>9s} {'wait%':>6s} {'stall%':>6s}")
for _, k, n, cyc, mu, cf, lb, vm, w, st in rows[:40]:
    print(f"{k[:78]:78s} {n:5d} {cyc / 1e3:9.1f} {mu:6.1f} {lb:8.1f} {cf:6.1f} {vm:9.1f} {w:6.1f} {st:6.1f}")
