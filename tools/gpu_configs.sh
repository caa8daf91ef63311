#!/bin/bash
# ON THE GPU BOX: one bench.py line per BASELINE.json configuration (1, 2, 4, 5; config 3 is gpu_refresh.sh's `bench` part).
#   usage: bash tools/gpu_configs.sh <tag>        -> gpurun_out/<tag>_cfg<K>.json
tag=${1:-r03}
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
for k in 1 2 4 5; do
  steps=10; [ $k = 4 ] && steps=4
  timeout -k 10 400 python bench.py --config $k --steps $steps --warmup 3 > $out/${tag}_cfg$k.log 2>&1 || { tail -5 $out/${tag}_cfg$k.log; exit $k; }
  grep '^{' $out/${tag}_cfg$k.log > $out/${tag}_cfg$k.json
  python - <<PY
import json
d = json.load(open("$out/${tag}_cfg$k.json"))
r = d["roofline"]
print("config $k:", d["value"], d["unit"], d["ms_per_step"], "ms;", r["kernel"], r["bound"], r["achieved"], r["unit"], "frac", r["frac"], "; cpu", (d.get("cpu_baseline") or {}).get("value"))
PY
done
echo configs-done
