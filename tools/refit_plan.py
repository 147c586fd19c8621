"""Re-fit the split planner's shipped tables (lotus_amd/plan_tables.json) from a bench.py line.
usage: python tools/refit_plan.py gpurun_out/<tag>/bench.json [label]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lotus_amd import plan

path = sys.argv[1]
line = [ln for ln in open(path).read().splitlines() if ln.startswith("{")][-1]
line = json.loads(line)
label = sys.argv[2] if len(sys.argv) > 2 else os.path.basename(os.path.dirname(path)) or path
t = plan.tables_from_bench(line, f"bench.py line {label} (csrc {line['roofline'].get('csrc_sha', '?')}, one MI355X of the pool, d = 768, k = 10)")
json.dump(t, open(plan.TABLES_PATH, "w"), indent=1)
print(json.dumps(t, indent=1))
for w in (2, 4, 8):
    plan.use_tables(t)
    print(w, "GPUs ->", plan.pick_split(w), {f"{gq}x{gc}": round(plan.projected_fraction(1e5 / gq, 1e6 / gc), 4) for gq, gc in plan.splits(w)})
