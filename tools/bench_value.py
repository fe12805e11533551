"""Print the headline numbers of a bench.py JSON line read from stdin (for quick A/B runs)."""
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
t = d.get("config", {}).get("train", {})
print(sys.argv[1] if len(sys.argv) > 1 else "", "value", round(d["value"], 1), d["unit"], "| ms/step", round(d["ms_per_step"], 3),
      "| m", t.get("m_inducing"), "graphs", t.get("hip_graphs"), "| sinkhorn", (d.get("sinkhorn") or {}).get("value"))
