"""One line of a bench.py JSON result: kernel table, gather parts, blend, step (stdin -> stdout).  usage: ... | python3 tools/bench_line.py NAME"""
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d["kernels"]
print(sys.argv[1] if len(sys.argv) > 1 else "-", {a: round(b["ms_per_step"], 4) for a, b in k.items()}, "parts", k["final_gather"].get("parts_ms"),
      "blend", d["roofline_blend"]["ms"], "step", d["ms_per_step"], "median", d["step_ms"]["median"])
