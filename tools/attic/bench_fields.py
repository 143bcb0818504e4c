#!/usr/bin/env python3
"""Reads a bench.py JSON line on stdin and prints the figures usually compared between two builds."""
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d["roofline"]["kernel_ms_per_step"]
op = d["roofline"].get("op") or {}
print(" ".join(sys.argv[1:]), "step %.3f ms" % d["ms_per_step"], "| tile %.3f block %.3f long %.3f |" % (k.get("permute_tile", 0), k.get("permute_block", 0), k.get("permute_long", 0)),
      "permute2d %.3f rcm %.3f |" % (op.get("permute2d", {}).get("ms", 0), op.get("rcm", {}).get("ms", 0)),
      "permute_apply %.3f" % ((d.get("permute_apply") or {}).get("ms_per_step", 0)), "parity", d.get("parity_on_bench_matrix"))
