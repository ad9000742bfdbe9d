"""Print value / ms_per_step of the last JSON line on stdin (helper for shell loops around bench.py)."""
import json
import sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(" ".join(sys.argv[1:]), d["value"], d["ms_per_step"])
