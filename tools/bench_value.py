"""Dev tool: value and predict seconds of a bench.py JSON line on stdin."""
import json, sys
j = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1])
print(sys.argv[1] if len(sys.argv) > 1 else "", round(j["value"], 2), round(j["predict_only"]["seconds"], 3), round(j.get("segment_only", {}).get("seconds", 0), 3))
