"""stdin: bench.py output -> one line: ms/step, roofline fraction, option-LSTM family times"""
import json
import sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d["roofline"]
print(d["ms_per_step"], "ms/step  frac", r["frac"], {k: round(v["ms_total_per_step"], 3) for k, v in r["families"].items()})
