"""Dev tool: the launches of ONE training step in order, with durations, from a rocprofv3 --kernel-trace CSV of
`bench.py --mode train --train-arithmetic split-bf16 --steps N`.  usage: trace_train.py <kernel_trace.csv> [name filter]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
adam = [i for i, e in enumerate(ev) if "adam_kernel" in e[2]]
a, b = adam[-2], adam[-1]          # the last full step: after one optimizer step up to the next
step = ev[a + 1:b + 1]
print(f"step: {len(step)} launches, {(step[-1][1] - step[0][0]) / 1e6:.2f} ms wall, {sum(e - s for s, e, _ in step) / 1e6:.2f} ms in kernels")
for s, e, n in step:
    short = n.split("(")[0].replace("void bsmi::", "").replace("bsmi::", "")
    if flt in short:
        print(f"{(s - step[0][0]) / 1e6:8.3f} ms  {(e - s) / 1e3:8.1f} us  {short[:90]}")
