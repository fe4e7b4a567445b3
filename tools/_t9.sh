set -e
mkdir -p gpurun_out/r03/plain2
timeout -k 10 600 python -m pytest tests/test_seg_gpu.py tests/test_volume_gpu.py -x -q -m gpu -W error::RuntimeWarning > gpurun_out/r03/plain2/t.log 2>&1 || { tail -n 30 gpurun_out/r03/plain2/t.log; exit 1; }
tail -n 2 gpurun_out/r03/plain2/t.log
timeout -k 10 300 python tools/probe_tail.py 20 20 > gpurun_out/r03/plain2/ev20.log 2>&1
timeout -k 10 300 python tools/probe_tail.py 5 5 > gpurun_out/r03/plain2/ev5.log 2>&1
timeout -k 10 300 python tools/probe_tail.py 64 20 > gpurun_out/r03/plain2/ev64.log 2>&1
grep -h "floods end" gpurun_out/r03/plain2/ev20.log gpurun_out/r03/plain2/ev5.log gpurun_out/r03/plain2/ev64.log
F="--no-cpu-baseline --no-modes --no-train --no-drivers"
timeout -k 10 200 python bench.py --steps 20 --warmup 5 $F > gpurun_out/r03/plain2/s20.log 2>&1
timeout -k 10 200 python bench.py $F > gpurun_out/r03/plain2/s64.log 2>&1
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03/plain2/s*.log')):
    for l in open(f):
        if l.startswith('{"metric"'):
            d=json.loads(l); print(f.split('/')[-1], round(d['value'],2), round(d['predict_only']['seconds'],4), round(d['segment_only']['seconds'],4), round(d['roofline']['frac'],3))
P
