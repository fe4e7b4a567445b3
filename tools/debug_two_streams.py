"""Dev tool: two engines of ONE process predicting at the same time on two streams: is each prediction still what it is alone?"""
import os, sys, threading, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from bootstrapper_amd.unet import Model, extract_block_reflect
from bootstrapper_amd.synth import synthetic_state_dict, synthetic_volume
prec = os.environ.get("PREC", "bf16x3")
N = int(os.environ.get("ENGINES", "2"))
sd = synthetic_state_dict(bench.NET_CONFIG, 0)
models = []; spacers = []
for _ in range(N):
    models.append(Model(bench.NET_CONFIG, device=0, precision=prec).load_state_dict(sd))
    if os.environ.get("SPACER_MB") and not os.environ.get("SPACER_AFTER"): spacers.append(torch.zeros(int(os.environ["SPACER_MB"]) << 20, dtype=torch.uint8, device="cuda:0"))
if os.environ.get("SPACER_AFTER"): spacers.append(torch.zeros(int(os.environ["SPACER_MB"]) << 20, dtype=torch.uint8, device="cuda:0"))
if os.environ.get("SPACER_FREE"):
    spacers.clear(); torch.cuda.empty_cache()
ACTIVE = [int(x) for x in os.environ.get("ACTIVE", ",".join(map(str, range(N)))).split(",")]
vol = synthetic_volume((256, 256, 256), seed=0, device=torch.device("cuda", 0))
A = extract_block_reflect(vol, [10, 20, 30], (156, 220, 220))
torch.cuda.synchronize()
refs = []
for m in models:
    r = m.predict_u8(A)[0].clone(); torch.cuda.synchronize(); refs.append(r)
assert all(torch.equal(refs[0], r) for r in refs)
streams = [torch.cuda.Stream() for _ in range(N)]
bad = [0] * N; worst = [0] * N; cnt = [0] * N
def work(i):
    torch.cuda.set_device(0)
    with torch.cuda.stream(streams[i]):
        t = time.time()
        while time.time() - t < 15:
            u = models[i].predict_u8(A)[0]
            streams[i].synchronize()
            d = (u.int() - refs[0].int()).abs().max().item()
            cnt[i] += 1
            if d: bad[i] += 1; worst[i] = max(worst[i], d)
th = [threading.Thread(target=work, args=(i,)) for i in ACTIVE]
[t.start() for t in th]; [t.join() for t in th]
print(f"{prec}, {N} engines of one process side by side: {bad} of {cnt} predictions differ (largest u8 difference {worst})", flush=True)
