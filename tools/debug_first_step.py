"""Dev tool: WHICH launch of a forward pass is the first whose output differs when another engine predicts beside it?
Engine 0 (victim) runs alone and every launch's output tensor is kept; then engine 1 (aggressor) predicts in a loop on
another stream, the victim predicts until its result differs, and its launches' outputs are compared one by one."""
import os, sys, threading, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
os.environ.setdefault("BSMI_FORWARD_CHAIN", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from bootstrapper_amd.unet import Model, extract_block_reflect
from bootstrapper_amd.synth import synthetic_state_dict, synthetic_volume
prec = os.environ.get("PREC", "bf16x3")
sd = synthetic_state_dict(bench.NET_CONFIG, 0)
victim, aggressor = [Model(bench.NET_CONFIG, device=0, precision=prec).load_state_dict(sd) for _ in range(2)]
vol = synthetic_volume((256, 256, 256), seed=0, device=torch.device("cuda", 0))
A = extract_block_reflect(vol, [10, 20, 30], (156, 220, 220))
torch.cuda.synchronize()

def dump(m):
    acts = []
    step = 0
    while True:
        try:
            acts.append(m.debug_activation(step))
        except Exception as e:
            if "out of range" in str(e): break
            acts.append(None)  # a head step: writes the caller's buffers
        step += 1
    return acts

ref = victim.predict_u8(A)[0].clone(); torch.cuda.synchronize()
clean = dump(victim)
again = victim.predict_u8(A)[0]; torch.cuda.synchronize()
assert torch.equal(ref, again)
clean2 = dump(victim)
for i, (a, b) in enumerate(zip(clean, clean2)):
    assert (a is None and b is None) or np.array_equal(a, b, equal_nan=True), f"launch {i} is not repeatable alone"
del clean2
print(f"{len(clean)} launches; alone, every launch's output repeats", flush=True)

stop = False
def loop():
    torch.cuda.set_device(0)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        while not stop:
            aggressor.predict_u8(A); s.synchronize()
th = threading.Thread(target=loop); th.start()
sv = torch.cuda.Stream()
found = 0
with torch.cuda.stream(sv):
    t0 = time.time()
    while time.time() - t0 < 60 and found < int(os.environ.get("CASES", "3")):
        u = victim.predict_u8(A)[0]; sv.synchronize()
        if torch.equal(u, ref): continue
        found += 1
        dirty = dump(victim)
        print(f"--- corrupted prediction {found}: {int((u != ref).sum())} u8 values differ", flush=True)
        first = True
        for i, (a, b) in enumerate(zip(clean, dirty)):
            if a is None: print(f"  launch {i:2d}: head"); continue
            ne = ~np.isclose(a, b, rtol=0, atol=0, equal_nan=True)
            n = int(ne.sum())
            line = f"  launch {i:2d}: shape {a.shape}: {n} of {a.size} values differ"
            if n:
                idx = np.argwhere(ne)
                lo, hi = idx.min(0), idx.max(0)
                line += f"; box z {lo[0]}..{hi[0]} y {lo[1]}..{hi[1]} x {lo[2]}..{hi[2]} c {lo[3]}..{hi[3]}; largest |d| {float(np.nanmax(np.abs(a - b)[ne])):.3g} (values up to {float(np.abs(a).max()):.3g})"
                if first:
                    first = False
                    print(line, flush=True)
                    # the pattern of the first dirty launch: voxels by flat index, channels
                    vox = np.unique(idx[:, 0] * a.shape[1] * a.shape[2] + idx[:, 1] * a.shape[2] + idx[:, 2])
                    runs = np.split(vox, np.where(np.diff(vox) != 1)[0] + 1)
                    print(f"    dirty voxels {vox.size}, in {len(runs)} runs of consecutive flat indices; first runs: " +
                          ", ".join(f"[{r[0]}..{r[-1]}] ({r.size})" for r in runs[:12]))
                    print(f"    run lengths: {sorted(set(r.size for r in runs))[:20]}; run starts mod 256: {sorted(set(int(r[0]) % 256 for r in runs))[:20]}")
                    ch = np.bincount(idx[:, 3], minlength=a.shape[3])
                    print(f"    per channel: {ch.tolist()}")
                    continue
            print(line, flush=True)
stop = True; th.join()
print(f"{found} corrupted predictions looked at", flush=True)
