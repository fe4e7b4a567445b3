"""Layer-by-layer difference of a precision mode against the f32 mode on one block of the full 3d_affs net (dev tool).
usage: probe_layers.py [D,H,W] [mode]"""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from bootstrapper_amd.unet import Model
from bootstrapper_amd.synth import synthetic_state_dict, synthetic_volume
from tests.test_lib_cpu import AFFS_NET_CONFIG as NC
shape = tuple(int(x) for x in sys.argv[1].split(",")) if len(sys.argv) > 1 else (32, 196, 196)
mode = sys.argv[2] if len(sys.argv) > 2 else "bf16x3"
sd = synthetic_state_dict(NC, 0)
raw = synthetic_volume(shape, 0)
m = Model(NC, precision="f32").load_state_dict(sd)
m.profile(True)
m.predict_u8(raw)
types = [t for t, _, _ in m.read_profile()]
ref = [m.debug_activation(i) if t != 4 else None for i, t in enumerate(types)]
m.set_precision(mode)
m.predict_u8(raw)
names = {0: "input", 1: "conv", 2: "pool", 3: "up", 4: "head"}
for i, t in enumerate(types):
    if t == 4:
        continue
    a = m.debug_activation(i)
    d = np.abs(a - ref[i])
    bad = d > 1e-3 * (np.abs(ref[i]).max() + 1e-9)
    msg = f"step {i:2d} {names[t]:5s} shape {a.shape} max|ref| {np.abs(ref[i]).max():.3e} max|diff| {d.max():.3e} bad {int(bad.sum())}"
    if bad.any():
        idx = np.argwhere(bad)
        vox = (idx[:, 0] * a.shape[1] + idx[:, 1]) * a.shape[2] + idx[:, 2]
        msg += f" | bad rows m in [{vox.min()}, {vox.max()}] ({len(np.unique(vox))} rows), channels [{idx[:,3].min()}, {idx[:,3].max()}] ({len(np.unique(idx[:,3]))})"
        msg += f" | nan {int(np.isnan(a).sum())}"
        tiles = np.unique(vox // 256)
        msg += f" | 256-row tiles hit: {len(tiles)} of {(a.shape[0]*a.shape[1]*a.shape[2]+255)//256}, first {tiles[:8].tolist()}"
    print(msg, flush=True)
    if bad.any() and "--detail" in sys.argv:
        hi, lo = m.debug_activation(i, 1), m.debug_activation(i, 2)
        r = ref[i].reshape(-1, a.shape[3]); hi = hi.reshape(r.shape); lo = lo.reshape(r.shape)
        k = np.argwhere(bad.reshape(r.shape))
        for mm, nn in k[:40]:
            print(f"   m {mm} (tile {mm // 256} row {mm % 256}) n {nn} (n%256 {nn % 256}): ref {r[mm, nn]:+.5e} hi {hi[mm, nn]:+.5e} lo {lo[mm, nn]:+.5e}")
        print("   bad per row-in-16:", np.bincount(k[:, 0] % 16, minlength=16).tolist())
        print("   bad per n%16:", np.bincount(k[:, 1] % 16, minlength=16).tolist())
        print("   bad per (n%256)//16:", np.bincount((k[:, 1] % 256) // 16, minlength=16).tolist())
        break
