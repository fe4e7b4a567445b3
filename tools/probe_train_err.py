"""Dev probe: gradient error of the training step against a reference golden, per split-bf16 component
(forward / input gradients / weight gradients switched on one at a time through the BSMI_*_X3 dev knobs)."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bootstrapper_amd.unet import Model
from bootstrapper_amd.training import Trainer
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_train_gpu import _net_config
tag = sys.argv[1] if len(sys.argv) > 1 else "mtlsd_f4i2"
d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", f"train_{tag}.npz"))
meta = json.loads(bytes(d["config"]).decode())
sd = {k[3:]: d[k] for k in d.files if k.startswith("w0:")}
for name, env in (("f32", None), ("fwd", "F"), ("dgrad", "D"), ("wgrad", "W"), ("all", "FDW")):
    for k, c in (("BSMI_FWD_X3", "F"), ("BSMI_DGRAD_X3", "D"), ("BSMI_WGRAD_X3", "W")):
        os.environ[k] = "1" if (env and c in env) else "0"
    m = Model(_net_config(meta), precision="f32").load_state_dict(sd)
    tr = Trainer(m, meta["in_shape"], lr=meta["lr"], arithmetic="f32" if env is None else "split-bf16")
    nh = len(m.heads)
    raw = torch.from_numpy(d["x"]).cuda()
    targets = [torch.from_numpy(d[f"gt{i}"][0]).cuda() for i in range(nh)]
    weights = [torch.from_numpy(d[f"w{i}"][0]).cuda() for i in range(nh)]
    loss = tr.forward_backward(raw, targets, weights)
    errs = []
    for k in sd:
        if "g0:" + k not in d.files:
            continue
        ref = d["g0:" + k].ravel()
        got = tr.read(k, "grad")
        errs.append((float(np.abs(got - ref).max() / max(1e-6, np.abs(ref).max())), k))
    errs.sort(reverse=True)
    print(f"{name:6s} loss {loss:.8f} (ref {float(d['loss0']):.8f})  worst:", [(f"{e:.2e}", k.replace("unet.", "")) for e, k in errs[:4]])
    tr.close()
    del m
