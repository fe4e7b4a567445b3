"""Dev tool: latency of the 3-D fragments mode (fragments_in_xy = false) on one 128^3 block with context, flood on the host
against flood on the device."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import NET_CONFIG, OUT_BLOCK, CONTEXT
from bootstrapper_amd.unet import Model, extract_block_reflect
from bootstrapper_amd.synth import synthetic_state_dict, synthetic_volume
from bootstrapper_amd.post.engine import SegEngine

dev = torch.device("cuda", 0)
m = Model(NET_CONFIG, precision="bf16").load_state_dict(synthetic_state_dict(NET_CONFIG, 0))
vol = synthetic_volume((512,) * 3, seed=0, device=dev)
in_block = tuple(o + 2 * c for o, c in zip(OUT_BLOCK, CONTEXT))
affs = m.predict_u8(extract_block_reflect(vol, [-CONTEXT[0], -CONTEXT[1], -CONTEXT[2]], in_block))[0][:3].contiguous()
res = {}
for name, host in (("host", True), ("device", False)):
    eng = SegEngine(OUT_BLOCK, 0, host_flood=host)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        frags, mx = eng.ws_fragments(affs, False, 10)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    res[name] = frags
    print(f"3-D fragments of a 128^3 block, flood on the {name}: {dt:.2f} s, {int(mx.item())} fragments")
print("equal:", bool(torch.equal(res["host"], res["device"])))
