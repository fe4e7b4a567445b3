"""Dev tool: `bs predict` + `bs segment --ws` on an on-disk Zarr store of the whole synthetic volume (what bench.py's `drivers`
leg times), with BSMI_IO_TRACE=1: where the wall time of the two commands goes.  python tools/probe_drivers.py [edge_blocks]"""
import os, sys, time
os.environ.setdefault("BSMI_IO_TRACE", "1")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from bootstrapper_amd.synth import synthetic_state_dict, synthetic_volume
edge = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sd = synthetic_state_dict(bench.NET_CONFIG, 0)
vol = synthetic_volume((128 * edge,) * 3, seed=0, device=torch.device("cuda", 0)).cpu().numpy()
torch.cuda.empty_cache()
for rep in range(int(os.environ.get("REPS", "1"))):
    out = bench.drivers_leg(vol, sd, "bf16x3")
    print({k: v for k, v in out.items() if k != "what"}, flush=True)
