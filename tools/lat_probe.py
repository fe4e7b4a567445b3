"""Dev probe: duration of a pointer chase (memory latency) and of a dependent ALU chain (shader clock) right after a
heavy predict phase, after idle, and in between.
Build the probe kernels first: hipcc -O2 --offload-arch=gfx950 -shared -fPIC -o ab/liblat.so tools/lat_probe.hip"""
import ctypes as C, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import NET_CONFIG
from bootstrapper_amd.unet import Model
from bootstrapper_amd.synth import synthetic_state_dict, synthetic_volume
lat = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ab", "liblat.so"))
lat.chase_launch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
lat.alu_launch.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
dev = torch.device("cuda", 0)
m = Model(NET_CONFIG, precision="bf16x3").load_state_dict(synthetic_state_dict(NET_CONFIG, 0))
raw = synthetic_volume((156, 220, 220), 0)
n = 1 << 24
perm = np.random.default_rng(0).permutation(n).astype(np.uint32)
nxt = np.empty(n, np.uint32); nxt[perm] = np.roll(perm, -1)      # one cycle through all 64 MB
buf = torch.from_numpy(nxt.view(np.int32)).to(dev)
out = torch.zeros(4, dtype=torch.int32, device=dev)
s = torch.cuda.current_stream().cuda_stream
def probe(label):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record(); lat.chase_launch(buf.data_ptr(), 20000, out.data_ptr(), s); e[1].record()
    lat.alu_launch(2000000, out.data_ptr(), s); e[2].record(); torch.cuda.synchronize()
    print(f"{label:40s} chase 20k loads {e[0].elapsed_time(e[1]):7.2f} ms ({e[0].elapsed_time(e[1]) * 50:.0f} ns per load)   alu 2M ops {e[1].elapsed_time(e[2]):7.2f} ms")
probe("first"); probe("again")
time.sleep(0.5); probe("after 0.5 s idle")
for _ in range(12): m.predict_u8(raw)
torch.cuda.synchronize(); probe("right after 12 blocks of predict"); probe("  + one probe later"); probe("  + two probes later")
for x in (30, 100, 300):
    for _ in range(12): m.predict_u8(raw)
    torch.cuda.synchronize(); time.sleep(x * 1e-3); probe(f"predict, {x} ms sleep")
