"""Dev probe: predict-only throughput while a side stream keeps some CUs busy with a dummy kernel."""
import ctypes as C, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import torch
sys.path.insert(0, ".")
from bootstrapper_amd.unet import Model
from bootstrapper_amd.synth import synthetic_state_dict, synthetic_volume
from tests.test_lib_cpu import AFFS_NET_CONFIG as NC
spin = C.CDLL(os.path.abspath("ab/libspin.so"))
spin.spin_launch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_uint, C.c_void_p]
m = Model(NC, precision="bf16").load_state_dict(synthetic_state_dict(NC, 0))
raw = synthetic_volume((156, 220, 220), 0)
buf = torch.randint(0, 1 << 22, (1 << 22,), dtype=torch.int32, device="cuda")
side = [torch.cuda.Stream() for _ in range(8)]
pred = torch.cuda.Stream(priority=-1)
def run(label, blocks, threads, lds, mode, nstreams):
    torch.cuda.synchronize()
    for s in side[:nstreams]:
        if blocks: spin.spin_launch(blocks, threads, lds, 400.0, mode, buf.data_ptr(), (1 << 22) - 1, s.cuda_stream)
    time.sleep(0.01)
    t0 = time.time()
    with torch.cuda.stream(pred):
        for _ in range(10): m.predict_u8(raw)
    pred.synchronize()
    dt = (time.time() - t0) / 10
    torch.cuda.synchronize()
    print(f"{label:60s} {dt*1e3:6.2f} ms/block")
run("baseline", 0, 0, 0, 0, 0)
run("8 streams x 1 WG x 64 thr, sleep loop", 1, 64, 0, 0, 8)
run("8 streams x 1 WG x 64 thr, sleep loop, 98 KB LDS", 1, 64, 98 * 1024, 0, 8)
run("8 streams x 1 WG x 64 thr, pointer chase", 1, 64, 0, 1, 8)
run("1 stream x 8 WG x 1024 thr, sleep loop, 128 KB LDS", 8, 1024, 128 * 1024, 0, 1)
run("1 stream x 8 WG x 1024 thr, pointer chase, 128 KB LDS", 8, 1024, 128 * 1024, 1, 1)
run("8 streams x 8 WG x 1024 thr, pointer chase", 8, 1024, 0, 1, 8)
run("1 stream x 32 WG x 64 thr, sleep loop", 32, 64, 0, 0, 1)
