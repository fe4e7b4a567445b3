"""Dev tool: standalone latency of one block's segmentation (fragments, agglomeration) on an idle GPU, on affinities
the U-Net predicts from the benchmark's synthetic volume; 1 lane, then 8 lanes side by side."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import NET_CONFIG, OUT_BLOCK, CONTEXT, THRESHOLDS
from bootstrapper_amd.unet import Model, extract_block_reflect
from bootstrapper_amd.synth import synthetic_state_dict, synthetic_volume
from bootstrapper_amd.post.engine import SegEngine

dev = torch.device("cuda", 0)
m = Model(NET_CONFIG, precision="bf16").load_state_dict(synthetic_state_dict(NET_CONFIG, 0))
vol = synthetic_volume((512,) * 3, seed=0, device=dev)
in_block = tuple(o + 2 * c for o, c in zip(OUT_BLOCK, CONTEXT))
affs = []
for i in range(8):
    raw = extract_block_reflect(vol, [128 * (i % 4) - CONTEXT[0], 128 * (i // 4) - CONTEXT[1], -CONTEXT[2]], in_block)
    affs.append(m.predict_u8(raw)[0][:3].contiguous())
torch.cuda.synchronize()
engines = [SegEngine(OUT_BLOCK, 0) for _ in range(8)]
streams = [torch.cuda.Stream(dev) for _ in range(8)]


def run(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks = []
    for i in range(n):
        with torch.cuda.stream(streams[i]):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e2 = torch.cuda.Event(enable_timing=True)
            e0.record()
            frags, _ = engines[i].ws_fragments(affs[i], True, 10)
            e1.record()
            engines[i].agglomerate_mean(affs[i], frags, THRESHOLDS)
            e2.record()
            marks.append((e0, e1, e2))
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
    ws = [a.elapsed_time(b) for a, b, _ in marks]
    ag = [b.elapsed_time(c) for _, b, c in marks]
    print(f"{n} lane(s): wall {wall:.1f} ms; fragments {min(ws):.1f}..{max(ws):.1f} ms, agglomeration {min(ag):.1f}..{max(ag):.1f} ms")


for n in (1, 1, 8, 8):
    run(n)
