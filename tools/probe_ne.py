import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch, numpy as np
from bench import NET_CONFIG, OUT_BLOCK, CONTEXT, SEG_CONTEXT, THRESHOLDS, job_blocks_for, FILTER_FRAGMENTS, REMOVE_DEBRIS
from bootstrapper_amd.unet import Model
from bootstrapper_amd.synth import synthetic_state_dict, synthetic_volume
from bootstrapper_amd.volume import VolumePipeline
m = Model(NET_CONFIG, precision="bf16x3").load_state_dict(synthetic_state_dict(NET_CONFIG, 0))
vol = synthetic_volume((1024,) * 3, seed=0, device=torch.device("cuda", 0))
pipe = VolumePipeline(m, OUT_BLOCK, CONTEXT, job_blocks_for(20), SEG_CONTEXT, THRESHOLDS, n_lanes=16, min_seed_distance=10,
                      filter_fragments=FILTER_FRAGMENTS, remove_debris=REMOVE_DEBRIS)
pipe.run(vol)
c = pipe.seg.counts_dev.cpu().numpy()
print("counts per block (edges, ...):", c[:, 0].tolist())
print("nums per block:", pipe.seg.block_nums.tolist())
