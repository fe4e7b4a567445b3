"""BASELINE-size checks (one 128^3 output block of the full 3d_affs network, 94.7 M parameters):
bf16 throughput mode against the f32 parity mode of the same engine, run-to-run determinism,
the opt-in halo kernel, and bit-exact segmentation against the oracle at full block size."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def full_block():
    from bootstrapper_amd.unet import Model
    from bootstrapper_amd.synth import synthetic_state_dict, synthetic_volume
    from tests.test_lib_cpu import AFFS_NET_CONFIG as NC
    sd = synthetic_state_dict(NC, 0)
    raw = synthetic_volume((156, 220, 220), 0)
    m = Model(NC, precision="bf16").load_state_dict(sd)
    u8_a, f_a = m.predict_u8(raw, want_f32=True)
    u8_b, f_b = m.predict_u8(raw, want_f32=True)
    m.set_precision("f32")
    u8_f, f_f = m.predict_u8(raw, want_f32=True)
    m.set_precision("bf16x3")
    u8_s, f_s = m.predict_u8(raw, want_f32=True)
    u8_t, f_t = m.predict_u8(raw, want_f32=True)
    torch.cuda.synchronize()
    return dict(bf16=(u8_a[0], f_a[0]), bf16_again=(u8_b[0], f_b[0]), f32=(u8_f[0], f_f[0]),
                bf16x3=(u8_s[0], f_s[0]), bf16x3_again=(u8_t[0], f_t[0]))


def test_full_block_shapes_and_determinism(full_block):
    u8, f = full_block["bf16"]
    assert tuple(u8.shape) == (6, 128, 128, 128) and u8.dtype == torch.uint8
    assert torch.equal(u8, full_block["bf16_again"][0]) and torch.equal(f, full_block["bf16_again"][1])
    assert torch.equal(full_block["bf16x3"][1], full_block["bf16x3_again"][1])
    # the synthetic weights give structured affinities (needed for a meaningful segmentation stage)
    assert 20 < float(u8.float().std()) < 90


def test_full_block_vs_cpu_oracle(full_block):
    """The whole 3d_affs network on one BASELINE-size block against the CPU restatement (oracle/unet_ref.py, about
    15 s on the box's cores): exact-f32 MFMA mode and the split-bf16 mode (bf16x3, the benchmarked one) within the
    1e-4 parity gate of the north star -- this is the test that sees the persistent split-K launches, the 8-wave
    kernels and K up to 48 600 at their real sizes -- and the bf16 throughput mode within 6e-3 (measured 4.3e-3) of
    the same reference."""
    from oracle import unet_ref as R
    from bootstrapper_amd.synth import synthetic_state_dict, synthetic_volume
    from tests.test_lib_cpu import AFFS_NET_CONFIG as NC
    raw = synthetic_volume((156, 220, 220), 0).cpu().numpy()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    ref = torch.from_numpy(R.predict_block(R.default_cfg(12, 5), synthetic_state_dict(NC, 0), raw, ["affs_head"])[0])
    d32 = (full_block["f32"][1].cpu() - ref).abs()
    d16 = (full_block["bf16"][1].cpu() - ref).abs()
    dx3 = (full_block["bf16x3"][1].cpu() - ref).abs()
    print("f32 vs oracle: max", float(d32.max()), "mean", float(d32.mean()), "| bf16x3 vs oracle: max", float(dx3.max()), "mean",
          float(dx3.mean()), "| bf16 vs oracle: max", float(d16.max()), "mean", float(d16.mean()))
    assert float(d32.max()) < 1e-4
    assert float(dx3.max()) < 1e-4
    assert float(d16.max()) < 6e-3 and float(d16.mean()) < 1e-3
    # u8 outputs of the split mode: a 1e-5 float difference can flip the truncation at an integer boundary
    dus = (full_block["bf16x3"][0].int() - full_block["f32"][0].int()).abs()
    assert int(dus.max()) <= 1 and float((dus == 0).float().mean()) > 0.995
    du = (full_block["bf16"][0].int() - full_block["f32"][0].int()).abs()
    assert int(du.max()) <= 3 and float((du <= 1).float().mean()) > 0.95


def test_full_block_segmentation_bit_exact(full_block):
    from bootstrapper_amd.post.engine import SegEngine
    from oracle import seg_ref as S
    affs = full_block["bf16"][0][:3].contiguous()
    eng = SegEngine((128, 128, 128))
    frags, mx = eng.ws_fragments(affs, True, 10)
    segs = eng.agglomerate_mean(affs, frags, [0.2, 0.35, 0.5])
    eng.status()
    a = affs.cpu().numpy()
    ref_frags, ref_max = S.ws_fragments_u8(a, True, 10)
    assert int(mx.item()) == ref_max
    assert np.array_equal(frags.cpu().numpy().astype(np.uint64), ref_frags)
    ref = S.agglomerate_mean_u8(a, ref_frags, [0.2, 0.35, 0.5])
    got = segs.cpu().numpy().astype(np.uint64)
    for t in range(3):
        assert np.array_equal(got[t], ref[t])
    assert len(np.unique(ref[2])) < len(np.unique(ref_frags)) / 2


def test_full_block_blockwise_stages_bit_exact(full_block):
    """One blockwise block at BASELINE size: read box 160^3 (128^3 write + 16 context) built by
    mirror-tiling the predicted affinities; fragment clean-up, 26-connected relabel and RAG edge scoring
    against the oracle, with the timings the DESIGN.md table quotes."""
    import time
    from bootstrapper_amd.post.engine import SegEngine
    from oracle import seg_ref as S
    a128 = full_block["bf16"][0][:3]
    affs = torch.nn.functional.pad(a128[None].float(), (16, 16, 16, 16, 16, 16), mode="reflect")[0].to(torch.uint8).contiguous()
    assert tuple(affs.shape) == (3, 160, 160, 160)
    eng = SegEngine((160, 160, 160))
    nv = 128 ** 3
    t = {}

    def timed(name, fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        t[name] = (time.perf_counter() - t0) * 1e3
        return out
    frags, _ = timed("ws_fragments", lambda: eng.ws_fragments(affs, True, 10))
    frags_raw = frags.clone()
    labels, num = timed("postprocess", lambda: eng.postprocess_fragments(affs, frags, 0.1, 64, (16, 16, 16), (128, 128, 128), 5 * nv))
    eng.status()
    # the neighbours' fragments are not available in a one-block test: score the RAG of the read box's own fragments
    frags_read, _ = eng.postprocess_fragments(affs, frags_raw, 0.1, 64, (0, 0, 0), (160, 160, 160), 5 * nv)
    edges, scores = timed("rag_merge_scores", lambda: eng.rag_merge_scores(affs, frags_read, 1.0, 256))
    print("blockwise stage timings (ms):", {k: round(v, 2) for k, v in t.items()}, "fragments", int(num.item()), "edges", len(edges))

    a = affs.cpu().numpy()
    ref_frags, _ = S.ws_fragments_u8(a, True, 10)
    ref_f = S.filter_fragments_u8(a, ref_frags, 0.1, 64)
    ref_lab, ref_n = S.label26(np.ascontiguousarray(ref_f[16:144, 16:144, 16:144]))
    assert int(num.item()) == ref_n and ref_n > 1000
    got = labels.cpu().numpy().view(np.uint64)
    assert np.array_equal(got, np.where(ref_lab > 0, ref_lab.astype(np.uint64) + np.uint64(5 * nv), np.uint64(0)))
    ref_read, _ = S.label26(ref_f)
    ref_read = np.where(ref_read > 0, ref_read.astype(np.uint64) + np.uint64(5 * nv), np.uint64(0))
    assert np.array_equal(frags_read.cpu().numpy().view(np.uint64), ref_read)
    e_ref, s_ref, _, _ = S.rag_merge_scores_u8(a, ref_read, 1.0, 256)
    assert np.array_equal(edges.cpu().numpy().view(np.uint64), e_ref)
    assert np.array_equal(scores.cpu().numpy().view(np.uint32), s_ref.view(np.uint32))


def test_full_net_volume_pipeline_vs_cpu_blockwise():
    """The full 94.7 M-parameter net through VolumePipeline (the thing bench.py times) on a box of 2 x 2 x 2 blocks of 128^3
    in the default precision: the slab's affinities are the blocks the model predicts one by one, and the fragments and the
    three stitched segmentations are bit-equal to the CPU composition (oracle/blockwise_ref.cpu_blockwise) on those
    affinities -- context reads across block faces, block ids, edge ownership and the global components at BASELINE size."""
    from bootstrapper_amd.unet import Model, extract_block_reflect
    from bootstrapper_amd.synth import synthetic_state_dict, synthetic_volume
    from bootstrapper_amd.volume import VolumePipeline
    from oracle.blockwise_ref import cpu_blockwise
    from tests.test_lib_cpu import AFFS_NET_CONFIG as NC
    m = Model(NC).load_state_dict(synthetic_state_dict(NC, 0))                  # default precision: bf16x3
    vol = synthetic_volume((256, 256, 256), 0)
    thr = [0.2, 0.35, 0.5]
    pipe = VolumePipeline(m, (128, 128, 128), (14, 46, 46), (2, 2, 2), (16, 16, 16), thr, min_seed_distance=10, filter_fragments=0.1,
                          remove_debris=64, n_lanes=8)
    segs = pipe.run(vol)
    affs = pipe.seg.interior(pipe.seg.affs).cpu().numpy()
    for (z, y, x) in [(0, 0, 0), (1, 0, 1), (1, 1, 1)]:
        off = [128 * z - 14, 128 * y - 46, 128 * x - 46]
        blk = m.predict_u8(extract_block_reflect(vol, off, (156, 220, 220)))[0][:3].cpu().numpy()
        assert np.array_equal(affs[:, 128 * z:128 * z + 128, 128 * y:128 * y + 128, 128 * x:128 * x + 128], blk), (z, y, x)
    frags_ref, nodes, E, Sc, segs_ref = cpu_blockwise(affs, (128, 128, 128), (16, 16, 16), 10, 0.1, 64, thr, 256,
                                                      workers=min(16, os.cpu_count() or 1))
    assert len(nodes) > 5000 and len(E) > len(nodes)
    assert np.array_equal(pipe.seg.interior(pipe.seg.frags).cpu().numpy().view(np.uint64), frags_ref)
    assert np.array_equal(pipe.seg.nodes, nodes)
    for t in range(len(thr)):
        assert np.array_equal(segs[t].cpu().numpy().view(np.uint64), segs_ref[t]), thr[t]
    counts = [len(np.unique(s)) for s in segs_ref]
    assert counts[0] >= counts[1] >= counts[2] and counts[2] < len(nodes)
    # predict lanes: the same job with two engines whose forward passes overlap (block k on engine k mod 2) -- same bits
    pipe2 = VolumePipeline([m, m.clone()], (128, 128, 128), (14, 46, 46), (2, 2, 2), (16, 16, 16), thr, min_seed_distance=10,
                           filter_fragments=0.1, remove_debris=64, n_lanes=8)
    segs2 = pipe2.run(vol)
    assert torch.equal(pipe2.seg.interior(pipe2.seg.affs), pipe.seg.interior(pipe.seg.affs))
    assert torch.equal(segs2, segs)


@pytest.mark.parametrize("variant,env", [
    ("raster-halo kernel", {"BSMI_USE_RH": "1"}),
    ("raster-halo kernel, persistent split-K tail", {"BSMI_USE_RH": "1", "BSMI_SK_GRID": "8", "BSMI_TILE_EFF": "0.01,0.01,0.01,1,0.01"}),
    ("persistent split-K tail, 256x256 tiles", {"BSMI_SK_GRID": "8", "BSMI_TILE_EFF": "0.01,0.01,0.01,1,0.01"}),
    ("persistent split-K tail, 256x320 tiles", {"BSMI_SK_GRID": "16", "BSMI_TILE_EFF": "0.01,0.01,0.01,0.01,1"}),
    ("4-wave kernels", {"BSMI_WAVES8": "0"}),
    ("split-bf16 as listed K-steps everywhere", {"BSMI_X3_FUSED": "0"}),
    ("fused split-bf16 kernel on 256x320 tiles, split-K tails forced", {"BSMI_X3_FUSED": "2", "BSMI_SK_GRID": "16", "BSMI_TILE_EFF": "0.01,0.01,0.01,0.01,1"}),
    ("fused and listed split-bf16 layers in one forward", {"BSMI_X3_FUSED": "3"}),
    ("fused split-bf16 kernel on 256x160 tiles", {"BSMI_TILE_EFF": "0.01,0.01,1,0.01,0.01"}),
    ("implicit GEMM for the first pass and the small-Cout layers (no first_pass / conv_box)", {"BSMI_FUSED_FIRST": "0", "BSMI_USE_BOX": "0"}),
    ("Winograd F(2x2,3x3) form on every 3x3x3 stage of the split mode, upsampling fused into its consumers", {"BSMI_WINO": "2"}),
    ("Winograd on every stage, upsampled maps materialised", {"BSMI_WINO": "2", "BSMI_FUSE_UP": "0"}),
    ("Winograd on every stage, batched persistent launches with split-K tails (256x256 tiles)",
     {"BSMI_WINO": "2", "BSMI_SK_GRID": "8", "BSMI_TILE_EFF": "0.01,0.01,0.01,1,0.01"}),
    ("Winograd on every stage, batched persistent launches with split-K tails (256x320 tiles)",
     {"BSMI_WINO": "2", "BSMI_SK_GRID": "16", "BSMI_TILE_EFF": "0.01,0.01,0.01,0.01,1"}),
    ("no Winograd stage", {"BSMI_WINO": "0"}),
    ("Winograd F(4x4,3x3) tiles (round 4) on every 3x3x3 stage: overhanging last tiles, upsampling fused into the 6x6 input transform",
     {"BSMI_WINO": "2", "BSMI_WINO4": "2"}),
    ("F(4x4) on every stage, upsampled maps materialised", {"BSMI_WINO": "2", "BSMI_WINO4": "2", "BSMI_FUSE_UP": "0"}),
    ("F(4x4) on every stage, batched persistent launches with split-K tails", {"BSMI_WINO": "2", "BSMI_WINO4": "2", "BSMI_SK_GRID": "8", "BSMI_TILE_EFF": "0.01,0.01,0.01,1,0.01"}),
    ("no F(4x4) stage: F(2x2) where the default rule has Winograd stages", {"BSMI_WINO4": "0"}),
    ("Winograd weight images transformed on the host (the device transform's cross-check), every stage F(4x4)", {"BSMI_WINO_PACK_HOST": "1", "BSMI_WINO": "2"}),
    ("... and F(2x2)", {"BSMI_WINO_PACK_HOST": "1", "BSMI_WINO": "2", "BSMI_WINO4": "0"}),
    ("halo-resident form (conv_h16.hip) on every stage of at most 64 output channels", {"BSMI_H16": "2"}),
    ("halo-resident form on every such stage, the first ConvPass included", {"BSMI_H16": "2", "BSMI_FUSED_FIRST": "0", "BSMI_WINO": "0"}),
    ("no halo-resident stage", {"BSMI_H16": "0"}),
])
def test_conv_kernel_variants_in_subprocess(variant, env):
    """The conv kernel variants that the golden nets do not reach by themselves (they are chosen by layer
    size, or opt-in) are selected through environment variables read once per process: run the whole U-Net
    parity module in a child process per variant."""
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_unet_gpu.py"), "-x", "-q"],
                       env=dict(os.environ, **env), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, f"{variant}:\n" + r.stdout[-3000:] + r.stderr[-2000:]


def test_halo_resident_kernel_takes_the_narrow_stages_in_subprocess():
    """The halo-resident kernel (conv_h16.hip) takes five of the six stages with at most 64 output channels of the full net by
    the default rule: the full-size parity test again with the plan printed, and once more with the form switched off."""
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_fullsize_gpu.py"), "-x", "-q", "-s", "-k",
                        "full_block_vs_cpu_oracle"], env=dict(os.environ, BSMI_PLAN_DEBUG="1"), capture_output=True, text=True,
                       timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert (r.stderr + r.stdout).count("halo-resident") >= 5
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_fullsize_gpu.py"), "-x", "-q", "-s", "-k",
                        "full_block_vs_cpu_oracle"], env=dict(os.environ, BSMI_PLAN_DEBUG="1", BSMI_H16="0"), capture_output=True, text=True,
                       timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "halo-resident" not in r.stderr + r.stdout


def test_merge_loop_general_form_in_subprocess():
    """The merge loop has a FAST form for graphs that fit the LDS (every test graph does) and a general one (queue overflow
    in HBM, tie look-ups, flags in HBM): run the segmentation parity module with the FAST form switched off."""
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_seg_gpu.py"), "-x", "-q"],
                       env=dict(os.environ, BSMI_AGG_FAST="0"), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_fullsize_gpu.py"), "-x", "-q", "-k", "seg or blockwise"],
                       env=dict(os.environ, BSMI_AGG_FAST="0"), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode in (0, 5), r.stdout[-3000:] + r.stderr[-2000:]


def test_cremi_single_block_96_predict_and_segment():
    """BASELINE config 1: one 96^3 output block of the full 3d_affs network ((124,188,188) in), predict + waterz-style
    segment, against the CPU plumbing case: affinities of the split-bf16 mode within 1e-4 of the torch-CPU oracle, and
    fragments + agglomeration on them bit-exact against the C restatement."""
    from bootstrapper_amd.unet import Model
    from bootstrapper_amd.post.engine import SegEngine
    from bootstrapper_amd.synth import synthetic_state_dict, synthetic_volume
    from oracle import unet_ref as R
    from oracle import seg_ref as S
    from tests.test_lib_cpu import AFFS_NET_CONFIG as NC
    sd = synthetic_state_dict(NC, 0)
    raw = synthetic_volume((124, 188, 188), 3)
    m = Model(NC, precision="bf16x3").load_state_dict(sd)
    assert m.output_shape((124, 188, 188)) == (96, 96, 96)
    u8, f32 = m.predict_u8(raw, want_f32=True)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    ref = R.predict_block(R.default_cfg(12, 5), sd, raw.cpu().numpy(), ["affs_head"])[0]
    err = float((f32[0].cpu() - torch.from_numpy(ref)).abs().max())
    print("96^3 block, bf16x3 vs CPU oracle: max abs err", err)
    assert err < 1e-4
    affs = u8[0][:3].contiguous()
    eng = SegEngine((96, 96, 96))
    frags, mx = eng.ws_fragments(affs, True, 10)
    segs = eng.agglomerate_mean(affs, frags, [0.2, 0.35, 0.5])
    eng.status()
    a = affs.cpu().numpy()
    ref_frags, ref_max = S.ws_fragments_u8(a, True, 10)
    assert int(mx.item()) == ref_max and np.array_equal(frags.cpu().numpy().astype(np.uint64), ref_frags)
    for t, r in enumerate(S.agglomerate_mean_u8(a, ref_frags, [0.2, 0.35, 0.5])):
        assert np.array_equal(segs[t].cpu().numpy().astype(np.uint64), r)


def test_two_engines_of_one_process_side_by_side():
    """Two handles predicting on two streams at the same time, really overlapping (no forward chain): each prediction is what it
    is alone, bit for bit.  Round 4, DESIGN.md section 5: with a head kernel that kept its channels in a scratch segment, nearly
    every prediction of this test was wrong."""
    assert os.environ.get("BSMI_FORWARD_CHAIN") != "1"
    import threading
    from bootstrapper_amd.unet import Model
    from bootstrapper_amd.synth import synthetic_state_dict, synthetic_volume
    from tests.test_lib_cpu import AFFS_NET_CONFIG as NC
    sd = synthetic_state_dict(NC, 0)
    models = [Model(NC, precision="bf16x3").load_state_dict(sd) for _ in range(2)]
    raw = synthetic_volume((156, 220, 220), 0)
    torch.cuda.synchronize()
    ref = models[0].predict_u8(raw)[0].clone()
    torch.cuda.synchronize()
    assert torch.equal(ref, models[1].predict_u8(raw)[0])
    streams = [torch.cuda.Stream() for _ in models]
    bad = [0, 0]

    def work(i):
        torch.cuda.set_device(0)
        with torch.cuda.stream(streams[i]):
            for _ in range(40):
                u = models[i].predict_u8(raw)[0]
                streams[i].synchronize()
                bad[i] += not torch.equal(u, ref)
    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert bad == [0, 0], bad
