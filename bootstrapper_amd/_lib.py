"""ctypes binding of libbsmi.so (include/bsmi.h).  There is no fallback: if the
library is missing or a symbol is absent, importing this module raises."""
import ctypes as C
import os

# PyTorch first: it brings its own copy of the HIP runtime (same soname as /opt/rocm's), and the process must bind
# ONE of them.  Loading libbsmi.so before torch starts a second runtime, whose calls then fail with
# "no ROCm-capable device is detected" while torch's work.
import torch  # noqa: F401,E402

_HERE = os.path.dirname(os.path.abspath(__file__))
# BSMI_LIB: developer knob to A/B another build of the same library (never a fallback)
LIB_PATH = os.environ.get("BSMI_LIB") or os.path.join(_HERE, "libbsmi.so")

MAX_LEVELS, MAX_CONVS, MAX_HEADS, NAME_LEN = 8, 4, 4, 32
PREC_F32, PREC_BF16, PREC_BF16X3 = 0, 1, 2
RAW_U8, RAW_F32, RAW_U8_UNIT = 0, 1, 2
ERR_INVALID, ERR_HIP, ERR_STATE, ERR_MISSING, ERR_OVERFLOW = -1, -2, -3, -4, -5


class UNetConfig(C.Structure):
    _fields_ = [
        ("in_channels", C.c_int32),
        ("num_fmaps", C.c_int32),
        ("fmap_inc_factor", C.c_int32),
        ("num_levels", C.c_int32),
        ("downsample_factors", (C.c_int32 * 3) * MAX_LEVELS),
        ("n_convs_down", C.c_int32 * MAX_LEVELS),
        ("kernel_size_down", ((C.c_int32 * 3) * MAX_CONVS) * MAX_LEVELS),
        ("n_convs_up", C.c_int32 * MAX_LEVELS),
        ("kernel_size_up", ((C.c_int32 * 3) * MAX_CONVS) * MAX_LEVELS),
        ("num_heads", C.c_int32),
        ("head_name", (C.c_char * NAME_LEN) * MAX_HEADS),
        ("head_dims", C.c_int32 * MAX_HEADS),
        ("num_fmaps_out", C.c_int32),
    ]


class Codec(C.Structure):
    """bsmi_codec of include/bsmi_io.h"""
    _fields_ = [("id", C.c_int32), ("level", C.c_int32), ("cname", C.c_int32), ("shuffle", C.c_int32),
                ("typesize", C.c_int32), ("blocksize", C.c_int32)]


class ChunkCopy(C.Structure):
    """bsmi_chunk_copy of include/bsmi_io.h"""
    _fields_ = [("path", C.c_char_p), ("base", C.c_void_p), ("start", C.c_int64 * 4), ("extent", C.c_int64 * 4),
                ("stride", C.c_int64 * 4), ("read_modify_write", C.c_int32), ("reserved", C.c_int32)]


CODEC_RAW, CODEC_ZLIB, CODEC_GZIP, CODEC_ZSTD, CODEC_LZ4, CODEC_BLOSC = range(6)
BLOSC_LZ4, BLOSC_ZLIB, BLOSC_ZSTD = 1, 3, 4
CHUNK_MISSING = 1


class BsmiError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libbsmi error {code}: {msg}")
        self.code = code
        self.msg = msg


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C bootstrapper_amd/csrc`).  There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    p, i32, i64p, vp = C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.c_void_p
    sigs = {
        "bsmi_last_error": (C.c_char_p, []),
        "bsmi_version": (i32, []),
        "bsmi_unet_create": (i32, [C.POINTER(UNetConfig), i32, C.POINTER(p)]),
        "bsmi_unet_destroy": (i32, [p]),
        "bsmi_unet_load_weight": (i32, [p, C.c_char_p, vp, i64p, i32]),
        "bsmi_unet_finalize": (i32, [p, i32]),
        "bsmi_unet_output_shape": (i32, [p, i64p, i64p]),
        "bsmi_unet_flops": (i32, [p, i64p, C.POINTER(C.c_double)]),
        "bsmi_unet_forward": (i32, [p, i32, vp, i32, i64p, C.POINTER(vp), C.POINTER(vp), vp]),
        "bsmi_unet_debug_activation": (i32, [p, i32, i32, i64p, vp, C.c_uint64]),
        "bsmi_unet_profile_enable": (i32, [p, i32]),
        "bsmi_unet_profile_read": (i32, [p, i32, C.POINTER(i32), C.POINTER(C.c_int32), C.POINTER(C.c_double),
                                         C.POINTER(C.c_double)]),
        "bsmi_unet_profile_totals": (i32, [p, C.POINTER(C.c_double), C.POINTER(C.c_double), i64p, i32]),
        "bsmi_extract_block_reflect_u8": (i32, [vp, i64p, i64p, i64p, vp, vp]),
        "bsmi_seg_create": (i32, [i32, i64p, C.POINTER(p)]),
        "bsmi_seg_destroy": (i32, [p]),
        "bsmi_ws_fragments_u8": (i32, [p, vp, i64p, i32, i32, vp, vp, vp]),
        "bsmi_ws_fragments_seeds_u8": (i32, [p, vp, i64p, i32, i32, vp, vp, vp, vp]),
        "bsmi_rag_agglomerate_u8": (i32, [p, vp, vp, i64p, C.c_float, C.c_int, vp]),
        "bsmi_rag_edge_stats": (i32, [p, vp, vp, C.c_uint64, vp]),
        "bsmi_agglomerate_mean_u8": (i32, [p, vp, vp, i64p, C.POINTER(C.c_float), i32, vp, vp]),
        "bsmi_agglomerate_hist_u8": (i32, [p, vp, vp, i64p, C.POINTER(C.c_float), i32, i32, i32, vp, vp]),
        "bsmi_agglomerate_hist_graph": (i32, [C.c_uint32, C.c_uint32, vp, vp, vp, i32, i32, C.POINTER(C.c_float), i32, vp]),
        "bsmi_frag_postprocess_u8": (i32, [p, vp, vp, i64p, C.c_double, C.c_int64, i64p, i64p, C.c_uint64, vp, vp, vp]),
        "bsmi_label_stats": (i32, [p, vp, i64p, C.c_uint64, C.c_uint64, vp, vp, vp]),
        "bsmi_rag_merge_scores_u8": (i32, [p, vp, vp, i64p, C.c_float, C.c_int, vp, vp, C.c_uint64, vp, vp, vp, vp]),
        "bsmi_lut_relabel": (i32, [C.c_int, vp, C.c_uint64, vp, vp, C.c_uint64, vp, vp]),
        "bsmi_lut_relabel_multi": (i32, [i32, vp, C.c_uint64, vp, vp, C.c_uint64, i32, vp, vp]),
        "bsmi_connected_components": (i32, [vp, C.c_uint64, vp, vp, C.c_uint64, C.c_float, vp]),
        "bsmi_connected_components_multi": (i32, [vp, C.c_uint64, vp, vp, C.c_uint64, vp, C.c_int, vp]),
        "bsmi_cc_affs_u8": (i32, [p, vp, i64p, C.c_int, C.c_int64, vp, vp, vp, vp]),
        "bsmi_mws_agglom_f64": (i32, [C.c_int, vp, C.c_int, vp, vp, C.c_uint32, i64p, vp, vp]),
        "bsmi_mws_cluster": (i32, [C.c_uint64, vp, vp, C.c_uint64, vp]),
        "bsmi_frag_pair_affinity_u8": (i32, [C.c_int, vp, C.c_int, vp, vp, i64p, C.c_uint64, vp, vp, vp, vp, vp]),
        "bsmi_label_table_u64": (i32, [p, vp, i64p, C.c_int64, vp, vp, vp, vp, C.c_uint64, vp, vp]),
        "bsmi_seg_status": (i32, [p, vp]),
        "bsmi_seg_set_host_flood": (i32, [p, i32]),
        "bsmi_rag_graph_u8": (i32, [p, vp, vp, i64p, vp, vp, vp, C.c_uint64, vp, vp]),
        "bsmi_rag_merge_scores_host": (i32, [i32, vp, vp, vp, vp, C.c_float, i32, vp, i32]),
        "bsmi_rag_merge_scores_host_rule": (i32, [i32, vp, vp, vp, vp, C.c_float, i32, i32, vp, i32]),
        "bsmi_unet_train_set_arithmetic": (i32, [p, i32]),
        "bsmi_unet_train_set_deterministic": (i32, [p, i32]),
        "bsmi_unet_train_begin": (i32, [p, i64p]),
        "bsmi_unet_train_forward_backward": (i32, [p, vp, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_float), vp]),
        "bsmi_unet_train_num_params": (i32, [p, C.POINTER(C.c_uint64)]),
        "bsmi_unet_train_buffers": (i32, [p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
        "bsmi_unet_train_param_info": (i32, [p, C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        "bsmi_unet_train_adam_step": (i32, [p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, vp]),
        "bsmi_unet_train_read_param": (i32, [p, C.c_char_p, C.c_int, vp]),
        "bsmi_unet_train_last_loss": (i32, [p, C.POINTER(C.c_float), vp]),
        "bsmi_unet_train_prediction": (i32, [p, i32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
        "bsmi_unet_train_grad_groups": (i32, [p, i32, C.POINTER(i32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        "bsmi_unet_train_wait_grad_group": (i32, [p, i32, vp]),
        "bsmi_unet_train_write_param": (i32, [p, C.c_char_p, C.c_int, vp]),
        "bsmi_unet_train_step_count": (i32, [p, C.c_int, C.POINTER(C.c_int)]),
        "bsmi_unet_train_end": (i32, [p]),
        "bsmi_train_affinity_targets": (i32, [C.c_int, vp, vp, i64p, C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_int, C.c_float,
                                              C.c_float, vp, vp, vp]),
        "bsmi_train_lsd_targets": (i32, [C.c_int, vp, vp, i64p, i64p, i64p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, vp, vp, vp]),
        "bsmi_unet_set_persistent_grid": (i32, [p, C.c_int]),
        "bsmi_unet_profile_executed": (i32, [p, C.POINTER(C.c_double), C.c_int]),
        "bsmi_debug_lds_canary": (i32, [i32, i32, i32, vp, vp]),
        "bsmi_debug_check_guards": (i32, []),
        "bsmi_stream_create_cu_mask": (i32, [C.c_int, vp, C.c_int, C.POINTER(C.c_void_p)]),
        "bsmi_stream_destroy": (i32, [C.c_int, vp]),
        # include/bsmi_io.h
        "bsmi_codec_bound": (C.c_size_t, [C.POINTER(Codec), C.c_size_t]),
        "bsmi_codec_decode": (i32, [C.POINTER(Codec), vp, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_size_t)]),
        "bsmi_codec_encode": (i32, [C.POINTER(Codec), vp, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_size_t)]),
        "bsmi_chunks_read": (i32, [C.POINTER(Codec), i32, C.POINTER(C.c_char_p), C.POINTER(vp), C.POINTER(C.c_size_t),
                                   C.POINTER(C.c_size_t), C.POINTER(i32), i32]),
        "bsmi_chunks_write": (i32, [C.POINTER(Codec), i32, C.POINTER(C.c_char_p), C.POINTER(vp), C.POINTER(C.c_size_t),
                                    C.POINTER(i32), i32]),
        "bsmi_blosc_dev_frame_bound": (C.c_size_t, [C.c_size_t]),
        "bsmi_blosc_dev_scratch_bytes": (C.c_size_t, [i32, C.c_size_t]),
        "bsmi_blosc_encode_dev_u64": (i32, [i32, vp, C.c_int64, C.c_int64, i32, i64p, i64p, i64p, vp, C.c_size_t, vp, C.c_size_t, vp, vp]),
        "bsmi_rag_write_sqlite": (i32, [C.c_char_p, C.c_uint64, vp, vp, vp, C.c_uint64, vp, vp]),
        "bsmi_chunks_read_into": (i32, [C.POINTER(Codec), i32, C.POINTER(ChunkCopy), i64p, i32, vp, C.POINTER(i32), i32]),
        "bsmi_chunks_write_from": (i32, [C.POINTER(Codec), i32, C.POINTER(ChunkCopy), i64p, i32, vp, C.POINTER(i32), i32]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    return lib, sorted(sigs)


lib, SYMBOLS = _load()


def check(rc):
    if rc != 0:
        raise BsmiError(rc, lib.bsmi_last_error().decode(errors="replace"))


def i64x3(v):
    return (C.c_int64 * 3)(*[int(x) for x in v])
