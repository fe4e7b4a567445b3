"""Host-side mirror of the reference model interface on top of libbsmi.

Reference being mirrored (paths relative to /root/reference/bootstrapper):
  models/3d_affs/model.py:28-64    Model()  -> forward(input) -> affs
  models/3d_mtlsd/model.py:28-68   Model()  -> forward(input) -> (lsds, affs)
  models/3d_affs/predict.py:98-108 checkpoint loading (state_dict / model_state_dict /
                                   raw dict, "model." prefix stripped), eval()

PyTorch is used for device memory and streams only; the arithmetic is in libbsmi.so.
"""
import ctypes as C
import json
import os

import numpy as np
import torch

from . import _lib
from ._lib import lib, check

# net_config["outputs"] key -> state-dict prefix of the head the reference Model builds
# (3-D: models/3d_mtlsd/model.py:54-59; 2-D: models/2d_mtlsd/model.py:56-61)
HEAD_OF_OUTPUT = {"3d_affs": "affs_head", "3d_lsds": "lsds_head", "2d_affs": "aff_head", "2d_lsds": "lsd_head"}

PRECISIONS = {"f32": _lib.PREC_F32, "fp32": _lib.PREC_F32, "float32": _lib.PREC_F32,
              "bf16": _lib.PREC_BF16, "bfloat16": _lib.PREC_BF16,
              # split bf16 (hi + lo operands, three MFMAs per product): the parity-grade fast mode
              "bf16x3": _lib.PREC_BF16X3}


def _tuplify(x):
    return [[int(v) for v in ks] for ks in x]


def _lift(k):
    """2-D kernel / factor (h, w) -> the 3-D one with unit depth."""
    k = [int(v) for v in k]
    if len(k) == 2:
        return [1] + k
    if len(k) != 3:
        raise ValueError("only 2-D and 3-D networks are supported by this engine")
    return k


def input_channels(net_config):
    """models/2d_mtlsd/model.py:46 (in_channels * adj_slices), models/3d_affs_from_2d_mtlsd/model.py:15 (sum of the
    input dims), models/3d_affs/model.py:13 (in_channels)."""
    if "in_channels" in net_config:
        return int(net_config["in_channels"]) * int(net_config.get("adj_slices", 1))
    return sum(int(v["dims"]) for v in net_config["inputs"].values())


def make_config(net_config):
    """net_config.json dict -> _lib.UNetConfig (same keys Model() reads, model.py:10-25).
    The 2-D setups (models/2d_*: Conv2d, MaxPool2d, bilinear upsampling over (h, w)) are the same network
    with unit-depth kernels and factors over a (1, h, w) volume; their state-dict tensors get the depth axis
    in Model.load_state_dict."""
    cfg = _lib.UNetConfig()
    cfg.in_channels = input_channels(net_config)
    cfg.num_fmaps = int(net_config["num_fmaps"])
    cfg.num_fmaps_out = int(net_config.get("num_fmaps_out") or 0)
    cfg.fmap_inc_factor = int(net_config["fmap_inc_factor"])
    dfs = net_config["downsample_factors"]
    nl = len(dfs) + 1
    if nl > _lib.MAX_LEVELS:
        raise ValueError(f"at most {_lib.MAX_LEVELS} levels supported")
    cfg.num_levels = nl
    for i, f in enumerate(dfs):
        f = _lift(f)
        for d in range(3):
            cfg.downsample_factors[i][d] = f[d]
    nd = len(dfs[0]) if dfs else 3
    k3 = [3] * nd
    ksd = net_config.get("kernel_size_down") or [[k3, k3]] * nl
    ksu = net_config.get("kernel_size_up") or [[k3, k3]] * (nl - 1)
    for dst_n, dst_k, src in ((cfg.n_convs_down, cfg.kernel_size_down, ksd),
                              (cfg.n_convs_up, cfg.kernel_size_up, ksu)):
        for i, ks in enumerate(src):
            if len(ks) > _lib.MAX_CONVS:
                raise ValueError(f"at most {_lib.MAX_CONVS} convolutions per pass supported")
            dst_n[i] = len(ks)
            for j, k in enumerate(ks):
                k = _lift(k)
                for d in range(3):
                    dst_k[i][j][d] = k[d]
    heads = []
    for name, val in net_config["outputs"].items():
        if name not in HEAD_OF_OUTPUT:
            raise ValueError(f"output {name!r} has no head in this engine")
        heads.append((HEAD_OF_OUTPUT[name], int(val["dims"])))
    if len(heads) > _lib.MAX_HEADS:
        raise ValueError("too many heads")
    cfg.num_heads = len(heads)
    for i, (hn, dims) in enumerate(heads):
        cfg.head_name[i].value = hn.encode()
        cfg.head_dims[i] = dims
    return cfg, heads


class Model:
    """Drop-in for the reference `Model` on one MI355X.

    forward(input) takes a float32 CUDA tensor (1, Cin, D, H, W) that is already
    normalised (as gp.torch.Predict hands it over) and returns the sigmoid head outputs,
    a single tensor for one head or a tuple in the reference's return order.
    predict_u8(raw_u8) additionally fuses the worker's u8 normalisation and the
    x255 -> uint8 store (predict.py:147-154).
    """

    def __init__(self, net_config, device=0, precision="bf16x3"):
        if isinstance(net_config, (str, os.PathLike)):
            with open(net_config) as f:
                net_config = json.load(f)
        self.net_config = net_config
        self.device = int(device)
        self.precision = PRECISIONS[precision]
        self._cfg, self.heads = make_config(net_config)
        dfs = net_config["downsample_factors"]
        self.two_d = bool(dfs) and len(dfs[0]) == 2
        # networks fed with raw data normalise to [-1, 1], the second-stage ones (inputs = earlier predictions) to [0, 1]
        self.raw_mode = _lib.RAW_U8 if "in_channels" in net_config else _lib.RAW_U8_UNIT
        self.stack_infer = False  # models/2d_mtlsd/model.py:31,71-73: add the z axis to the 2-D outputs
        self.param_shapes = {}  # state_dict key -> shape, as loaded
        self._h = C.c_void_p()
        check(lib.bsmi_unet_create(C.byref(self._cfg), self.device, C.byref(self._h)))
        self._finalized = set()

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            lib.bsmi_unet_destroy(h)
            self._h = None

    # -- weights -----------------------------------------------------------------------
    def load_state_dict(self, state_dict):
        """Strict, like torch's: unknown or missing keys raise (at finalize)."""
        self._state_dict = state_dict  # kept (a reference) for clone()
        for k, v in state_dict.items():
            if hasattr(v, "detach"):
                v = v.detach().cpu().numpy()
            a = np.ascontiguousarray(v, dtype=np.float32)
            self.param_shapes[k] = tuple(a.shape)
            if self.two_d and a.ndim == 4:
                a = a[:, :, None]  # Conv2d weight (O, I, kh, kw) -> (O, I, 1, kh, kw)
            shape = (C.c_int64 * a.ndim)(*a.shape)
            check(lib.bsmi_unet_load_weight(self._h, k.encode(), a.ctypes.data_as(C.c_void_p), shape, a.ndim))
        self._finalized.clear()
        self._finalize(self.precision)
        return self

    def clone(self):
        """A second engine with the same weights on the same device: its own activation buffers, so that its forward passes can
        overlap this one's (predict lanes, volume.VolumePipeline / predict.py)."""
        sd = getattr(self, "_state_dict", None)
        if sd is None:
            raise RuntimeError("clone(): no weights loaded")
        other = Model(self.net_config, device=self.device, precision={v: k for k, v in PRECISIONS.items()}[self.precision])
        other.stack_infer = self.stack_infer
        return other.load_state_dict(sd)

    def load_checkpoint(self, checkpoint):
        """predict.py:98-107: accept `ckpt` or `ckpt.ckpt`; take state_dict / model_state_dict /
        the dict itself; strip the Lightning 'model.' prefix."""
        path = checkpoint if os.path.exists(checkpoint) else f"{checkpoint}.ckpt"
        if not os.path.exists(path):
            raise FileNotFoundError(f"Neither {checkpoint} nor {checkpoint}.ckpt were found.")
        try:  # tensors only: a checkpoint is data, not code
            sd = torch.load(path, map_location="cpu", weights_only=True)
        except Exception:  # noqa: BLE001 - Lightning checkpoints may carry pickled hyper-parameter objects
            sd = torch.load(path, map_location="cpu", weights_only=False)
        sd = sd.get("state_dict", sd.get("model_state_dict", sd))
        sd = {k.removeprefix("model."): v for k, v in sd.items()}
        return self.load_state_dict(sd)

    def _finalize(self, prec):
        if prec not in self._finalized:
            check(lib.bsmi_unet_finalize(self._h, prec))
            self._finalized.add(prec)

    def set_precision(self, precision):
        self.precision = PRECISIONS[precision]
        self._finalize(self.precision)
        return self

    def eval(self):
        return self

    def set_persistent_grid(self, n_cus):
        """CUs available to the stream `forward` runs on (-1: the whole device, 0: no persistent launches)."""
        check(lib.bsmi_unet_set_persistent_grid(self._h, int(n_cus)))
        return self

    # -- shapes ------------------------------------------------------------------------
    def output_shape(self, in_shape):
        out = (C.c_int64 * 3)()
        check(lib.bsmi_unet_output_shape(self._h, _lib.i64x3(in_shape), out))
        return tuple(out)

    def flops(self, in_shape):
        f = C.c_double()
        check(lib.bsmi_unet_flops(self._h, _lib.i64x3(in_shape), C.byref(f)))
        return f.value

    # -- profiling ---------------------------------------------------------------------
    def profile(self, on=True):
        """Time every launch of every `on`-th forward with HIP events (True = every forward)."""
        check(lib.bsmi_unet_profile_enable(self._h, int(on)))
        return self

    def read_profile(self):
        """[(type, ms, flops)] per launch of the last forward (types: 0 input, 1 conv,
        2 pool, 3 upsample, 4 head).  Synchronises on the recorded events."""
        n = C.c_int()
        cap = 256
        types = (C.c_int32 * cap)()
        ms = (C.c_double * cap)()
        fl = (C.c_double * cap)()
        check(lib.bsmi_unet_profile_read(self._h, cap, C.byref(n), types, ms, fl))
        return [(types[i], ms[i], fl[i]) for i in range(min(n.value, cap))]

    def profile_totals(self, reset=False):
        """{type_name: (ms, flops, launches)} summed over profiled forwards since the last reset."""
        ms = (C.c_double * 5)()
        fl = (C.c_double * 5)()
        cnt = (C.c_int64 * 5)()
        check(lib.bsmi_unet_profile_totals(self._h, ms, fl, cnt, 1 if reset else 0))
        names = ["input", "conv", "pool", "upsample", "head"]
        return {names[i]: (ms[i], fl[i], cnt[i]) for i in range(5)}

    def profile_executed(self, reset=False):
        """FLOPs the matrix pipe was given by the profiled conv launches since the last reset (padding, Winograd batches and the
        three products of the split mode counted): bsmi_unet_profile_executed."""
        v = C.c_double()
        check(lib.bsmi_unet_profile_executed(self._h, C.byref(v), 1 if reset else 0))
        return v.value

    def debug_activation(self, step, what=0):
        """Development aid: output tensor of launch `step` of the last forward as a float32 (D, H, W, C) array
        (what: 0 value, 1 / 2 the hi / lo plane of the split mode)."""
        shape = (C.c_int64 * 4)()
        check(lib.bsmi_unet_debug_activation(self._h, int(step), int(what), shape, None, 0))
        out = np.empty(tuple(shape), dtype=np.float32)
        check(lib.bsmi_unet_debug_activation(self._h, int(step), int(what), shape, out.ctypes.data_as(C.c_void_p), out.size))
        return out

    # -- forward -----------------------------------------------------------------------
    def _run(self, raw, raw_dtype, in_shape, want_f32, want_u8):
        dev = torch.device("cuda", self.device)
        osz = self.output_shape(in_shape)
        nh = len(self.heads)
        f32 = [torch.empty((d,) + osz, dtype=torch.float32, device=dev) if want_f32 else None
               for _, d in self.heads]
        u8 = [torch.empty((d,) + osz, dtype=torch.uint8, device=dev) if want_u8 else None
              for _, d in self.heads]
        pf = (C.c_void_p * nh)(*[t.data_ptr() if t is not None else None for t in f32])
        pu = (C.c_void_p * nh)(*[t.data_ptr() if t is not None else None for t in u8])
        stream = torch.cuda.current_stream(dev).cuda_stream
        check(lib.bsmi_unet_forward(self._h, self.precision, C.c_void_p(raw.data_ptr()), raw_dtype,
                                    _lib.i64x3(in_shape), pf, pu, C.c_void_p(stream)))
        return f32, u8

    def forward(self, *inputs):
        """3-D setups: (1, C, D, H, W).  Second-stage setups take their inputs one by one, in the order of
        net_config["inputs"] (models/3d_affs_from_2d_mtlsd/model.py:62-64), and concatenate the channels.
        2-D setups: (1, C, H, W), or (1, c, d, H, W) which is viewed as (1, c d, H, W) (models/2d_mtlsd/model.py:63-68)."""
        if not all(t.is_cuda for t in inputs):
            raise RuntimeError("bootstrapper_amd.Model runs on the GPU only; move the input to cuda")
        input = inputs[0] if len(inputs) == 1 else torch.cat(inputs, dim=1)
        if self.two_d:
            if input.dim() == 5:
                n, c, d, hh, ww = input.shape
                input = input.reshape(n, c * d, hh, ww)
            if input.dim() != 4 or input.shape[0] != 1:
                raise ValueError("expected input of shape (1, C, H, W)")
            input = input[:, :, None]
        if input.dim() != 5 or input.shape[0] != 1:
            raise ValueError("expected input of shape (1, C, D, H, W)")
        if input.shape[1] != self._cfg.in_channels:
            raise ValueError(f"expected {self._cfg.in_channels} input channels, got {input.shape[1]}")
        x = input.to(torch.float32).contiguous()
        f32, _ = self._run(x, _lib.RAW_F32, x.shape[2:], True, False)
        outs = [t[None] for t in f32]
        if self.two_d and not self.stack_infer:
            outs = [t[:, :, 0] for t in outs]
        return outs[0] if len(outs) == 1 else tuple(outs)

    __call__ = forward

    def predict_u8(self, raw_u8, want_f32=False):
        """raw_u8: uint8 CUDA tensor (D,H,W) or (Cin,D,H,W) -> list of uint8 (dims,d,h,w)
        tensors in head order (and the float32 sigmoid outputs if want_f32).  Normalisation as in the
        setup's predict.py: u8/255*2-1 for raw data, u8/255 for the predictions a second-stage net reads.
        A 2-D setup treats D as a stack of independent sections (its kernels have unit depth)."""
        if raw_u8.dtype != torch.uint8 or not raw_u8.is_cuda:
            raise ValueError("raw_u8 must be a uint8 CUDA tensor")
        x = raw_u8.contiguous()
        shape = x.shape[-3:]
        cin = 1 if x.dim() == 3 else x.shape[0]
        if cin != self._cfg.in_channels:
            raise ValueError(f"expected {self._cfg.in_channels} input channels, got {cin}")
        f32, u8 = self._run(x, self.raw_mode, shape, want_f32, True)
        return (u8, f32) if want_f32 else u8


def extract_block_reflect(vol_u8, offset, block_shape):
    """gp.Pad(raw, None, mode='reflect') + ROI read (predict.py:145-148) on the device."""
    out = torch.empty(tuple(int(s) for s in block_shape), dtype=torch.uint8, device=vol_u8.device)
    stream = torch.cuda.current_stream(vol_u8.device).cuda_stream
    check(lib.bsmi_extract_block_reflect_u8(C.c_void_p(vol_u8.data_ptr()), _lib.i64x3(vol_u8.shape),
                                            _lib.i64x3(offset), _lib.i64x3(block_shape),
                                            C.c_void_p(out.data_ptr()), C.c_void_p(stream)))
    return out
