"""Training step on the device: the counterpart of the reference's Lightning module for one process per GPU.

Reference being mirrored (paths relative to /root/reference/bootstrapper):
  models/3d_affs/train.py:138-159   LitModel: training_step = loss_fn(model(raw), gt_affs, affs_weights),
                                    configure_optimizers = Adam(lr=0.5e-4)
  models/3d_mtlsd/train.py          the same with (lsds, affs) heads; the loss is the sum of both
  training.py:96-137                fit(): implicit DDP -- one process per GPU, gradients averaged over the ranks

The arithmetic (forward, loss, backward, Adam) is in libbsmi (csrc/train.hip); this module only wires buffers and,
when torch.distributed is initialised, all-reduces the flat gradient buffer over RCCL (`nccl` backend) or gloo.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import lib, check


def sum_gradients(flat):
    """Sum the flat gradient buffer over the ranks of the default process group, in place, and return the factor that
    turns the sum into DDP's mean (1 / world_size; 1.0 without a process group).  `nccl` (= RCCL over xGMI) reduces
    the device buffer directly; `gloo` (tests, CPU rehearsal) goes through a host copy."""
    dist = torch.distributed
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 1.0
    if flat.is_cuda and dist.get_backend() != "nccl":
        torch.cuda.current_stream(flat.device).synchronize()
        host = flat.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        flat.copy_(host)
    else:
        if flat.is_cuda:
            torch.cuda.current_stream(flat.device).synchronize()  # the backward kernels run on this stream, RCCL on its own
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return 1.0 / dist.get_world_size()


def save_checkpoint(trainer, path, iteration):
    """`model_checkpoint_<iteration>` in the layout the reference's predict worker loads
    (models/3d_affs/predict.py:98-108: a dict with "state_dict" whose keys carry the Lightning "model." prefix)."""
    sd = {"model." + k: torch.from_numpy(trainer.read(k).reshape(shape)) for k, shape in trainer.param_shapes().items()}
    torch.save({"state_dict": sd, "global_step": int(iteration)}, path)


def fit(trainer, batches, max_iterations, save_checkpoints_every=0, setup_dir=None, log_every=10, log=print, start_iteration=0):
    """The training loop of training.py:96-137 without Lightning: `batches` is any iterable of reference-style batch
    dicts ("raw", "gt_affs", "affs_weights"[, "gt_lsds", "lsds_weights"]) of CUDA float32 tensors."""
    import os
    it = int(start_iteration)
    for batch in batches:
        if it >= max_iterations:
            break
        loss = trainer.training_step(batch)
        it += 1
        if log and (it % log_every == 0 or it == start_iteration + 1):
            log(f"step {it}: train_loss {loss:.6f}")
        if save_checkpoints_every and setup_dir and it % save_checkpoints_every == 0:
            rank = torch.distributed.get_rank() if torch.distributed.is_available() and torch.distributed.is_initialized() else 0
            if rank == 0:
                save_checkpoint(trainer, os.path.join(setup_dir, f"model_checkpoint_{it}.ckpt"), it)
    return it


class _DevBuf:
    """a raw device buffer as seen through __cuda_array_interface__ (zero-copy torch view)"""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 2}


class Trainer:
    def __init__(self, model, in_shape, lr=0.5e-4, betas=(0.9, 0.999), eps=1e-8):
        """model: bootstrapper_amd.unet.Model with weights loaded; in_shape: (D, H, W) of the training block."""
        self.model = model
        self.in_shape = tuple(int(s) for s in in_shape)
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        model._finalize(_lib.PREC_F32)
        check(lib.bsmi_unet_train_begin(model._h, _lib.i64x3(self.in_shape)))
        n = C.c_uint64()
        check(lib.bsmi_unet_train_num_params(model._h, C.byref(n)))
        pw, pg = C.c_void_p(), C.c_void_p()
        check(lib.bsmi_unet_train_buffers(model._h, C.byref(pw), C.byref(pg)))
        dev = torch.device("cuda", model.device)
        self.params = torch.as_tensor(_DevBuf(pw.value, n.value), device=dev)
        self.grads = torch.as_tensor(_DevBuf(pg.value, n.value), device=dev)
        self.out_shape = model.output_shape(self.in_shape)
        self.last_loss = None
        self._shapes = dict(model.param_shapes)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(torch.device("cuda", self.model.device)).cuda_stream)

    def forward_backward(self, raw, targets, weights):
        """raw: float32 CUDA (Cin, D, H, W) or (D, H, W); targets / weights: per head float32 CUDA (C, d, h, w).
        Returns the loss (python float; synchronises).  Gradients are in self.grads."""
        nh = len(self.model.heads)
        if len(targets) != nh or len(weights) != nh:
            raise ValueError(f"the model has {nh} head(s)")
        raw = raw.contiguous()
        if raw.dtype != torch.float32 or not raw.is_cuda:
            raise ValueError("raw must be a float32 CUDA tensor")
        ts = [t.contiguous() for t in targets]
        ws = [w.contiguous() for w in weights]
        for (name, dims), t, w in zip(self.model.heads, ts, ws):
            want = (dims,) + tuple(self.out_shape)
            if tuple(t.shape) != want or tuple(w.shape) != want or t.dtype != torch.float32 or w.dtype != torch.float32:
                raise ValueError(f"{name}: target and weights must be float32 of shape {want}")
        tp = (C.c_void_p * nh)(*[t.data_ptr() for t in ts])
        wp = (C.c_void_p * nh)(*[w.data_ptr() for w in ws])
        loss = C.c_float()
        check(lib.bsmi_unet_train_forward_backward(self.model._h, C.c_void_p(raw.data_ptr()), tp, wp, C.byref(loss), self._stream()))
        self.last_loss = float(loss.value)
        return self.last_loss

    def optimizer_step(self):
        """Average the gradients over the ranks (if torch.distributed is up) and apply Adam."""
        scale = sum_gradients(self.grads)
        check(lib.bsmi_unet_train_adam_step(self.model._h, self.lr, self.betas[0], self.betas[1], self.eps, scale, self._stream()))

    def training_step(self, batch):
        """batch: dict like the reference's ("raw", then per head "gt_<x>", "<x>_weights" in head order)."""
        heads = [h for h, _ in self.model.heads]
        key = {"affs_head": ("gt_affs", "affs_weights"), "lsds_head": ("gt_lsds", "lsds_weights")}
        loss = self.forward_backward(batch["raw"], [batch[key[h][0]] for h in heads], [batch[key[h][1]] for h in heads])
        self.optimizer_step()
        return loss

    def read(self, key, what="param"):
        idx = {"param": 0, "grad": 1, "exp_avg": 2, "exp_avg_sq": 3}[what]
        off, cnt = C.c_uint64(), C.c_uint64()
        check(lib.bsmi_unet_train_param_info(self.model._h, key.encode(), C.byref(off), C.byref(cnt)))
        out = np.empty(cnt.value, dtype=np.float32)
        check(lib.bsmi_unet_train_read_param(self.model._h, key.encode(), idx, out.ctypes.data_as(C.c_void_p)))
        return out

    def param_shapes(self):
        return dict(self._shapes)

    def close(self):
        if self.model is not None:
            check(lib.bsmi_unet_train_end(self.model._h))
            self.model = None
