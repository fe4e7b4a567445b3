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


REDUCE_CHUNK = 8 << 20  # floats per all-reduce call (32 MB): the 243 MB weight of l_conv.3 goes out in eight pieces


def reduce_gradient_groups(trainer):
    """Data-parallel gradient reduction overlapped with the backward pass: the gradients of a ConvPass / head are summed
    over the ranks as soon as the backward pass has left that pass (bsmi_unet_train_wait_grad_group), group after group
    on a side stream, in pieces of REDUCE_CHUNK floats, while the earlier layers are still being differentiated.
    Returns the factor that turns the sums into DDP's mean; the caller's stream then waits for every piece.
    `nccl` (= RCCL over xGMI) reduces device memory in place; `gloo` (tests) stages each piece through the host."""
    dist = torch.distributed
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 1.0
    dev = trainer.grads.device
    comm = trainer.comm_stream
    nccl = dist.get_backend() == "nccl"
    works = []
    with torch.cuda.stream(comm):
        for g, (off, cnt) in enumerate(trainer.grad_groups):
            check(lib.bsmi_unet_train_wait_grad_group(trainer.model._h, g, C.c_void_p(comm.cuda_stream)))
            for a in range(off, off + cnt, REDUCE_CHUNK):
                piece = trainer.grads[a:min(a + REDUCE_CHUNK, off + cnt)]
                if nccl:
                    works.append(dist.all_reduce(piece, op=dist.ReduceOp.SUM, async_op=True))
                else:
                    comm.synchronize()
                    host = piece.cpu()
                    dist.all_reduce(host, op=dist.ReduceOp.SUM)
                    piece.copy_(host)
        done = torch.cuda.Event()
        done.record(comm)
    for w in works:
        w.wait()  # the current (backward) stream waits for the reduction
    torch.cuda.current_stream(dev).wait_event(done)
    return 1.0 / dist.get_world_size()


def save_checkpoint(trainer, path, iteration):
    """`model_checkpoint_<iteration>` in the layout the reference's predict worker loads
    (models/3d_affs/predict.py:98-108: a dict with "state_dict" whose keys carry the Lightning "model." prefix), plus
    the optimizer state in torch.optim.Adam's state_dict layout as Lightning stores it ("optimizer_states": parameters
    numbered in state_dict order), so that a resumed run continues with its moments and step count."""
    shapes = trainer.param_shapes()
    sd = {"model." + k: torch.from_numpy(trainer.read(k).reshape(shape)) for k, shape in shapes.items()}
    step = trainer.step_count()
    state = {i: {"step": torch.tensor(float(step)), "exp_avg": torch.from_numpy(trainer.read(k, "exp_avg").reshape(shape)),
                 "exp_avg_sq": torch.from_numpy(trainer.read(k, "exp_avg_sq").reshape(shape))}
             for i, (k, shape) in enumerate(shapes.items())}
    group = {"lr": trainer.lr, "betas": trainer.betas, "eps": trainer.eps, "weight_decay": 0, "amsgrad": False,
             "params": list(range(len(shapes)))}
    torch.save({"state_dict": sd, "global_step": int(iteration), "optimizer_states": [{"state": state, "param_groups": [group]}]}, path)


def load_optimizer_state(trainer, checkpoint):
    """Restore Adam's moments and step count from a checkpoint written by save_checkpoint (or by Lightning for the
    reference model: same parameter order).  -> True if the checkpoint carried an optimizer state."""
    try:
        ck = torch.load(checkpoint, map_location="cpu", weights_only=True)
    except Exception:  # noqa: BLE001 - an older torch or a checkpoint with non-tensor payloads
        ck = torch.load(checkpoint, map_location="cpu", weights_only=False)
    states = ck.get("optimizer_states") if isinstance(ck, dict) else None
    if not states:
        return False
    state = states[0]["state"]
    keys = list(trainer.param_shapes())
    step = 0
    for i, k in enumerate(keys):
        st = state.get(i)
        if st is None:
            raise ValueError(f"optimizer state of the checkpoint has no entry for parameter {i} ({k})")
        trainer.write(k, "exp_avg", st["exp_avg"].numpy())
        trainer.write(k, "exp_avg_sq", st["exp_avg_sq"].numpy())
        step = int(float(st["step"]))
    trainer.step_count(step)
    return True


def save_snapshot(setup_dir, voxel_size, step, rank, data):
    """SnapshotCallback._save_snapshot (training.py:69-93): `<setup_dir>/snapshots/batch_<step>_rank_<rank>.zarr` with one
    dataset per entry of the batch and of the predictions; floats in [-1, 1] go back to uint8 [0, 255], everything else
    as it is; output-sized arrays are centred inside the input-sized ones (`offset`, in world units)."""
    import os
    from .zarr_io import prepare_ds
    path = os.path.join(setup_dir, "snapshots", f"batch_{step}_rank_{rank}.zarr")
    n = len(voxel_size)
    arrays = {k: (v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)) for k, v in data.items()}
    largest = [max(a.shape[-n:][d] for a in arrays.values()) for d in range(n)]
    for key, a in arrays.items():
        if not np.issubdtype(a.dtype, np.integer):
            lo, hi = a.min(), a.max()
            if -1 <= lo < 0 and hi <= 1:
                a = ((a * 0.5 + 0.5) * 255).astype(np.uint8)
        offset = [(big - s) // 2 * int(v) for big, s, v in zip(largest, a.shape[-n:], voxel_size)]
        ds = prepare_ds(f"{path}/{key}", a.shape, offset=offset, voxel_size=[int(v) for v in voxel_size], dtype=a.dtype)
        ds[(slice(None),) * a.ndim] = a
    return path


def fit(trainer, batches, max_iterations, save_checkpoints_every=0, setup_dir=None, log_every=10, log=print, start_iteration=0,
        save_snapshots_every=0, voxel_size=None):
    """The training loop of training.py:96-137 without Lightning: `batches` is any iterable of reference-style batch
    dicts ("raw", "gt_affs", "affs_weights"[, "gt_lsds", "lsds_weights"]) of CUDA float32 tensors.  Every `log_every`
    steps the loss goes to a TensorBoard event file under `<setup_dir>/log/version_<n>/` -- the layout of the reference's
    TensorBoardLogger(setup_dir, name="log"), training.py:130, written by tb_events.ScalarWriter -- and to
    `<setup_dir>/log/train_loss.csv`; snapshots as SnapshotCallback writes them (step 1 and every
    `save_snapshots_every` steps)."""
    import os
    it = int(start_iteration)
    rank = torch.distributed.get_rank() if torch.distributed.is_available() and torch.distributed.is_initialized() else 0
    scalars = events = None
    if setup_dir and rank == 0:
        from .tb_events import ScalarWriter
        os.makedirs(os.path.join(setup_dir, "log"), exist_ok=True)
        scalars = open(os.path.join(setup_dir, "log", "train_loss.csv"), "a")
        if scalars.tell() == 0:
            scalars.write("step,train_loss\n")
        events = ScalarWriter(os.path.join(setup_dir, "log"))
    for batch in batches:
        if it >= max_iterations:
            break
        loss = trainer.training_step(batch)
        it += 1
        if log and (it % log_every == 0 or it == start_iteration + 1):
            log(f"step {it}: train_loss {loss:.6f}")
        if scalars and it % log_every == 0:
            scalars.write(f"{it},{loss:.8g}\n")
            scalars.flush()
            events.add_scalar("train_loss", loss, it)
        if save_snapshots_every and setup_dir and voxel_size is not None and (it == 1 or it % save_snapshots_every == 0):
            data = dict(batch)
            data.update(trainer.predictions())
            save_snapshot(setup_dir, voxel_size, it, rank, data)
        if save_checkpoints_every and setup_dir and it % save_checkpoints_every == 0:
            rank = torch.distributed.get_rank() if torch.distributed.is_available() and torch.distributed.is_initialized() else 0
            if rank == 0:
                save_checkpoint(trainer, os.path.join(setup_dir, f"model_checkpoint_{it}.ckpt"), it)
    if scalars:
        scalars.close()
        events.close()
    return it


class _DevBuf:
    """a raw device buffer as seen through __cuda_array_interface__ (zero-copy torch view)"""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 2}


_COMM_STREAMS = {}


class Trainer:
    def __init__(self, model, in_shape, lr=0.5e-4, betas=(0.9, 0.999), eps=1e-8, arithmetic="split-bf16", deterministic=False):
        """model: bootstrapper_amd.unet.Model with weights loaded; in_shape: (D, H, W) of the training block.
        arithmetic: "split-bf16" (default: the convolutions multiply f32 operands as bf16 hi + lo pairs on the bf16 matrix
        pipe, f32 accumulation; tensors, loss, gradients and Adam are fp32) or "f32" (exact f32 MFMA, about half the speed).
        deterministic: every reduction of the step in a fixed order instead of float atomics -- two runs give the same bits."""
        if arithmetic not in ("split-bf16", "f32"):
            raise ValueError(f"arithmetic must be 'split-bf16' or 'f32', not {arithmetic!r}")
        self.arithmetic = arithmetic
        self.model = model
        model._state_dict = None   # the handle's weights are about to change: Model.clone() must not hand out the loaded ones
        self.in_shape = tuple(int(s) for s in in_shape)
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        model._finalize(_lib.PREC_F32)
        check(lib.bsmi_unet_train_set_arithmetic(model._h, 1 if arithmetic == "split-bf16" else 0))
        check(lib.bsmi_unet_train_set_deterministic(model._h, 1 if deterministic else 0))
        self.deterministic = bool(deterministic)
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
        ng = C.c_int()
        check(lib.bsmi_unet_train_grad_groups(model._h, 0, C.byref(ng), None, None))
        offs, cnts = (C.c_uint64 * ng.value)(), (C.c_uint64 * ng.value)()
        check(lib.bsmi_unet_train_grad_groups(model._h, ng.value, C.byref(ng), offs, cnts))
        self.grad_groups = [(int(o), int(c)) for o, c in zip(offs, cnts)]   # in the order the backward pass finishes them
        self._comm_stream = None   # made when a reduction first needs it: a stream costs a hardware queue (few; volume.py)

    @property
    def comm_stream(self):
        """the side stream the gradient groups are all-reduced on (one per device and process)"""
        if self._comm_stream is None:
            dev = torch.device("cuda", self.model.device)
            if dev not in _COMM_STREAMS:
                _COMM_STREAMS[dev] = torch.cuda.Stream(dev)
            self._comm_stream = _COMM_STREAMS[dev]
        return self._comm_stream

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(torch.device("cuda", self.model.device)).cuda_stream)

    def forward_backward(self, raw, targets, weights, wait=True):
        """raw: float32 CUDA (Cin, D, H, W) or (D, H, W); targets / weights: per head float32 CUDA (C, d, h, w).
        Returns the loss (python float; synchronises) -- or, with wait=False, None: the step is only queued and the
        loss is read later (last_loss()).  Gradients are in self.grads."""
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
        self._keep = (raw, ts, ws)  # alive until the queued kernels have read them
        if not wait:
            check(lib.bsmi_unet_train_forward_backward(self.model._h, C.c_void_p(raw.data_ptr()), tp, wp, None, self._stream()))
            return None
        loss = C.c_float()
        check(lib.bsmi_unet_train_forward_backward(self.model._h, C.c_void_p(raw.data_ptr()), tp, wp, C.byref(loss), self._stream()))
        self.last_loss = float(loss.value)
        return self.last_loss

    def read_last_loss(self):
        loss = C.c_float()
        check(lib.bsmi_unet_train_last_loss(self.model._h, C.byref(loss), self._stream()))
        self.last_loss = float(loss.value)
        return self.last_loss

    def optimizer_step(self, overlapped=True):
        """Average the gradients over the ranks (if torch.distributed is up) and apply Adam.  overlapped: reduce group by
        group behind the backward pass (reduce_gradient_groups); else one all-reduce of the whole buffer once it is done."""
        scale = reduce_gradient_groups(self) if overlapped else sum_gradients(self.grads)
        check(lib.bsmi_unet_train_adam_step(self.model._h, self.lr, self.betas[0], self.betas[1], self.eps, scale, self._stream()))

    def training_step(self, batch):
        """batch: dict like the reference's ("raw", then per head "gt_<x>", "<x>_weights" in head order).  The forward /
        backward pass is queued without waiting, the gradient reduction follows it group by group, then Adam; the loss is
        read last (one synchronisation per step)."""
        heads = [h for h, _ in self.model.heads]
        key = {"affs_head": ("gt_affs", "affs_weights"), "lsds_head": ("gt_lsds", "lsds_weights")}
        self.forward_backward(batch["raw"], [batch[key[h][0]] for h in heads], [batch[key[h][1]] for h in heads], wait=False)
        self.optimizer_step()
        return self.read_last_loss()

    def predictions(self):
        """{"pred_<x>": float32 CUDA (dims, d, h, w)} of the last step, named like the reference's training_step outputs"""
        name = {"affs_head": "pred_affs", "lsds_head": "pred_lsds"}
        dev = torch.device("cuda", self.model.device)
        out = {}
        for i, (head, dims) in enumerate(self.model.heads):
            ptr, cnt = C.c_void_p(), C.c_uint64()
            check(lib.bsmi_unet_train_prediction(self.model._h, i, C.byref(ptr), C.byref(cnt)))
            out[name.get(head, "pred_" + head)] = torch.as_tensor(_DevBuf(ptr.value, cnt.value), device=dev).view((dims,) + tuple(self.out_shape)).clone()
        return out

    def read(self, key, what="param"):
        idx = {"param": 0, "grad": 1, "exp_avg": 2, "exp_avg_sq": 3}[what]
        off, cnt = C.c_uint64(), C.c_uint64()
        check(lib.bsmi_unet_train_param_info(self.model._h, key.encode(), C.byref(off), C.byref(cnt)))
        out = np.empty(cnt.value, dtype=np.float32)
        check(lib.bsmi_unet_train_read_param(self.model._h, key.encode(), idx, out.ctypes.data_as(C.c_void_p)))
        return out

    def write(self, key, what, values):
        """restore an Adam moment ("exp_avg" / "exp_avg_sq") of one parameter"""
        idx = {"exp_avg": 2, "exp_avg_sq": 3}[what]
        a = np.ascontiguousarray(values, dtype=np.float32).ravel()
        off, cnt = C.c_uint64(), C.c_uint64()
        check(lib.bsmi_unet_train_param_info(self.model._h, key.encode(), C.byref(off), C.byref(cnt)))
        if a.size != cnt.value:
            raise ValueError(f"{key}: {a.size} values for {cnt.value} parameters")
        check(lib.bsmi_unet_train_write_param(self.model._h, key.encode(), idx, a.ctypes.data_as(C.c_void_p)))

    def step_count(self, set_to=None):
        v = C.c_int()
        check(lib.bsmi_unet_train_step_count(self.model._h, -1 if set_to is None else int(set_to), C.byref(v)))
        return v.value

    def param_shapes(self):
        return dict(self._shapes)

    def close(self):
        if self.model is not None:
            check(lib.bsmi_unet_train_end(self.model._h))
            # the library dropped the bf16 / split-bf16 images of the old weights (re-packed from the trained ones on demand):
            # the mirror's record of what is finalized follows, so a predict in those modes on the same Model re-packs
            self.model._finalized.intersection_update({_lib.PREC_F32})
            self.model = None
