"""Thin evaluation / pattern-dump harness around the models.

Mirrors, for the forward (inference) path only:

    engine.py:118-151     evaluate_multiclass()      -> evaluate_multiclass()
    utils.py:20-75        SmoothedValue global_avg + synchronize_between_processes (sum of count/total over ranks)
    timm.utils.accuracy   top-k accuracy in percent   -> accuracy()
    validate.py:163-229   per-image `Stage-{loc}` records, relative -> absolute kept-token composition -> image_records()
    validate.py:26-30, 278-287   NumpyArrayEncoder / write_viz -> write_viz()

    engine.py:14-114      train_one_epoch()          -> train_one_epoch()  (grad accumulation, clipping, EMA, frozen groups, step-wise LR)
    timm ModelEmaV2       (train.py:399, engine.py:88) -> ModelEma

Host logic only: the model is any callable returning logits or `(logits, viz_data)`; in train mode `loss.backward()` runs the HIP
backward executor (training.py) and the optimizer is a stock torch optimizer over the models' fp32 master parameters.
"""
import contextlib
import copy
import math
import json
from typing import Dict, Iterable, List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist
import torch.nn.functional as F


def accuracy(output: torch.Tensor, target: torch.Tensor, topk: Sequence[int] = (1,)) -> List[torch.Tensor]:
    """timm.utils.accuracy: percentage of rows whose target is among the k largest logits."""
    maxk = min(max(topk), output.shape[1])
    _, pred = output.topk(maxk, 1, True, True)
    correct = pred.t().eq(target.reshape(1, -1).expand(maxk, -1))
    return [correct[:min(k, maxk)].reshape(-1).float().sum(0) * 100.0 / target.shape[0] for k in topk]


class _Meter:
    """utils.py:20-75 reduced to what evaluate_* reads: global_avg = total / count, summed over ranks on synchronize."""

    def __init__(self):
        self.count, self.total = 0, 0.0

    def update(self, value: float, n: int = 1):
        self.count += n
        self.total += value * n

    def synchronize_between_processes(self, device=None):
        if not (dist.is_available() and dist.is_initialized()):       # utils.py:40-41
            return
        dev = device if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([self.count, self.total], dtype=torch.float64, device=dev)
        dist.barrier()
        dist.all_reduce(t)
        self.count, self.total = int(t[0].item()), float(t[1].item())

    @property
    def global_avg(self) -> float:
        return self.total / self.count


def _check_status(model):
    """What only the device knows about the forwards just run (models.VisionTransformer.check_status: the fused Mlp's stream-K hand-over
    record) becomes an exception here, where the results are consumed; DDP-style wrappers are looked through."""
    m = getattr(model, "module", model)
    if hasattr(m, "check_status"):
        m.check_status()


@torch.no_grad()
def evaluate_multiclass(data_loader: Iterable, model, device) -> Dict[str, float]:
    """engine.py:118-151.  `loss` averages the per-batch mean losses (meter weight 1 per batch, as the reference's
    `metric_logger.update(loss=loss.item())` does); acc1/acc5 are weighted by batch size."""
    meters = {"loss": _Meter(), "acc1": _Meter(), "acc5": _Meter()}
    if hasattr(model, "eval"):
        model.eval()
    # One batch of lookahead where the model offers it (VisionTransformer.forward_async): batch k + 1 is launched -- on a side stream with its
    # own workspace -- before batch k's logits are consumed (the .item() calls below synchronise), so that two forwards are in flight and
    # the device fills the tail of one's launches with the other's.  Same results, in the loader's order.
    launch = getattr(getattr(model, "module", model), "forward_async", None)

    def consume(pending, target, n):
        output = pending.result() if launch is not None else pending
        if isinstance(output, (tuple, list)):
            output = output[0]
        loss = F.cross_entropy(output.float(), target)
        acc1, acc5 = accuracy(output, target, topk=(1, 5))
        meters["loss"].update(loss.item())
        meters["acc1"].update(acc1.item(), n=n)
        meters["acc5"].update(acc5.item(), n=n)

    prev = None
    for images, target in data_loader:
        images = images.to(device, non_blocking=True)
        target = target.to(device, non_blocking=True)
        cur = (launch(images) if launch is not None else model(images), target, images.shape[0])
        if prev is not None:
            consume(*prev)
        prev = cur
    if prev is not None:
        consume(*prev)
    _check_status(model)
    for m in meters.values():
        m.synchronize_between_processes(device)
    return {k: m.global_avg for k, m in meters.items()}


class ModelEma:
    """timm 0.4.12 ModelEmaV2 (train.py:399-402, engine.py:88-89): a deep copy of the model in eval mode whose parameters and
    buffers follow ema = decay * ema + (1 - decay) * model after every optimizer step."""

    def __init__(self, model, decay: float = 0.9999, device=None):
        self.module = copy.deepcopy(model)          # VisionTransformer.__deepcopy__ leaves every executor cache behind (workspaces, graphs, tape)
        self.module.eval()
        self.decay = decay
        self.device = device
        if device is not None:
            self.module.to(device=device)

    @torch.no_grad()
    def update(self, model):
        ema_v, model_v = list(self.module.state_dict().values()), list(model.state_dict().values())
        fl_e = [e for e, m in zip(ema_v, model_v) if e.is_floating_point()]
        fl_m = [m.to(e.device) for e, m in zip(ema_v, model_v) if e.is_floating_point()]
        torch._foreach_lerp_(fl_e, fl_m, 1.0 - self.decay)                       # e + (1 - decay) * (m - e)
        for e, m in zip(ema_v, model_v):
            if not e.is_floating_point():
                e.copy_(m)
        if hasattr(self.module, "weights_changed"):                               # (any nn.Module can be averaged; the HIP models repack)
            self.module.weights_changed()                                         # the packed bf16 copies follow the new values


def train_one_epoch(model, criterion, data_loader: Iterable, optimizer, device, epoch: int, lr_scheduler=None, max_norm: float = 0,
                    model_ema=None, mixup_fn=None, grad_accum_steps: int = 1, num_steps_epoch: int = 1000, reducer=None,
                    reduce_every_micro_step: bool = False):
    """engine.py:14-114 without the logging: frozen parameter groups (`fix_step`, :35-37), gradient accumulation with the
    optimizer stepping every `grad_accum_steps` batches or at the end of the loader (:41, :76-91), loss / grad_accum_steps
    (:62-63), finite-loss check (:65-67), gradient clipping by total norm (:71-75 `dispatch_clip_grad(mode="norm")`), EMA update
    and step-wise LR schedule after each optimizer step (:88-89, :108-111).  `criterion(samples, output, targets, model)` as the
    reference's loss wrappers are called (:60).  `reducer` (dp.FlatGradReducer): the reference's DDP all-reduces on every
    micro-step (no `no_sync`); here the accumulation micro-steps skip the collective unless `reduce_every_micro_step`.
    Returns ({"loss": mean over the epoch and over ranks, "lr-i": ...}, total_step) like engine.py:113-114."""
    model.train(True)
    meter = _Meter()
    total_step = epoch * num_steps_epoch
    n_batches = len(data_loader) if hasattr(data_loader, "__len__") else None
    epoch_step = 0
    for samples, targets in data_loader:
        for g in optimizer.param_groups:                       # engine.py:35-37
            if epoch < g.get("fix_step", 0):
                g["lr"] = 0
        epoch_step += 1
        opt_step = (epoch_step % grad_accum_steps == 0) or (n_batches is not None and epoch_step == n_batches)
        samples = samples.to(device, non_blocking=True)
        targets = targets.to(device, non_blocking=True)
        if mixup_fn is not None:
            samples, targets = mixup_fn(samples, targets)
        skip_sync = reducer is not None and not opt_step and not reduce_every_micro_step
        with (reducer.no_sync() if skip_sync else contextlib.nullcontext()):
            output = model(samples)
            loss = criterion(samples, output, targets, model)
            loss_value = loss.item() / grad_accum_steps
            loss = loss / grad_accum_steps
            if not math.isfinite(loss_value):
                raise FloatingPointError(f"Loss is {loss_value}, stopping training")      # engine.py:65-67
            loss.backward()
        if max_norm is not None and max_norm > 0.0:
            torch.nn.utils.clip_grad_norm_(list(model.parameters()), max_norm)
        if opt_step:
            optimizer.step()
            optimizer.zero_grad()
            if model_ema is not None:
                model_ema.update(model)
            total_step += 1
        if samples.is_cuda:
            torch.cuda.synchronize()                           # engine.py:93
        meter.update(loss_value)
        if lr_scheduler is not None and opt_step:
            lr_scheduler.step_update(num_updates=total_step)
    meter.synchronize_between_processes(device)
    stats = {"loss": meter.global_avg}
    for i, g in enumerate(optimizer.param_groups):
        stats[f"lr-{i}"] = g["lr"]
    return stats, total_step


def plain_criterion(fn):
    """Adapter: a `fn(output, targets)` loss called the way engine.py:60 calls its loss wrappers."""
    def wrapped(samples, output, targets, model):
        if isinstance(output, (tuple, list)):
            output = output[0]
        return fn(output.float(), targets)
    return wrapped


def _np(a):
    return a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)


def image_records(model_name: str, reduction_count: Sequence[int], viz_data: dict, i: int) -> Dict[str, dict]:
    """The `Stage-{loc}` part of one image's record (validate.py:199-229).

    `Kept_Tokens` of the first stage are patch indices of the input grid; later stages index into the PREVIOUS stage's kept
    list, so they are composed into absolute indices here.  Padding entries (-1, the ATS static-K padding) are dropped before
    the composition for every family but EViT (validate.py:213-214).  `Kept_Tokens_Abs` (heuristic patterns) and
    `Assignment_Maps` (the merging families) are copied through; the reference switches the soft maps, centre features and
    fusion assignment off (validate.py:178-180)."""
    out: Dict[str, dict] = {}
    prev: Optional[str] = None
    for stage_idx, stage in enumerate(reduction_count):
        name = f"Stage-{stage}"
        rec: dict = {}
        if "Kept_Tokens" in viz_data:
            cur = _np(viz_data["Kept_Tokens"][stage][i])
            if stage_idx == 0:
                rec["Kept_Token"] = cur
            else:
                rel = cur
                if "evit" not in model_name:
                    rel = rel[rel >= 0]
                rec["Kept_Token"] = out[prev]["Kept_Token"][rel]
        if "Kept_Tokens_Abs" in viz_data:
            rec["Kept_Token"] = _np(viz_data["Kept_Tokens_Abs"][stage][i])
        if "Assignment_Maps" in viz_data:
            rec["Assignment_Maps"] = _np(viz_data["Assignment_Maps"][stage][i])
        out[name] = rec
        prev = name
    return out


@torch.no_grad()
def validate(data_loader: Iterable, model, device, model_name: str, image_names: Sequence[str], keep_rate=None,
             reduction_loc=None) -> dict:
    """validate.py:59-276 without dataset/checkpoint plumbing: runs the loader, returns the `*_viz_results.json` dictionary
    (per image: top-5 `Predictions`, `Target`, the batch `Loss`, and -- with `model.viz_mode` -- the stage records)."""
    data = {"Model": model_name, "Ratio": keep_rate, "Location": reduction_loc}
    top1, top5 = _Meter(), _Meter()
    viz_mode = bool(getattr(model, "viz_mode", False))
    count = 0
    # one batch of lookahead, as in evaluate_multiclass (with viz_mode the forward runs synchronously: forward_async says so itself)
    launch = getattr(getattr(model, "module", model), "forward_async", None)

    def consume(pending, target, n):
        nonlocal count
        output = pending.result() if launch is not None else pending
        viz_data = None
        if viz_mode:
            output, viz_data = output
        loss = F.cross_entropy(output.float(), target)
        acc1, acc5 = accuracy(output, target, topk=(1, 5))
        _, pred = output.topk(min(5, output.shape[1]), 1, True, True)
        top1.update(acc1.item(), n)
        top5.update(acc5.item(), n)
        for i in range(n):
            rec = {"Predictions": _np(pred[i]), "Target": _np(target[i]), "Loss": loss.item()}
            if viz_mode:
                rec.update(image_records(model_name, model.get_reduction_count(), viz_data, i))
            data[image_names[count + i]] = rec
        count += n

    prev = None
    for images, target in data_loader:
        images = images.to(device, non_blocking=True)
        target = target.to(device, non_blocking=True)
        cur = (launch(images) if launch is not None else model(images), target, images.shape[0])
        if prev is not None:
            consume(*prev)
        prev = cur
    if prev is not None:
        consume(*prev)
    _check_status(model)
    data["Top1-Acc"] = round(top1.global_avg, 4)
    data["Top5-Acc"] = round(top5.global_avg, 4)
    if hasattr(model, "parameters"):
        data["Params"] = round(sum(p.numel() for p in model.parameters()) / 1e6, 2)
    return data


class NumpyArrayEncoder(json.JSONEncoder):
    """validate.py:26-30."""

    def default(self, obj):
        if isinstance(obj, np.ndarray):
            return obj.tolist()
        if isinstance(obj, np.generic):
            return obj.item()
        return json.JSONEncoder.default(self, obj)


def write_viz(viz_file: str, viz_data: dict) -> None:
    """validate.py:285-287."""
    with open(viz_file, "w") as f:
        json.dump(viz_data, f, cls=NumpyArrayEncoder, indent=4)
