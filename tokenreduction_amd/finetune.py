"""Host logic of the fine-tune recipe: which parameters train at which learning rate, and the LR schedule.

Mirrors, for models built by tokenreduction_amd.create_model (same parameter names and `get_new_module_names()` as the reference):

    optim.py:39-100      get_parameter_groups()   decay / no_decay backbone groups at `bone_lr_scale` x lr, frozen for `fix_steps`
                                                   epochs; new modules (head, pos_embed, patch_embed, the method's own layers) at
                                                   full lr; optional constant cls / pos tokens
    train.py:416-419     linear LR scaling         lr * total_batch_size / lr_batch_normalizer
    engine.py:35-37      frozen_lr()               groups whose `fix_step` has not been reached run at lr 0
    scheduler_factory.py:10-66 + timm==0.4.12 timm/scheduler/cosine_lr.py (not vendored by the reference; its published algorithm is
                         restated in CosineSchedule): linear warm-up, cosine to lr_min, counted in optimizer STEPS when
                         `sched_in_steps` (t_in_epochs=False) -- `step_update(num_updates)` -- or in epochs -- `step(epoch)`.

    train.py:343-370     load_finetune_checkpoint() ingest of a DeiT-layout checkpoint: mismatching classifier dropped, position
                                                   embedding resized to the model's patch grid

Host-side bookkeeping only -- nothing here touches the GPU; the step it configures (forward, backward, the data-parallel gradient
mean) is the HIP training path (training.py, dp.py), driven by harness.train_one_epoch.
The parameter-group membership is pinned against lists recorded from the reference's own function
(tests/golden/param_groups.json, tests/golden/gen_param_groups.py); the cosine schedule has no reference run behind it (timm is
not installed here) and is checked against its closed form only.
"""
import math
from typing import Iterable, List, Optional, Sequence

import torch

NEW_MODULE_NAMES = ["head.weight", "head.bias", "head_dist.weight", "head_dist.bias", "pos_embed", "patch_embed"]   # optim.py:43


def scale_lr(lr: float, total_batch_size: int, lr_batch_normalizer: float) -> float:
    """train.py:416-419 (skipped by the reference under --unscale_lr)."""
    return lr * total_batch_size / lr_batch_normalizer


def get_parameter_groups(model, learning_rate, weight_decay=1e-5, bone_lr_scale=0.01, fix_steps=5, constant_cls=False,
                         constant_pos=False, skip_list=(), with_names=False):
    """optim.py:39-100.  Returns the optimizer's param-group dicts in first-seen order; `with_names=True` returns the same
    groups with parameter NAMES in `params` (what the reference prints)."""
    new_names = list(NEW_MODULE_NAMES)
    if hasattr(model, "get_new_module_names"):
        new_names.extend(model.get_new_module_names())
    groups = {}
    for name, param in model.named_parameters():
        if not param.requires_grad:
            continue
        if ("cls_token" in name or "dist_token" in name) and constant_cls:
            continue
        if "pos_embed" in name and constant_pos:
            continue
        no_decay = len(param.shape) == 1 or name.endswith(".bias") or name in skip_list
        if any(name in s for s in new_names) or any(s in name for s in new_names):
            group, wd, scale, fix = ("new_param_no_decay", 0, 1.0, 0) if no_decay else ("new_param", weight_decay, 1.0, 0)
        else:
            group, wd, scale, fix = ("no_decay", 0.0, bone_lr_scale, fix_steps) if no_decay else ("decay", weight_decay, bone_lr_scale, fix_steps)
        if group not in groups:
            groups[group] = {"weight_decay": wd, "params": [], "lr": learning_rate * scale, "fix_step": fix}
        groups[group]["params"].append(name if with_names else param)
    return list(groups.values())


def frozen_lr(param_groups: Iterable[dict], epoch: int) -> None:
    """engine.py:35-37: called before every step; a group is frozen (lr 0) until its `fix_step` epoch."""
    for g in param_groups:
        if epoch < g.get("fix_step", 0):
            g["lr"] = 0


class CosineSchedule:
    """timm 0.4.12 CosineLRScheduler as scheduler_factory.py:34-50 builds it (t_mul and decay_rate per cycle, cycle_limit,
    warm-up not a prefix).  `base_lrs` are the groups' initial lrs; with warm-up the groups start at `warmup_lr_init`."""

    def __init__(self, param_groups: Sequence[dict], t_initial: int, lr_min: float = 0.0, warmup_t: int = 0, warmup_lr_init: float = 0.0,
                 t_in_epochs: bool = True, t_mul: float = 1.0, decay_rate: float = 1.0, cycle_limit: int = 1):
        self.groups = list(param_groups)
        for g in self.groups:
            g.setdefault("initial_lr", g["lr"])
        self.base = [g["initial_lr"] for g in self.groups]
        self.t_initial, self.lr_min, self.warmup_t, self.warmup_lr_init = max(int(t_initial), 1), lr_min, warmup_t, warmup_lr_init
        self.t_in_epochs, self.t_mul, self.decay_rate, self.cycle_limit = t_in_epochs, t_mul, decay_rate, cycle_limit
        if warmup_t:
            self.warmup_steps = [(v - warmup_lr_init) / warmup_t for v in self.base]
            self._set([warmup_lr_init] * len(self.base))
        else:
            self.warmup_steps = [1.0] * len(self.base)

    @classmethod
    def from_args(cls, args, param_groups):
        """scheduler_factory.py:10-50 for args.sched == 'cosine' (no LR noise)."""
        in_epochs = not args.sched_in_steps
        per = 1 if in_epochs else args.num_steps_epoch
        return cls(param_groups, t_initial=args.epochs * per, lr_min=args.min_lr, warmup_t=args.warmup_epochs * per,
                   warmup_lr_init=args.warmup_lr, t_in_epochs=in_epochs, t_mul=getattr(args, "lr_cycle_mul", 1.0),
                   decay_rate=args.decay_rate, cycle_limit=getattr(args, "lr_cycle_limit", 1))

    def lr_at(self, t: int) -> List[float]:
        if t < self.warmup_t:
            return [self.warmup_lr_init + t * s for s in self.warmup_steps]
        if self.t_mul != 1:
            i = math.floor(math.log(1 - t / self.t_initial * (1 - self.t_mul), self.t_mul))
            t_i = self.t_mul ** i * self.t_initial
            t_curr = t - (1 - self.t_mul ** i) / (1 - self.t_mul) * self.t_initial
        else:
            i = t // self.t_initial
            t_i = self.t_initial
            t_curr = t - self.t_initial * i
        gamma = self.decay_rate ** i
        lr_min = self.lr_min * gamma
        if self.cycle_limit == 0 or i < self.cycle_limit:
            return [lr_min + 0.5 * (v * gamma - lr_min) * (1 + math.cos(math.pi * t_curr / t_i)) for v in self.base]
        return [self.lr_min for _ in self.base]

    def _set(self, lrs):
        for g, v in zip(self.groups, lrs):
            g["lr"] = v

    def step(self, epoch: int) -> None:
        if self.t_in_epochs:
            self._set(self.lr_at(epoch))

    def step_update(self, num_updates: int) -> None:          # engine.py:110-111
        if not self.t_in_epochs:
            self._set(self.lr_at(num_updates))


def load_finetune_checkpoint(model, checkpoint):
    """Ingest of a pre-trained checkpoint for fine-tuning (train.py:343-370): `checkpoint` is what torch.load returns for a
    DeiT-layout .pth ({"model": state_dict}) or the state dict itself.  A classifier whose shape does not fit the model is dropped,
    the position embedding is resized to the model's patch grid (class / distillation rows kept, the grid bicubically
    interpolated -- 14 x 14 -> 24 x 24 for a 224 -> 384 fine-tune), everything else is loaded non-strictly.  Returns load_state_dict's
    (missing_keys, unexpected_keys).  Host-side only: the executor re-packs its operand copies at the next forward (new addresses
    and values are detected by the packing key)."""
    import torch.nn.functional as F
    sd = dict(checkpoint["model"] if isinstance(checkpoint, dict) and "model" in checkpoint else checkpoint)
    own = model.state_dict()
    for k in ("head.weight", "head.bias", "head_dist.weight", "head_dist.bias"):
        if k in sd and k in own and sd[k].shape != own[k].shape:
            del sd[k]
    pos = sd.get("pos_embed")
    if pos is not None and pos.shape != model.pos_embed.shape:
        n_patches = model.patch_embed.num_patches
        n_extra = model.pos_embed.shape[-2] - n_patches
        old, new = int((pos.shape[-2] - n_extra) ** 0.5), int(n_patches ** 0.5)
        grid = pos[:, n_extra:].reshape(-1, old, old, pos.shape[-1]).permute(0, 3, 1, 2)
        grid = F.interpolate(grid, size=(new, new), mode="bicubic", align_corners=False)
        sd["pos_embed"] = torch.cat((pos[:, :n_extra], grid.permute(0, 2, 3, 1).flatten(1, 2)), dim=1)
    return model.load_state_dict(sd, strict=False)
