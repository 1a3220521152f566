#!/usr/bin/env python3
"""Record the reference's parameter groups (build container only):  python tests/golden/gen_param_groups.py

Imports the reference's optim.get_parameter_groups() unmodified (test-only timm stand-in, see gen_golden.py) and runs it on the
reference's own models; writes tests/golden/param_groups.json: per case the printed group dict (names, lr, weight_decay, fix_step)."""
import contextlib
import io
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(HERE, "timm_shim"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import torch  # noqa: E402

with contextlib.redirect_stdout(io.StringIO()):
    import models_act  # noqa: E402,F401
    import optim as ref_optim  # noqa: E402
from timm.models import create_model  # noqa: E402

from tests.test_finetune import CASES, ORDER_FACTORIES, case_args  # noqa: E402

out = {}
for key, (factory, kw) in CASES.items():
    torch.manual_seed(0)
    model = create_model(factory, pretrained=False, num_classes=10, drop_rate=0.0, drop_path_rate=0.0, drop_block_rate=None,
                         img_size=224, args=case_args())
    if kw.pop("freeze_patch_embed", False):
        for p in model.patch_embed.parameters():
            p.requires_grad = False
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        groups = ref_optim.get_parameter_groups(model, **kw)
    printed = json.loads(buf.getvalue().split("Param groups = ", 1)[1])
    assert [len(g["params"]) for g in groups] == [len(v["params"]) for v in printed.values()]
    out[key] = printed
# parameter (= optimizer-state) order and state_dict key order of one factory per family
order = {}
for factory in ORDER_FACTORIES:
    with contextlib.redirect_stdout(io.StringIO()):
        model = create_model(factory, pretrained=False, num_classes=10, drop_rate=0.0, drop_path_rate=0.0, drop_block_rate=None,
                             img_size=224, args=case_args())
    order[factory] = {"parameters": [n for n, _ in model.named_parameters()], "state_dict": list(model.state_dict().keys())}
out["__order__"] = order
with open(os.path.join(HERE, "param_groups.json"), "w") as f:
    json.dump(out, f, indent=1)
print({k: {g: len(v["params"]) for g, v in d.items()} for k, d in out.items() if k != "__order__"})
