"""Average the weights of several training checkpoints
(training/caiman_asr_train/export/checkpoint_averaging.py:17-62): elementwise mean of `state_dict` and, when every
checkpoint has one, of `ema_state_dict`."""
from collections import OrderedDict
from typing import List, Optional, Tuple

import torch


def average_checkpoints(checkpoint_paths: List[str]) -> Tuple[OrderedDict, Optional[OrderedDict]]:
    assert checkpoint_paths, "no checkpoints given"
    sums = {"state_dict": OrderedDict(), "ema_state_dict": OrderedDict()}
    have_ema = True
    for path in checkpoint_paths:
        ckpt = torch.load(path, map_location="cpu", weights_only=False)
        for name in sums:
            sd = ckpt.get(name)
            if not sd:
                if name == "ema_state_dict":
                    have_ema = False
                continue
            for k, v in sd.items():
                if k in sums[name]:
                    sums[name][k] += v
                else:
                    sums[name][k] = v.clone()
    n = len(checkpoint_paths)
    for name in sums:
        for v in sums[name].values():
            if v.is_floating_point():
                v.div_(n)
            else:
                v.copy_(torch.div(v, n, rounding_mode="floor"))
    return sums["state_dict"], (sums["ema_state_dict"] if have_ema else None)
