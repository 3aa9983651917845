"""Training checkpoints in the reference's format.

Mirror of training/caiman_asr_train/export/checkpointer.py:20-231: `<name>_step{N}_checkpoint.pt` /
`_best_` / `_last_` files holding `{epoch, step, best_wer, state_dict, ema_state_dict, optimizer, tokenizer_kw,
logmel_norm_weight}` (keys pinned by training/tests/export/test_checkpointer.py:76-132).  `state_dict` uses the
reference's parameter names (RNNT.state_dict drops the aliased joint_fc.* keys), so a checkpoint written by
either implementation loads in the other.  The inference hand-off file is written by `export/hardware_ckpt.py`.
"""
import glob
import os
import re
from collections import OrderedDict
from typing import Optional

import torch
import torch.distributed as dist


def _unwrap(model):
    return getattr(model, "module", model)


class Checkpointer:
    def __init__(self, save_dir, model_name, allow_partial_load: bool = False):
        self.save_dir = save_dir
        self.model_name = model_name
        self.allow_partial_load = allow_partial_load
        tracked = [(int(re.search(r"step(\d+)_", f).group(1)), f)
                   for f in glob.glob(f"{save_dir}/{self.model_name}_step*_checkpoint.pt")]
        self.tracked = OrderedDict(sorted(tracked, key=lambda t: t[0]))

    def save(self, model, ema_model, optimizer, epoch, step, best_wer, tokenizer_kw, logmel_norm_weight: float,
             config_path: Optional[str] = None, is_best: bool = False, is_last: bool = False,
             filepath: Optional[str] = None) -> None:
        """`ema_model`: a module, a ready state_dict, or None."""
        rank = 0
        if dist.is_initialized():
            dist.barrier()
            rank = dist.get_rank()
        if rank != 0:
            return
        if filepath:
            fpath = filepath
        elif is_best:
            fpath = os.path.join(self.save_dir, f"{self.model_name}_best_checkpoint.pt")
        elif is_last:
            fpath = os.path.join(self.save_dir, f"{self.model_name}_last_checkpoint.pt")
        else:
            fpath = os.path.join(self.save_dir, f"{self.model_name}_step{step}_checkpoint.pt")
        if ema_model is None:
            ema_sd = None
        elif isinstance(ema_model, dict):
            ema_sd = ema_model
        else:
            ema_sd = _unwrap(ema_model).state_dict()
        state = {
            "epoch": epoch, "step": step, "best_wer": best_wer,
            "state_dict": {k: v.detach().cpu().clone() for k, v in _unwrap(model).state_dict().items()},
            "ema_state_dict": None if ema_sd is None else {k: v.detach().cpu().clone() for k, v in ema_sd.items()},
            "optimizer": optimizer.state_dict() if optimizer is not None else None,
            "tokenizer_kw": tokenizer_kw, "logmel_norm_weight": logmel_norm_weight,
        }
        torch.save(state, fpath, pickle_protocol=5)
        if not is_best and not is_last:
            self.tracked[step] = fpath

    def last_checkpoint(self):
        tracked = list(self.tracked.values())
        if len(tracked) >= 1:
            try:
                torch.load(tracked[-1], map_location="cpu", weights_only=False)
                return tracked[-1]
            except Exception:
                print(f"Last checkpoint {tracked[-1]} appears corrupted.")
                if len(tracked) >= 2:
                    return tracked[-2]
        return None

    def _load(self, model, state_dict):
        missing, unexpected = model.load_state_dict(state_dict, strict=not self.allow_partial_load)
        if not unexpected and not missing:
            return
        if not set(k for k in state_dict.keys() if k not in unexpected):
            raise ValueError("No keys loaded from the checkpoint.")

    def load(self, fpath, model, ema_model, optimizer=None, meta=None):
        """Restores weights (+ EMA, optimizer, counters); returns the stored tokenizer_kw."""
        checkpoint = torch.load(fpath, map_location="cpu", weights_only=False)
        self._load(_unwrap(model), checkpoint["state_dict"])
        if ema_model is not None:
            key = "ema_state_dict" if checkpoint.get("ema_state_dict") is not None else "state_dict"
            self._load(_unwrap(ema_model), checkpoint[key])
        if optimizer is not None and checkpoint.get("optimizer") is not None:
            optimizer.load_state_dict(checkpoint["optimizer"])
        if meta is not None:
            meta["start_epoch"] = checkpoint.get("epoch")
            meta["best_wer"] = checkpoint.get("best_wer", meta["best_wer"])
            meta["step"] = checkpoint.get("step", meta["step"])
        return checkpoint.get("tokenizer_kw")


def ema_state_dict(model, optimizer):
    """state_dict whose parameter tensors are the optimiser's EMA values (the model the reference evaluates and
    exports, training/caiman_asr_train/train.py:58-64,417); buffers and frozen parameters pass through."""
    ema = optimizer.ema_tensors()
    by_id = {id(p): e for p, e in ema.items()}
    sd = _unwrap(model).state_dict()
    named = dict(_unwrap(model).named_parameters())
    out = OrderedDict()
    for k, v in sd.items():
        p = named.get(k)
        out[k] = by_id[id(p)].detach().clone() if p is not None and id(p) in by_id else v
    return out
