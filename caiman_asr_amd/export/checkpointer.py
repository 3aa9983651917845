"""Training checkpoints in the reference's format.

Mirror of training/caiman_asr_train/export/checkpointer.py:20-231: `<name>_step{N}_checkpoint.pt` /
`_best_` / `_last_` files holding `{epoch, step, best_wer, state_dict, ema_state_dict, optimizer, tokenizer_kw,
logmel_norm_weight}` (keys pinned by training/tests/export/test_checkpointer.py:76-132).  `state_dict` uses the
reference's parameter names (RNNT.state_dict drops the aliased joint_fc.* keys), so MODEL weights (and the EMA
weights) written by either implementation load in the other.  The `optimizer` entry does not: this build's FusedLAMB
keeps its moments in flat arenas (`flat_m`, `flat_v`, `flat_ema`, `layout`), apex's in per-parameter `exp_avg` /
`exp_avg_sq`; loading a reference checkpoint restores weights, the EMA (copied into the optimiser's EMA arena by
parameter name) and converts the per-parameter moments into the arenas (`convert_foreign_optimizer_state`; a state
whose shapes do not match raises a clear error).  `best` / `last`
checkpoints also get their `.hw.pt` inference hand-off file (`export/hardware_ckpt.py`), as the reference's do
(checkpointer.py:107-143).
"""
import glob
import math
import os
import re
from collections import OrderedDict
from pathlib import Path
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
        if is_best or is_last:
            self.save_hardware_checkpoint(fpath, logmel_norm_weight, config_path, state["state_dict"])
        else:
            self.tracked[step] = fpath

    def save_hardware_checkpoint(self, fpath, logmel_norm_weight: float, config_path: Optional[str], model_state_dict) -> Optional[str]:
        """`<fpath>.hw.pt` next to a best / last checkpoint when the model has one of the supported inference schemas,
        the log-mel normalisation has finished its ramp to dataset statistics and the training config is known
        (reference checkpointer.py:107-143).  Returns the path written, or None with the reason printed."""
        from caiman_asr_amd.export.hardware_ckpt import create_hardware_ckpt, save_hardware_ckpt
        from caiman_asr_amd.export.model_schema import get_schema, return_schemas

        if config_path is None:
            print("Not saving hardware checkpoint: no training config path given")
            return None
        if get_schema(model_state_dict) not in return_schemas():
            print("Not saving hardware checkpoint as model is not supported on FPGA")
            return None
        if not math.isclose(logmel_norm_weight, 1.0):
            print(f"Not saving hardware checkpoint as {logmel_norm_weight=} is not yet 1.0")
            return None
        out = str(Path(fpath).with_suffix(".hw.pt"))
        save_hardware_ckpt(create_hardware_ckpt(fpath, config_path), out)
        print(f"Saved hardware checkpoint to {out}")
        return out

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
        opt_sd = checkpoint.get("optimizer")
        restored_ema = False
        if optimizer is not None and opt_sd is not None:
            if isinstance(opt_sd, dict) and "layout" in opt_sd and "flat_m" in opt_sd:
                optimizer.load_state_dict(opt_sd)
                restored_ema = opt_sd.get("flat_ema") is not None
            else:
                convert_foreign_optimizer_state(optimizer, opt_sd, fpath)
        if optimizer is not None and getattr(optimizer, "flat_ema", None) is not None and not restored_ema:
            # the EMA lives in the optimiser's arena: without this it would stay a clone of the weights taken when the
            # optimiser was built
            src = checkpoint.get("ema_state_dict") or checkpoint["state_dict"]
            load_ema_into_optimizer(_unwrap(model), optimizer, src)
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


@torch.no_grad()
def load_ema_into_optimizer(model, optimizer, ema_sd) -> int:
    """Copy a name-keyed EMA state_dict into the optimiser's flat EMA arena; returns the number of tensors copied."""
    views = {id(p): v for p, v in optimizer.ema_tensors().items()}
    n = 0
    for name, p in _unwrap(model).named_parameters():
        key = name if name in ema_sd else name.replace("joint_fc.", "joint_net.2.")
        v = views.get(id(p))
        if v is not None and key in ema_sd:
            v.copy_(torch.as_tensor(ema_sd[key]).to(v.device, v.dtype))
            n += 1
    return n


@torch.no_grad()
def convert_foreign_optimizer_state(optimizer, opt_sd, origin="checkpoint") -> None:
    """A torch-style optimiser state_dict ({"state": {i: {"exp_avg", "exp_avg_sq", ["step"]}}, "param_groups": [{"params":
    [i, ...], ["step"]}]} -- what apex FusedLAMB, the reference's optimiser, writes) -> this build's moment arenas.
    Parameter order inside the groups is `model.param_groups(lr)` order in both implementations."""
    try:
        state, groups = opt_sd["state"], opt_sd["param_groups"]
        ids = [i for g in groups for i in g["params"]]
    except (KeyError, TypeError):
        raise RuntimeError(f"{origin}: unrecognised optimizer state (keys {sorted(opt_sd) if isinstance(opt_sd, dict) else type(opt_sd)}); "
                           "expected this build's FusedLAMB arenas or a torch-style {'state', 'param_groups'} dict") from None
    params = optimizer._params
    if len(ids) != len(params):
        raise RuntimeError(f"{origin}: optimizer state holds {len(ids)} parameters, the model has {len(params)} trainable ones")
    step = 0
    for i, p, off in zip(ids, params, optimizer._offsets):
        st = state.get(i, state.get(str(i)))
        if st is None:
            continue
        m, v = st["exp_avg"], st["exp_avg_sq"]
        if tuple(m.shape) != tuple(p.shape):
            raise RuntimeError(f"{origin}: optimizer state entry {i} has shape {tuple(m.shape)}, parameter has {tuple(p.shape)}")
        n = p.numel()
        optimizer.flat_m[off:off + n].copy_(m.reshape(-1).to(optimizer.flat_m.device, torch.float32))
        optimizer.flat_v[off:off + n].copy_(v.reshape(-1).to(optimizer.flat_v.device, torch.float32))
        step = max(step, int(st.get("step", 0)))
    step = max([step] + [int(g.get("step", 0)) for g in groups])
    optimizer._step.fill_(step)
    for g, saved in zip(optimizer.param_groups, groups):
        for k in ("lr", "weight_decay", "betas", "eps"):
            if k in saved:
                g[k] = saved[k]
