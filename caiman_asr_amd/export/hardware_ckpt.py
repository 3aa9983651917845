"""Training checkpoint -> hardware / inference checkpoint.

Format of training/caiman_asr_train/export/hardware_ckpt.py:135-167 (pinned by the reference's own fixture
training/tests/test_data/hardware_ckpt.pt): one `torch.save`d dict

    state_dict (= the EMA weights), epoch, step, best_wer, melmeans, melvars, melalpha (0.0),
    sentpiece_model (the .model file's bytes), ngram {binary, scale_factor}, version, rnnt_config

so weights trained here load in the reference's `val.py` / inference server and the other way round.
"""
import argparse
import math
import os
from pathlib import Path
from typing import Dict, Optional, Tuple

import torch

from caiman_asr_amd.export.config_schema import inference_only_config
from caiman_asr_amd.export.model_schema import check_model_schema
from caiman_asr_amd.lm.kenlm_ngram import find_ngram_path
from caiman_asr_amd.rnnt import config

HARDWARE_CKPT_VERSION = "1.14.0"   # hardware_ckpt.py:162


def load_mel_stats(train_cfg: Dict, ckpt: Dict, ckpt_path: str = "") -> Tuple[torch.Tensor, torch.Tensor]:
    """Dataset log-mel statistics; refuses a checkpoint whose normalisation ramp has not finished
    (hardware_ckpt.py:61-88)."""
    w = ckpt["logmel_norm_weight"]
    if not math.isclose(w, 1.0):
        raise AssertionError(
            f"logmel_norm_weight should be 1.0 but it is {w}. When this value is less than 1.0, the ramp period did "
            f"not complete during training.\nThis means WER could be improved as the trained model at '{ckpt_path}' "
            "does not expect NormType.DATASET_STATS at inference time.\n\nPlease --resume training from this "
            "checkpoint to some --training_steps greater than --norm_ramp_end_step and then run this script again "
            "with the newly trained checkpoint.")
    stats_dir = Path(train_cfg["input_val"]["filterbank_features"]["stats_path"])
    assert stats_dir.exists(), f"Stats directory {stats_dir} does not exist."
    return (torch.load(stats_dir / "melmeans.pt", map_location="cpu"),
            torch.load(stats_dir / "melvars.pt", map_location="cpu"))


def read_ngram_lm(cfg: Dict, skip_ngram: bool, override_ngram_path: Optional[str]):
    if skip_ngram:
        return None, None
    ngram_cfg = cfg["ngram"]
    path = override_ngram_path or find_ngram_path(ngram_cfg["ngram_path"])
    if path is None:
        raise FileNotFoundError(
            f"N-gram not found in {ngram_cfg['ngram_path']}. Ensure you have a valid binary n-gram, or pass the "
            "`--skip_ngram` argument to skip adding an ngram to your hardware checkpoint.")
    assert os.path.splitext(path)[1] == ".binary", (
        f"Invalid file format: {path}. Please provide a binary n-gram file.")
    with open(path, "rb") as f:
        return f.read(), ngram_cfg["scale_factor"]


def create_hardware_ckpt(ckpt: str, config_path: str, skip_ngram: bool = False,
                         override_ngram_path: Optional[str] = None) -> Dict:
    traincp = torch.load(ckpt, map_location="cpu", weights_only=False)
    train_cfg = config.load(config_path)
    melmeans, melvars = load_mel_stats(train_cfg, traincp, ckpt)
    spm_fn = train_cfg["tokenizer"]["sentpiece_model"]
    assert spm_fn, "Sentencepiece model file not found in config."
    with open(spm_fn, "rb") as f:
        spm_bytes = f.read()
    ngram_lm, ngram_sf = read_ngram_lm(train_cfg, skip_ngram, override_ngram_path)
    return {
        "state_dict": traincp["ema_state_dict"],   # so that val.py can load the file like a training checkpoint
        "epoch": traincp["epoch"],
        "step": traincp["step"],
        "best_wer": traincp["best_wer"],
        "melmeans": melmeans,
        "melvars": melvars,
        "melalpha": 0.0,
        "sentpiece_model": spm_bytes,
        "ngram": {"binary": ngram_lm, "scale_factor": ngram_sf},
        "version": HARDWARE_CKPT_VERSION,
        "rnnt_config": inference_only_config(train_cfg),
    }


def save_hardware_ckpt(hardcp: Dict, output_ckpt: str) -> None:
    torch.save(hardcp, output_ckpt, pickle_protocol=5)


def main(argv=None) -> None:
    ap = argparse.ArgumentParser(description="Gather training results into a hardware checkpoint")
    ap.add_argument("--ckpt", type=str, default="/results/RNN-T_best_checkpoint.pt")
    ap.add_argument("--config", type=str, required=True)
    ap.add_argument("--output_ckpt", type=str, default="/results/hardware_ckpt.pt")
    ap.add_argument("--skip_ngram", action="store_true")
    ap.add_argument("--override_ngram_path", type=str, default=None)
    ap.add_argument("--skip_state_dict_check", action="store_true")
    args = ap.parse_args(argv)
    hardcp = create_hardware_ckpt(args.ckpt, args.config, args.skip_ngram, args.override_ngram_path)
    if not args.skip_state_dict_check:
        check_model_schema(hardcp["state_dict"])
    save_hardware_ckpt(hardcp, args.output_ckpt)


if __name__ == "__main__":
    main()
