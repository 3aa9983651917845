"""Inference-only view of a training YAML (training/caiman_asr_train/export/config_schema.py:5-113).

The hardware checkpoint carries the subset of the config an inference server needs.  The reference expresses the
subset as pydantic models with `extra = "forbid"`; here it is one nested table: per section the REQUIRED fields
with their types, the fields that may be present but are dropped, and the sub-sections.  An unknown field raises,
as it does in the reference (a new training-only field must be added to the ignore set deliberately).
"""
from typing import Any, Dict

_MODEL_IGNORE = {
    "custom_lstm", "enc_batch_norm", "enc_dropout", "enc_freeze", "enc_rw_dropout", "forget_gate_bias",
    "hidden_hidden_bias_scale", "weights_init_scale", "joint_apex_relu_dropout", "joint_apex_transducer",
    "joint_dropout", "pred_batch_norm", "pred_dropout", "pred_rw_dropout", "quantize", "gpu_unavailable",
    "hard_activation_functions", "enc_lr_factor", "pred_lr_factor", "joint_enc_lr_factor", "joint_pred_lr_factor",
    "joint_net_lr_factor",
}

SCHEMA: Dict[str, Any] = {
    "ignore": {"input_train", "grad_noise_scheduler", "ngram", "user_tokens"},
    "fields": {},
    "sections": {
        "input_val": {
            "ignore": {"audio_dataset"},
            "fields": {},
            "sections": {
                "filterbank_features": {
                    "ignore": {"stats_path"},
                    "fields": {"dither": float, "n_fft": int, "n_filt": int, "normalize": str, "sample_rate": int,
                               "window": str, "window_size": float, "window_stride": float},
                    "sections": {},
                },
                "frame_splicing": {"ignore": set(), "fields": {"frame_stacking": int, "frame_subsampling": int},
                                   "sections": {}},
            },
        },
        "rnnt": {
            "ignore": _MODEL_IGNORE,
            "fields": {"enc_n_hid": int, "enc_post_rnn_layers": int, "enc_pre_rnn_layers": int,
                       "enc_stack_time_factor": int, "in_feats": int, "joint_n_hid": int, "pred_n_hid": int,
                       "pred_rnn_layers": int},
            "sections": {},
        },
        "tokenizer": {"ignore": {"sampling"}, "fields": {"labels": list, "sentpiece_model": str}, "sections": {}},
    },
}


class ConfigSchemaError(ValueError):
    pass


def _coerce(path: str, value, typ):
    if typ is float and isinstance(value, int) and not isinstance(value, bool):
        return float(value)
    if typ is int and isinstance(value, bool):
        raise ConfigSchemaError(f"{path}: expected int, got bool")
    if typ is int and isinstance(value, float) and value.is_integer():
        return int(value)
    if not isinstance(value, typ):
        raise ConfigSchemaError(f"{path}: expected {typ.__name__}, got {type(value).__name__}")
    return value


def _filter(cfg: Dict, spec: Dict, path: str) -> Dict:
    if not isinstance(cfg, dict):
        raise ConfigSchemaError(f"{path or 'config'}: expected a mapping")
    known = set(spec["fields"]) | set(spec["sections"]) | set(spec["ignore"])
    extra = sorted(set(cfg) - known)
    if extra:
        raise ConfigSchemaError(f"{path or 'config'}: extra fields not permitted: {extra}")
    out = {}
    for name, typ in spec["fields"].items():
        if name not in cfg:
            raise ConfigSchemaError(f"{path}{name}: field required")
        out[name] = _coerce(path + name, cfg[name], typ)
    for name, sub in spec["sections"].items():
        if name not in cfg:
            raise ConfigSchemaError(f"{path}{name}: field required")
        out[name] = _filter(cfg[name], sub, f"{path}{name}.")
    return out


def inference_only_config(config: Dict) -> Dict:
    """Loaded YAML dict -> the minimal dict stored as `rnnt_config` in a hardware checkpoint
    (hardware_ckpt.py:123-131)."""
    return _filter(config, SCHEMA, "")
