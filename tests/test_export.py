"""Hardware / inference checkpoint export, CPU.

Pins: the reference's RNNTInferenceConfigSchema applied to its own training YAMLs, and the structure of the
reference's own hardware-checkpoint fixture (tests/golden/export.json, oracle/gen_golden.py)."""
import copy
import json
import os

import numpy as np
import pytest
import torch
import yaml

from caiman_asr_amd.export.checkpoint_averaging import average_checkpoints
from caiman_asr_amd.export.config_schema import ConfigSchemaError, inference_only_config
from caiman_asr_amd.export.hardware_ckpt import HARDWARE_CKPT_VERSION, create_hardware_ckpt, main, save_hardware_ckpt
from caiman_asr_amd.export.model_schema import (CheckpointNotSupportedError, check_model_schema, check_schema_training,
                                                get_schema, return_schemas)
from tests.test_checkpoint import _mini

GOLD = os.path.join(os.path.dirname(__file__), "golden")
EXPORT = json.load(open(os.path.join(GOLD, "export.json")))


@pytest.mark.parametrize("name", sorted(EXPORT["configs"]))
def test_inference_only_config_matches_reference(name):
    case = EXPORT["configs"][name]
    if "error" in case:   # the reference's schema rejects this YAML (a field it does not know); so must this one
        with pytest.raises(ConfigSchemaError):
            inference_only_config(case["full"])
        return
    assert inference_only_config(copy.deepcopy(case["full"])) == case["inference"]


def test_inference_only_config_errors():
    full = copy.deepcopy(EXPORT["configs"]["base-8703sp"]["full"])
    full["rnnt"]["brand_new_knob"] = 1
    with pytest.raises(ConfigSchemaError, match="extra fields"):
        inference_only_config(full)
    del full["rnnt"]["brand_new_knob"], full["rnnt"]["enc_n_hid"]
    with pytest.raises(ConfigSchemaError, match="enc_n_hid: field required"):
        inference_only_config(full)
    full["rnnt"]["enc_n_hid"] = "wide"
    with pytest.raises(ConfigSchemaError, match="expected int"):
        inference_only_config(full)


def test_model_schema_variants():
    base, large = return_schemas()
    assert base == json.load(open(os.path.join(GOLD, "schema_base.json")))
    assert large == json.load(open(os.path.join(GOLD, "schema_large.json")))
    fake = {k: torch.empty(v, device="meta") for k, v in base.items()}
    check_model_schema(fake)
    g, sd, m = _mini()
    with pytest.raises(CheckpointNotSupportedError, match="BASE"):
        check_model_schema(sd)
    with pytest.raises(CheckpointNotSupportedError, match="skip_state_dict_check"):
        check_schema_training(sd, skip_state_dict_check=False)
    check_schema_training(sd, skip_state_dict_check=True)
    assert get_schema(sd) == {k: list(v.shape) for k, v in sd.items()}


def _write_training_run(tmp_path, logmel_norm_weight=1.0):
    from caiman_asr_amd.export.checkpointer import Checkpointer

    g, sd, m = _mini()
    m.load_state_dict(sd)
    ema = {k: v * 0.5 for k, v in m.state_dict().items()}
    Checkpointer(str(tmp_path), "RNN-T").save(m, ema, None, epoch=10, step=100, best_wer=3.05,
                                              tokenizer_kw={}, logmel_norm_weight=logmel_norm_weight)
    stats = tmp_path / "stats"
    stats.mkdir()
    torch.save(torch.tensor(np.load(os.path.join(GOLD, "melmeans.npy"))), stats / "melmeans.pt")
    torch.save(torch.tensor(np.load(os.path.join(GOLD, "melvars.npy"))), stats / "melvars.pt")
    (tmp_path / "sp.model").write_bytes(b"\\x0a\\x05piece")
    cfg = copy.deepcopy(EXPORT["configs"]["testing-1023sp_run"]["full"])
    cfg["tokenizer"]["sentpiece_model"] = str(tmp_path / "sp.model")
    cfg["input_val"]["filterbank_features"]["stats_path"] = str(stats)
    cfg["rnnt"].update(json.loads(str(g["rnnt_config"])))
    (tmp_path / "cfg.yaml").write_text(yaml.safe_dump(cfg))
    return ema, str(tmp_path / "RNN-T_step100_checkpoint.pt"), str(tmp_path / "cfg.yaml")


def test_create_hardware_ckpt_has_the_reference_layout(tmp_path):
    ema, ckpt, cfg = _write_training_run(tmp_path)
    hard = create_hardware_ckpt(ckpt, cfg, skip_ngram=True)
    ref = EXPORT["hardware_ckpt"]
    assert sorted(hard) == ref["keys"]
    assert (hard["epoch"], hard["step"], hard["best_wer"], hard["melalpha"]) == (ref["epoch"], ref["step"], ref["best_wer"],
                                                                               ref["melalpha"])
    assert hard["version"] == HARDWARE_CKPT_VERSION and sorted(hard["ngram"]) == ref["ngram_keys"]
    assert hard["ngram"] == {"binary": None, "scale_factor": None}
    assert isinstance(hard["sentpiece_model"], bytes) == ref["sentpiece_is_bytes"]
    assert list(hard["melmeans"].shape) == ref["mel_shape"] and str(hard["melmeans"].dtype) == ref["mel_dtype"]
    # same sections and field names as the reference's fixture
    def skeleton(d):
        return {k: skeleton(v) if isinstance(v, dict) else type(v).__name__ for k, v in d.items()}
    assert skeleton(hard["rnnt_config"]) == skeleton(ref["rnnt_config"])
    assert list(hard["state_dict"]) == list(ref["state_dict"])       # parameter names and order of the fixture
    for k, v in hard["state_dict"].items():                            # EMA weights are what ships
        assert torch.equal(v, ema[k])
    out = tmp_path / "hw.pt"
    save_hardware_ckpt(hard, str(out))
    back = torch.load(out, weights_only=False)
    assert sorted(back) == ref["keys"] and back["rnnt_config"] == hard["rnnt_config"]
    # the CLI refuses a state dict that is neither base nor large unless told otherwise
    with pytest.raises(CheckpointNotSupportedError):
        main(["--ckpt", ckpt, "--config", cfg, "--output_ckpt", str(tmp_path / "x.pt"), "--skip_ngram"])
    main(["--ckpt", ckpt, "--config", cfg, "--output_ckpt", str(tmp_path / "x.pt"), "--skip_ngram", "--skip_state_dict_check"])
    assert (tmp_path / "x.pt").exists()


def test_export_refuses_unfinished_norm_ramp_and_missing_ngram(tmp_path):
    _, ckpt, cfg = _write_training_run(tmp_path, logmel_norm_weight=0.6)
    with pytest.raises(AssertionError, match="ramp period did not complete"):
        create_hardware_ckpt(ckpt, cfg, skip_ngram=True)
    sub = tmp_path / "b"
    sub.mkdir()
    _, ckpt, cfg = _write_training_run(sub)
    full = yaml.safe_load(open(cfg))
    full["ngram"] = {"ngram_path": str(sub), "scale_factor": 0.05}
    open(cfg, "w").write(yaml.safe_dump(full))
    with pytest.raises(FileNotFoundError, match="skip_ngram"):
        create_hardware_ckpt(ckpt, cfg)
    (sub / "ngram.binary").write_bytes(b"kenlm")
    hard = create_hardware_ckpt(ckpt, cfg)
    assert hard["ngram"] == {"binary": b"kenlm", "scale_factor": 0.05}


def test_checkpoint_averaging(tmp_path):
    paths = []
    for i, scale in enumerate((1.0, 3.0)):
        sd = {"w": torch.full((2, 2), scale), "b": torch.tensor([scale])}
        p = tmp_path / f"c{i}.pt"
        torch.save({"state_dict": sd, "ema_state_dict": {k: v * 10 for k, v in sd.items()} if i == 0 else None}, p)
        paths.append(str(p))
    avg, ema = average_checkpoints(paths)
    assert torch.equal(avg["w"], torch.full((2, 2), 2.0)) and torch.equal(avg["b"], torch.tensor([2.0])) and ema is None
    avg, ema = average_checkpoints(paths[:1])
    assert torch.equal(ema["w"], torch.full((2, 2), 10.0))


def test_best_checkpoint_gets_its_hardware_file(tmp_path, monkeypatch, capsys):
    """Checkpointer.save(is_best / is_last) also writes `<name>.hw.pt` when the schema is supported, the norm ramp is
    complete and the config is known (reference checkpointer.py:107-143); otherwise it says why not."""
    from caiman_asr_amd.export import model_schema
    from caiman_asr_amd.export.checkpointer import Checkpointer

    ema, _, cfg = _write_training_run(tmp_path)
    full = yaml.safe_load(open(cfg))
    full["ngram"] = {"ngram_path": str(tmp_path), "scale_factor": 0.05}
    open(cfg, "w").write(yaml.safe_dump(full))
    (tmp_path / "ngram.binary").write_bytes(b"kenlm")
    g, sd, m = _mini()
    m.load_state_dict(sd)
    ck = Checkpointer(str(tmp_path), "RNN-T")
    ck.save(m, ema, None, 10, 100, 3.05, {}, 1.0, config_path=cfg, is_best=True)
    assert "not supported on FPGA" in capsys.readouterr().out and not (tmp_path / "RNN-T_best_checkpoint.hw.pt").exists()
    mini_schema = model_schema.get_schema(m.state_dict())
    monkeypatch.setattr(model_schema, "return_schemas", lambda: [mini_schema])
    ck.save(m, ema, None, 10, 100, 3.05, {}, 0.7, config_path=cfg, is_best=True)
    assert "is not yet 1.0" in capsys.readouterr().out
    ck.save(m, ema, None, 10, 100, 3.05, {}, 1.0, is_last=True)
    assert "no training config" in capsys.readouterr().out
    ck.save(m, ema, None, 10, 100, 3.05, {}, 1.0, config_path=cfg, is_best=True)
    hw = torch.load(tmp_path / "RNN-T_best_checkpoint.hw.pt", weights_only=False)
    assert hw["version"] == HARDWARE_CKPT_VERSION and hw["ngram"]["binary"] == b"kenlm"
    for k, v in hw["state_dict"].items():
        assert torch.equal(v, ema[k])
