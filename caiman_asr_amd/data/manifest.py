"""JSON manifests -> {file name: {label, duration}} + {label: transcript}
(training/caiman_asr_train/data/dali/utils.py:93-176; manifest layout utils.py:19-35: a list of
{"transcript", "files": [{"fname"}...], "original_duration"})."""
import json
import random
from typing import Callable, Dict, List, Optional, Tuple, Union

Manifest = Dict[str, Dict[str, Union[int, float]]]


def set_predicate(max_duration: float, max_transcript_len: float, min_duration: float = 0.05) -> Callable[[dict], bool]:
    """Keep an utterance when min_duration < duration <= max_duration and the transcript is short enough."""
    return lambda utt: (min_duration < utt["original_duration"] <= max_duration
                        and len(utt["transcript"]) < max_transcript_len)


def parse_json(json_path: str, start_label: int = 0, predicate: Callable[[dict], bool] = lambda utt: True
               ) -> Tuple[Manifest, Dict[int, str]]:
    """Labels are consecutive from `start_label` over the utterances that pass `predicate`; the LAST entry of
    `files` names the audio (utils.py:137-150)."""
    with open(json_path, "r") as f:
        entries = json.load(f)
    files: Manifest = {}
    transcripts: Dict[int, str] = {}
    label = start_label
    for utt in entries:
        if not predicate(utt):
            continue
        transcripts[label] = utt["transcript"]
        files[utt["files"][-1]["fname"]] = dict(label=label, duration=utt["original_duration"])
        label += 1
    return files, transcripts


def filter_files(files: Manifest, transcripts: Dict[int, str], n_utterances_only: Optional[int], seed: int
                 ) -> Tuple[Manifest, Dict[int, str]]:
    """Random subset of n utterances, relabelled 0..n-1; the same on every rank for the same seed
    (utils.py:153-176; `random.Random(seed).sample` over the items in manifest order)."""
    if n_utterances_only is None:
        return files, transcripts
    picked = random.Random(seed).sample(list(files.items()), min(n_utterances_only, len(files)))
    new_files = {name: {"label": i, "duration": info["duration"]} for i, (name, info) in enumerate(picked)}
    new_tr = {i: transcripts[info["label"]] for i, (_, info) in enumerate(picked)}
    return new_files, new_tr


def load_manifests(json_paths: List[str], predicate=lambda utt: True, n_utterances_only: Optional[int] = None,
                   seed: int = 0) -> Tuple[List[Manifest], Dict[int, str]]:
    """Several manifests with one label space (data_loader.py:259-300): labels continue across files."""
    out, transcripts, label = [], {}, 0
    for p in json_paths:
        files, tr = parse_json(p, label, predicate)
        out.append(files)
        transcripts.update(tr)
        label += len(files)
    if n_utterances_only is not None:
        merged = {k: v for m in out for k, v in m.items()}
        merged, transcripts = filter_files(merged, transcripts, n_utterances_only, seed)
        out = [merged]
    return out, transcripts
