"""Data feed, CPU: utterance order per rank equals the reference samplers' on seeded synthetic
manifests (tests/golden/sampler.json, oracle/gen_golden.py)."""
import json
import os

import numpy as np
import pytest

from caiman_asr_amd.data import sampler as S

GOLD = os.path.join(os.path.dirname(__file__), "golden")
SAMPLER = json.load(open(os.path.join(GOLD, "sampler.json")))


def _build(case):
    kw = dict(case["kwargs"])
    if case["seed"] is not None:
        kw["rng"] = np.random.default_rng(case["seed"])
    ratios = None
    if case["ratios"] is not None:
        kind, val = case["ratios"]
        ratios = {"relative": S.RelativeManifestRatios, "absolute": S.AbsoluteManifestRatios,
                  "canary": S.CanaryManifestRatios}[kind](val)
    return getattr(S, case["klass"])(**kw), ratios


@pytest.mark.parametrize("tag", sorted(SAMPLER["cases"]))
def test_sampler_order_matches_reference(tag):
    case = SAMPLER["cases"][tag]
    smp, ratios = _build(case)
    files, epoch_size = smp.process_output_files([dict(m) for m in SAMPLER["manifests"]], SAMPLER["names"], ratios)
    assert epoch_size == case["epoch_size"]
    assert [u.label for u in files] == case["labels"]
    # what a rank reads: contiguous shards of whole batches, together the whole list
    smp.make_file_list([dict(m) for m in SAMPLER["manifests"]], SAMPLER["names"], ratios) if case["seed"] is None else None
    W = smp.world_size
    shards = [smp.rank_shard(r, files) for r in range(W)]
    assert [u.label for s in shards for u in s] == case["labels"]
    if len(files) % W == 0:
        assert len({len(s) for s in shards}) == 1


def test_sampler_properties_and_errors():
    man = [dict(m) for m in SAMPLER["manifests"]]
    smp = S.BucketingSampler(total_batches=200, batch_size=4, global_batch_size=32, world_size=4, resume_step=0,
                             rng=np.random.default_rng(0), num_buckets=6)
    smp.make_file_list(man, SAMPLER["names"])
    assert smp.dataset_size == 320 and smp.epoch_size == 320 and len(smp.read_file_list()) == 960
    files = smp._files
    durs = np.array([u.duration for u in files])
    # no file twice within an epoch, for any rank (a rank reads 80 utterances of each epoch)
    for r in range(4):
        shard = smp.rank_shard(r)
        assert len(shard) == 240
        for e in range(3):
            names = [u.file_name for u in shard[80 * e: 80 * (e + 1)]]
            assert len(set(names)) == 80
    # the first step of every rank holds the longest material: worst case first
    first_global = np.concatenate([[u.duration for u in smp.rank_shard(r)[:8]] for r in range(4)])
    assert first_global.max() == durs.max() and first_global.sum() >= np.sort(durs[:320])[-32:].sum() * 0.6
    # buckets: utterances of a batch have similar duration (std far below the corpus std)
    batch_std = np.mean([np.std([u.duration for u in smp.rank_shard(0)[i:i + 4]]) for i in range(80, 240, 4)])
    assert batch_std < 0.5 * durs.std()
    with pytest.raises(AssertionError):
        S.BucketingSampler(total_batches=10, batch_size=4, global_batch_size=6, world_size=1, resume_step=0,
                           rng=np.random.default_rng(0), num_buckets=2)
    with pytest.raises(ValueError, match="randomize_n_epochs"):
        S.BucketingSampler(total_batches=200, batch_size=4, global_batch_size=32, world_size=4, resume_step=0,
                           rng=np.random.default_rng(0), num_buckets=6, randomize_n_epochs=2).make_file_list(man, SAMPLER["names"])
    with pytest.raises(AssertionError, match="smaller than global batch size"):
        S.BucketingSampler(total_batches=10, batch_size=128, global_batch_size=128, world_size=1, resume_step=0,
                           rng=np.random.default_rng(0), num_buckets=2).make_file_list(man, SAMPLER["names"])
    with pytest.raises(ValueError, match="At most one"):
        S.build_manifest_ratios([1.0], [1.0], None)
    assert S.build_manifest_ratios(None, None, None) is None
    assert S.build_json_fracs(S.build_manifest_ratios(None, [2.0, 1.0], None), [10, 20]) == [20.0, 20.0]
