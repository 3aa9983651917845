"""Utterance order of a training run (SURVEY §8 f2, row e's partitioning rule).

Behaviour of training/caiman_asr_train/data/dali/sampler.py:71-713 and manifest_ratios.py: epochs without repeated
files that hold the requested share of every manifest, an ordering per epoch (as read / by duration / random /
duration buckets), the longest utterances pulled into the first batches so that an out-of-memory configuration
fails at step 0, and the arrangement that makes contiguous per-rank shards see whole batches round-robin.  Same
class names and constructor keywords; given the same seeded `numpy.random.Generator` the order is the reference's
utterance for utterance (`tests/golden/sampler.json`) -- shuffles are issued on Python lists of the same lengths
in the same sequence.

Unlike the reference nothing is written to /tmp for a DALI reader: `process_output_files` returns the global list
and `rank_shard` cuts a rank's contiguous part out of it (what DALI's `shard_id / num_shards` does with the file
list, pipeline.py:116-121,233-242).  Internally an utterance is an integer id into flat arrays.
"""
import heapq
from dataclasses import dataclass
from itertools import chain
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np

Utterance = Dict[str, Union[int, float]]   # {"label": int, "duration": float}
Manifest = Dict[str, Utterance]            # file name -> Utterance


@dataclass
class SamplerUtt:
    file_name: str
    label: int
    duration: float


@dataclass
class AbsoluteManifestRatios:
    ratios: List[float]


@dataclass
class RelativeManifestRatios:
    ratios: List[float]


@dataclass
class CanaryManifestRatios:
    exponent: float


ManifestRatios = Union[AbsoluteManifestRatios, RelativeManifestRatios, CanaryManifestRatios, None]


def build_manifest_ratios(train_manifest_ratios=None, relative_train_manifest_ratios=None,
                          canary_manifest_exponent=None) -> ManifestRatios:
    given = [x is not None for x in (train_manifest_ratios, relative_train_manifest_ratios, canary_manifest_exponent)]
    if sum(given) > 1:
        raise ValueError("At most one kind of manifest mode should be set")
    if train_manifest_ratios is not None:
        return AbsoluteManifestRatios(list(train_manifest_ratios))
    if relative_train_manifest_ratios is not None:
        return RelativeManifestRatios(list(relative_train_manifest_ratios))
    if canary_manifest_exponent is not None:
        return CanaryManifestRatios(canary_manifest_exponent)
    return None


def build_json_fracs(ratios: ManifestRatios, lengths: Sequence[int], durations: Optional[Sequence[float]] = None):
    """Weight of every manifest in an epoch (manifest_ratios.py:63-85)."""
    if ratios is None:
        return [float(n) for n in lengths]
    if isinstance(ratios, RelativeManifestRatios):
        assert len(ratios.ratios) == len(lengths)
        return [r * n for r, n in zip(ratios.ratios, lengths)]
    if isinstance(ratios, AbsoluteManifestRatios):
        assert len(ratios.ratios) == len(lengths)
        return list(ratios.ratios)
    if isinstance(ratios, CanaryManifestRatios):
        assert durations is not None
        total = sum(durations)
        return [(u / n) * (n / total) ** ratios.exponent for u, n in zip(lengths, durations)]
    raise ValueError(f"Invalid valid for manifest_ratios={ratios}")


def _chunks(seq: List[int], n: int) -> List[List[int]]:
    return [seq[i:i + n] for i in range(0, len(seq), n)]


class Sampler:
    def __init__(self, *, total_batches: Optional[int], batch_size: int, global_batch_size: Optional[int],
                 world_size: int, resume_step: int = 0, rng: Optional[np.random.Generator] = None,
                 pessimistic_first_batch: bool = True, dump_shard_lists: bool = False, randomize_n_epochs: int = 0):
        if global_batch_size is None:
            global_batch_size = batch_size * world_size
        self.world_size = world_size
        self.batch_size = batch_size
        self.dist_batch_size = batch_size * world_size
        self.global_batch_size = global_batch_size
        self.randomize_n_epochs = randomize_n_epochs
        if randomize_n_epochs > 0:
            assert rng is not None, "Randomize_n_epochs requires rng"
        assert global_batch_size % world_size == 0
        assert global_batch_size % batch_size == 0
        assert global_batch_size % self.dist_batch_size == 0
        self.total_utts = None if total_batches is None else total_batches * batch_size
        self.resume_step = resume_step
        self.rng = rng
        self.pessimistic_first_batch = pessimistic_first_batch
        self.dump_shard_lists = dump_shard_lists
        self._dataset_size: Optional[int] = None
        self._epoch_size: Optional[int] = None
        self._files: Optional[List[SamplerUtt]] = None
        self._dur: np.ndarray = np.zeros(0)

    # ---- what a subclass defines ------------------------------------------------------------------------------
    def is_sampler_random(self) -> bool:
        raise NotImplementedError

    def _order_epoch(self, epoch: List[int]) -> List[int]:
        raise NotImplementedError

    # ---- public surface ------------------------------------------------------------------------------------------
    @property
    def dataset_size(self) -> int:
        assert self._dataset_size is not None, "DatasetFile not initialized"
        return self._dataset_size

    @property
    def epoch_size(self) -> int:
        assert self._epoch_size is not None, "Epoch size not initialized"
        return self._epoch_size

    def make_file_list(self, output_files: List[Manifest], json_names: List[str], manifest_ratios: ManifestRatios = None):
        """Every rank computes the same list from the same seed (the reference computes it on rank 0 and
        broadcasts a path, sampler.py:246-261); `rank_shard` then takes this rank's part."""
        self._dataset_size = sum(len(m) for m in output_files)
        self._files, self._epoch_size = self.process_output_files(output_files, json_names, manifest_ratios)

    def read_file_list(self) -> List[str]:
        assert self._files is not None, "File list not initialized!"
        return [u.file_name for u in self._files]

    def rank_shard(self, rank: int, files: Optional[List[SamplerUtt]] = None) -> List[SamplerUtt]:
        files = self._files if files is None else files
        assert files is not None, "File list not initialized!"
        n, w = len(files), self.world_size
        return files[n * rank // w: n * (rank + 1) // w]

    def process_output_files(self, output_files: List[Manifest], json_names: List[str],
                             manifest_ratios: ManifestRatios = None) -> Tuple[List[SamplerUtt], int]:
        names: List[str] = []
        labels: List[int] = []
        durs: List[float] = []
        per_manifest: List[List[int]] = []
        for m in output_files:
            ids = list(range(len(names), len(names) + len(m)))
            for path, utt in m.items():
                names.append(path)
                labels.append(utt["label"])
                durs.append(float(utt["duration"]))
            per_manifest.append(ids)
        self._dur = np.asarray(durs, dtype=np.float64)
        epochs = self._build_epochs(per_manifest, names, json_names, manifest_ratios)
        epochs = [self._order_epoch(e) for e in epochs]
        if self.pessimistic_first_batch:
            epochs[0] = self._find_pessimistic_batch(epochs[0])
        if self.randomize_n_epochs > 0:
            if not (self.pessimistic_first_batch and self.randomize_n_epochs == 1):
                # sampler.py:206-215 replaces the affected epochs by one nested list in every other setting,
                # which its own sharding then rejects; say so up front
                raise ValueError("Cannot shard the epochs evenly: randomize_n_epochs is only usable as 1 together "
                                 "with pessimistic_first_batch (the reference fails likewise)")
            tail = epochs[0][self.global_batch_size:]   # the first global batch keeps its pessimistic content
            self.rng.shuffle(tail)
            epochs[0][self.global_batch_size:] = tail
        order = self._to_dali_order(epochs)
        files = [SamplerUtt(names[i], labels[i], durs[i]) for i in order]
        return files, (len(epochs[0]) if epochs else 0)

    # ---- epochs ------------------------------------------------------------------------------------------------
    def _utts_per_epoch(self, fracs: List[float], lens: List[int]) -> Tuple[int, List[int]]:
        total = sum(fracs)
        targets = [f / total for f in fracs]
        epoch_len = min(n / t for t, n in zip(targets, lens))
        per = [int(t * epoch_len) for t in targets]
        per = [u - u % self.global_batch_size for u in per]   # whole global batches: epochs and optimiser steps stay aligned
        assert all(0 < u <= n for u, n in zip(per, lens)), (
            f"Number of utterances in a manifest is smaller than global batch size={self.global_batch_size}")
        return -(-self.total_utts // sum(per)), per

    def _build_epochs(self, per_manifest: List[List[int]], names: List[str], json_names: List[str],
                      ratios: ManifestRatios) -> List[List[int]]:
        lens = [len(ids) for ids in per_manifest]
        if ratios is None and self.total_utts is None:   # exactly one pass over everything
            assert len(set(names)) == sum(lens), f"Duplicates in {json_names}"
            return [list(chain.from_iterable(per_manifest))]
        if self.total_utts is None:
            raise ValueError("Please provide total_batches or json_fracs")
        durations = [float(self._dur[ids].sum()) for ids in per_manifest]
        n_epochs, per_epoch = self._utts_per_epoch(build_json_fracs(ratios, lens, durations), lens)
        data = [list(ids) for ids in per_manifest]
        if self.is_sampler_random():
            for d in data:
                self.rng.shuffle(d)
        epochs = []
        for e in range(n_epochs):
            epoch: List[int] = []
            for d, u in zip(data, per_epoch):      # the e-th run of u utterances of the endlessly repeated manifest
                start = (e * u) % len(d)
                epoch.extend(d[(start + j) % len(d)] for j in range(u))
            assert len(set(epoch)) == len(epoch), "Repeated file(s) in epoch"
            epochs.append(epoch)
        return epochs

    # ---- worst case first (sampler.py:265-319) -------------------------------------------------------------------
    def _move_chunk_to_front(self, n: int, epoch: List[int]) -> List[int]:
        if len(epoch) <= n:
            return epoch
        sums = [sum(self._dur[i] for i in c) for c in _chunks(epoch, n)]   # left-to-right, as the reference adds them
        offset = n * max(range(len(sums)), key=lambda i: sums[i])
        for i in range(n):
            epoch[i], epoch[offset + i] = epoch[offset + i], epoch[i]
        return epoch

    def _find_pessimistic_batch(self, epoch: List[int]) -> List[int]:
        if len(epoch) <= self.global_batch_size:
            return epoch
        for n in (self.global_batch_size, self.dist_batch_size, self.batch_size):
            epoch = self._move_chunk_to_front(n, epoch)
        n_big = self.global_batch_size // self.batch_size
        top = heapq.nlargest(n_big, range(len(epoch)), lambda i: self._dur[epoch[i]])
        for b, k in enumerate(top):     # one of the longest utterances into each batch of the first global batch
            lo = b * self.batch_size
            j = max(range(lo, lo + self.batch_size), key=lambda i: self._dur[epoch[i]])
            assert self._dur[epoch[k]] >= self._dur[epoch[j]]
            epoch[j], epoch[k] = epoch[k], epoch[j]
        return epoch

    # ---- per-rank arrangement (sampler.py:321-361) ------------------------------------------------------------------
    def _to_dali_order(self, epochs: List[List[int]]) -> List[int]:
        if not epochs:
            return []
        n_drop = self.resume_step * self.batch_size
        if len(epochs) == 1:
            if self.world_size > 1:
                assert n_drop == 0, "Cannot resume single batch with multiple GPUs"
            return epochs[0][n_drop:]
        if any(len(e) % self.dist_batch_size for e in epochs):
            raise ValueError("Cannot shard the epochs evenly")
        shards: List[List[int]] = [[] for _ in range(self.world_size)]
        for epoch in epochs:
            for b, batch in enumerate(_chunks(epoch, self.batch_size)):
                shards[b % self.world_size].extend(batch)
        return list(chain.from_iterable(s[n_drop:] for s in shards))


class SimpleSampler(Sampler):
    """Manifest order."""

    def __init__(self, *, total_batches, batch_size, global_batch_size, world_size, resume_step: int = 0,
                 dump_shard_lists: bool = False):
        super().__init__(total_batches=total_batches, batch_size=batch_size, global_batch_size=global_batch_size,
                         world_size=world_size, resume_step=resume_step, rng=None, pessimistic_first_batch=False,
                         dump_shard_lists=dump_shard_lists)

    def _order_epoch(self, epoch):
        return epoch

    def is_sampler_random(self):
        return False


class SortedSampler(Sampler):
    """Longest first; with several ranks, dealt out so that every contiguous shard is itself sorted."""

    def __init__(self, *, total_batches, batch_size, global_batch_size, world_size, resume_step: int = 0,
                 dump_shard_lists: bool = False):
        super().__init__(total_batches=total_batches, batch_size=batch_size, global_batch_size=global_batch_size,
                         world_size=world_size, resume_step=resume_step, rng=None, pessimistic_first_batch=False,
                         dump_shard_lists=dump_shard_lists)

    def _order_epoch(self, epoch):
        by_len = sorted(epoch, key=lambda i: self._dur[i], reverse=True)
        if self.world_size > 1:
            by_len = list(chain.from_iterable(by_len[r::self.world_size] for r in range(self.world_size)))
        return by_len

    def is_sampler_random(self):
        return False


class RandomSampler(Sampler):
    def __init__(self, *, total_batches, batch_size, global_batch_size, world_size, resume_step: int,
                 rng: np.random.Generator, pessimistic_first_batch: bool = True, dump_shard_lists: bool = False):
        super().__init__(total_batches=total_batches, batch_size=batch_size, global_batch_size=global_batch_size,
                         world_size=world_size, resume_step=resume_step, rng=rng,
                         pessimistic_first_batch=pessimistic_first_batch, dump_shard_lists=dump_shard_lists)

    def _order_epoch(self, epoch):
        self.rng.shuffle(epoch)
        return epoch

    def is_sampler_random(self):
        return True


class BucketingSampler(Sampler):
    """Batches of similar duration: sort, cut into `num_buckets` buckets, shuffle inside each bucket, cut the buckets
    into all-rank batches and shuffle those (sampler.py:675-709)."""

    def __init__(self, *, total_batches, batch_size, global_batch_size, world_size, resume_step: int,
                 rng: np.random.Generator, num_buckets: int, pessimistic_first_batch: bool = True,
                 dump_shard_lists: bool = False, randomize_n_epochs: int = 0):
        super().__init__(total_batches=total_batches, batch_size=batch_size, global_batch_size=global_batch_size,
                         world_size=world_size, resume_step=resume_step, rng=rng,
                         pessimistic_first_batch=pessimistic_first_batch, dump_shard_lists=dump_shard_lists,
                         randomize_n_epochs=randomize_n_epochs)
        self.num_buckets = num_buckets

    def _order_epoch(self, utts):
        N = self.dist_batch_size
        assert len(utts) > 0, "Empty epoch"
        assert len(utts) % self.batch_size == 0, "Epoch not divisible by batch size"
        assert len(utts) % N == 0, "Batches not divisible by number of GPUs"
        self.rng.shuffle(utts)                       # random tie-break for the stable sort
        utts.sort(key=lambda i: self._dur[i])
        size = -(-len(utts) // self.num_buckets)
        size = max(-(-size // N) * N, N)
        buckets = _chunks(utts, size)
        assert len(buckets) <= self.num_buckets and all(len(b) % N == 0 for b in buckets)
        for b in buckets:
            self.rng.shuffle(b)
        batches = [c for b in buckets for c in _chunks(b, N)]
        self.rng.shuffle(batches)
        return list(chain.from_iterable(batches))

    def is_sampler_random(self):
        return True
