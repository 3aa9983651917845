"""Training / validation batches from audio files: the reference's `DaliDataLoader`
(training/caiman_asr_train/data/dali/data_loader.py:45-404) with the DALI pipeline replaced by

    sampler shard of this rank -> host threads read + decode FLAC / WAV into pinned memory (csrc/audio_decode.hip)
    -> one H2D copy on a side stream -> log-mel kernel (a1) -> normalisation (a2) -> [SpecAugment (a3)]
    -> frame splicing (a4) -> (feats [T, B, F], feat_lens, txt [B, U], txt_lens)

A background thread keeps `prefetch` batches in flight, so decode and the frontend kernels overlap the training
step of the previous batch; the consumer waits on an event, not on the host.
"""
import os
import queue
import threading
from typing import Dict, Iterator, List, Optional, Sequence

import numpy as np
import torch

from caiman_asr_amd.data.audio import decode_files
from caiman_asr_amd.data.features import FrameSplicing, augment_splice_permute
from caiman_asr_amd.data.frontend import LogMelFrontend, MelFeatNormalizer
from caiman_asr_amd.data.sampler import SamplerUtt


class AudioBatchLoader:
    def __init__(self, utterances: Sequence[SamplerUtt], transcripts: Dict[int, Sequence[int]], dataset_path: str,
                 batch_size: int, frontend: LogMelFrontend, normalizer: Optional[MelFeatNormalizer] = None,
                 spec_augment=None, frame_stacking: int = 3, frame_subsampling: int = 3, max_duration: float = 16.7,
                 sample_rate: int = 16000, decode_threads: int = 8, prefetch: int = 2, drop_last: bool = True,
                 seed: int = 0, device: str = "cuda"):
        self.utts = list(utterances)
        self.transcripts = transcripts            # label -> token ids
        self.root = dataset_path
        self.B = batch_size
        self.frontend, self.normalizer, self.spec_augment = frontend, normalizer, spec_augment
        self.stacking, self.subsampling = frame_stacking, frame_subsampling
        self._splice = FrameSplicing(frame_stacking, frame_subsampling)
        self.sample_rate = sample_rate
        self.max_samples = int(round(max_duration * sample_rate)) + 1
        self.threads, self.prefetch, self.drop_last = decode_threads, prefetch, drop_last
        self.seed = seed
        self.device = torch.device(device)
        self.n_batches = len(self.utts) // batch_size if drop_last else -(-len(self.utts) // batch_size)

    def __len__(self):
        return self.n_batches

    # ---- host side: one batch of files into a pinned buffer ---------------------------------------------------------
    def _decode(self, batch: List[SamplerUtt], pinned: torch.Tensor):
        paths = [os.path.join(self.root, u.file_name) for u in batch]
        lens, rates = decode_files(paths, pinned.numpy(), self.threads)
        bad = [p for p, r in zip(paths, rates) if r != self.sample_rate]
        if bad:
            raise ValueError(f"{bad[0]} is sampled at a rate other than {self.sample_rate} Hz; resample the dataset "
                             "(the decoder does not resample)")
        return lens

    def _worker(self, out: "queue.Queue", stop: threading.Event):
        stream = torch.cuda.Stream(device=self.device)
        pool = [torch.empty(self.B, self.max_samples, dtype=torch.float32).pin_memory() for _ in range(self.prefetch + 1)]
        free_at = [None] * len(pool)     # event after which a pinned buffer may be overwritten
        try:
            for i in range(self.n_batches):
                if stop.is_set():
                    return
                batch = self.utts[i * self.B:(i + 1) * self.B]
                slot = i % len(pool)
                if free_at[slot] is not None:
                    free_at[slot].synchronize()
                lens = self._decode(batch, pool[slot])
                n_max = int(lens.max())
                with torch.cuda.stream(stream), torch.no_grad():
                    # whole rows: a contiguous pinned block is one asynchronous DMA, a strided slice is not
                    audio = pool[slot][: len(batch)].to(self.device, non_blocking=True)[:, :n_max].contiguous()
                    free_at[slot] = torch.cuda.Event()
                    free_at[slot].record(stream)
                    a_lens = torch.from_numpy(lens).to(self.device, non_blocking=True)
                    feats, f_lens = self.frontend(audio, a_lens, seed=self.seed + i)
                    if self.normalizer is not None:
                        feats = self.normalizer(feats, f_lens)
                    # SpecAugment masks + frame stacking / subsampling + the [T, B, F] permute as ONE kernel (what bench.py's
                    # timed step runs; bit-identical to the three modules, tests/test_gpu_specaugment.py).  Frame counts on the
                    # host from the sample counts: no device sync.
                    total = lens.astype(np.int64) + self.frontend.initial_pad
                    n_fr = np.where(total >= self.frontend.win_len, (total - self.frontend.win_len) // self.frontend.hop + 1, 0)
                    feats, f_lens_h = augment_splice_permute(self.spec_augment, self._splice, feats, f_lens,
                                                             torch.from_numpy(n_fr).int())
                    f_lens = f_lens_h.to(self.device, non_blocking=True)
                    toks = [self.transcripts[u.label] for u in batch]
                    t_lens = torch.tensor([len(t) for t in toks], dtype=torch.int32)
                    txt = torch.zeros(len(batch), max(int(t_lens.max()), 1), dtype=torch.int64)
                    for b, t in enumerate(toks):
                        txt[b, :len(t)] = torch.as_tensor(t, dtype=torch.int64)
                    txt, t_lens = txt.to(self.device, non_blocking=True), t_lens.to(self.device, non_blocking=True)
                    ready = torch.cuda.Event()
                    ready.record(stream)
                out.put((feats, f_lens, txt, t_lens, ready))
            out.put(None)
        except BaseException as e:   # surface decode / kernel errors in the consumer
            out.put(e)

    def __iter__(self) -> Iterator:
        q: "queue.Queue" = queue.Queue(maxsize=self.prefetch)
        stop = threading.Event()
        t = threading.Thread(target=self._worker, args=(q, stop), daemon=True)
        t.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    return
                if isinstance(item, BaseException):
                    raise item
                feats, f_lens, txt, t_lens, ready = item
                torch.cuda.current_stream().wait_event(ready)
                for x in (feats, f_lens, txt, t_lens):
                    x.record_stream(torch.cuda.current_stream())
                yield feats, f_lens, txt, t_lens
        finally:
            stop.set()
            while t.is_alive():
                try:
                    q.get_nowait()
                except queue.Empty:
                    pass
                t.join(timeout=0.05)
