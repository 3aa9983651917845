"""Data-parallel gradient exchange over RCCL (torch.distributed backend "nccl" on ROCm).

The reference wraps the model in torch DDP (training/caiman_asr_train/setup/train.py:190-196):
25 MB buckets of copied gradients, mean all-reduce during backward.  Here the gradients already
live in ONE contiguous fp32 arena (train_utils/optimizer.py), so a bucket is just a slice of that
arena: no bucket copies, and the all-reduce of a slice is launched on a side stream as soon as
the last gradient inside it has been produced by the backward pass, overlapping the remaining
LSTM backward.  Arena order is parameter-group order (encoder, prediction, joint_enc,
joint_pred, joint_net) and backward produces gradients roughly in REVERSE arena order, so
buckets are cut from the tail.

xGMI is point-to-point (7 links x ~153 GB/s per GPU): few large messages beat many small ones,
so the default bucket is 64 MB (base model: 339 MB of fp32 gradients -> 6 collectives).
"""
from typing import List, Optional

import torch
import torch.distributed as dist


class FlatGradReducer:
    def __init__(self, params: List[torch.nn.Parameter], offsets: List[int], flat_grad: torch.Tensor,
                 process_group=None, bucket_bytes: int = 64 << 20, overlap: bool = True):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.flat = flat_grad
        self.overlap = overlap and flat_grad.is_cuda
        self.comm_stream = torch.cuda.Stream() if self.overlap else None
        self.main_stream = torch.cuda.current_stream() if self.overlap else None
        # cut buckets from the tail of the arena
        order = sorted(range(len(params)), key=lambda i: offsets[i], reverse=True)
        self.buckets = []  # (start, end, n_params)
        self.param_bucket = {}
        cur_end, cur_start, cur_n = flat_grad.numel(), flat_grad.numel(), 0
        limit = max(1, bucket_bytes // flat_grad.element_size())
        for i in order:
            cur_start = offsets[i]
            cur_n += 1
            self.param_bucket[id(params[i])] = len(self.buckets)
            if cur_end - cur_start >= limit:
                self.buckets.append((cur_start, cur_end, cur_n))
                cur_end, cur_n = cur_start, 0
        if cur_n:
            self.buckets.append((0, cur_end, cur_n))
        else:
            # leading alignment padding (none today) would belong to the last bucket
            pass
        self._pending = [b[2] for b in self.buckets]
        self._handles = []
        self._launched = [False] * len(self.buckets)
        self._hooks = []
        if self.world > 1:
            for p in params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
            # gradients produced on the side stream (train_utils/overlap.py) bypass autograd's accumulation:
            # they report here instead, and the collective then also waits for the side stream
            from caiman_asr_amd.train_utils import overlap

            self._overlap = overlap
            overlap.register_grad_ready_callback(self._on_grad)
            overlap.register_comm_stream(self.comm_stream)   # LSTM stacks fence against it (overlap.fence_collectives)

    # ---- per-step protocol: backward() ... finish() ---------------------------------------
    def _on_grad(self, p):
        if id(p) not in self.param_bucket:
            return
        b = self.param_bucket[id(p)]
        self._pending[b] -= 1
        if self._pending[b] == 0:
            self._launch(b)

    def _launch(self, b):
        if self._launched[b] or self.world == 1:
            return
        self._launched[b] = True
        s, e, _ = self.buckets[b]
        chunk = self.flat[s:e]
        if self.overlap:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            self.comm_stream.wait_stream(self.main_stream)
            for s_ in (self._overlap.side_streams() if getattr(self, "_overlap", None) else ()):
                self.comm_stream.wait_stream(s_)
            with torch.cuda.stream(self.comm_stream):
                self._handles.append(dist.all_reduce(chunk, group=self.group, async_op=True))
        else:
            self._handles.append(dist.all_reduce(chunk, group=self.group, async_op=True))

    def finish(self, average: bool = True):
        """Launch whatever was not triggered by hooks (frozen / unused parameters), wait for all
        collectives and turn the sums into means.  Call after backward(), before optimizer.step()."""
        if self.world > 1:
            for b in range(len(self.buckets)):
                self._launch(b)
            for h in self._handles:
                h.wait()
            if self.overlap:
                torch.cuda.current_stream().wait_stream(self.comm_stream)
            if average:
                self.flat.mul_(1.0 / self.world)
        self._handles = []
        self._pending = [b[2] for b in self.buckets]
        self._launched = [False] * len(self.buckets)

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


def broadcast_parameters(flat_params: torch.Tensor, src: int = 0, process_group=None):
    """DDP's constructor broadcast (rank 0 -> all), on the flat arena: one message."""
    if dist.is_initialized() and dist.get_world_size(process_group) > 1:
        dist.broadcast(flat_params, src=src, group=process_group)


def shard_utterances(n_items: int, rank: int, world: int):
    """Contiguous shard of a globally ordered (bucketed + shuffled) utterance list, as the
    reference's sampler / DALI reader do (training/caiman_asr_train/data/dali/sampler.py:225-262,
    pipeline.py:116-121,233-242)."""
    per = n_items // world
    return range(rank * per, (rank + 1) * per)
