"""world_size-2 gloo tests (CPU) of the data-parallel path: arena-slice gradient all-reduce driven by
post-accumulate hooks, parameter broadcast, utterance sharding, cross-rank NaN agreement.
(the reference's only distributed test re-launches itself under torchrun with NCCL,
training/tests/rnnt/test_batch_split.py:155-245.)"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from caiman_asr_amd.train_utils.core import is_loss_nan
        from caiman_asr_amd.train_utils.distributed import FlatGradReducer, broadcast_parameters, shard_utterances

        torch.manual_seed(100 + rank)  # different init per rank on purpose
        shapes = [(40, 30), (30,), (7, 5), (1000, 3), (11,)]
        offsets, total = [], 0
        for s in shapes:
            offsets.append(total)
            n = 1
            for d in s:
                n *= d
            total += (n + 63) // 64 * 64
        flat_p = torch.randn(total)
        flat_g = torch.zeros(total)
        params = []
        for s, o in zip(shapes, offsets):
            n = int(torch.tensor(s).prod())
            p = torch.nn.Parameter(flat_p[o:o + n].view(s))
            p.grad = flat_g[o:o + n].view(s)
            params.append(p)
        broadcast_parameters(flat_p)
        # tiny bucket size -> several buckets, exercised out of order
        red = FlatGradReducer(params, offsets, flat_g, bucket_bytes=2000 * 4, overlap=False)
        assert len(red.buckets) >= 2
        assert red.buckets[0][1] == total and red.buckets[-1][0] == 0  # tail first, whole arena covered
        results = {}
        for step in range(2):
            flat_g.zero_()
            x = torch.full((1,), float(rank + 1 + step))
            loss = sum((p * x).sum() * (i + 1) for i, p in enumerate(params[:-1]))  # last param unused
            loss.backward()
            red.finish()
            # d/dp = (i+1) * x  -> mean over ranks of x = (1+2)/2 + step
            for i, p in enumerate(params[:-1]):
                expect = (i + 1) * (1.5 + step)
                assert torch.allclose(p.grad, torch.full_like(p.grad, expect)), (step, i)
            assert torch.all(params[-1].grad == 0)
            results[step] = float(flat_g.sum())
        # parameters identical on both ranks after the broadcast
        gathered = [torch.zeros_like(flat_p) for _ in range(world)]
        dist.all_gather(gathered, flat_p)
        assert torch.equal(gathered[0], gathered[1])
        # NaN agreement: only rank 1 sees a NaN, both must decide to skip
        l = torch.tensor(float("nan") if rank == 1 else 1.0)
        assert is_loss_nan(l, world) is True
        assert is_loss_nan(torch.tensor(1.0), world) is False
        assert list(shard_utterances(10, rank, world)) == list(range(rank * 5, rank * 5 + 5))
        out.put((rank, "ok", results))
    except Exception as e:  # pragma: no cover
        import traceback

        out.put((rank, "fail", traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def _worker_protocol(rank, world, port, out):
    """Batch splitting x gradient accumulation through the reducer's step protocol (no_sync / mark_ready / finish) on
    a toy two-stage model: `enc` (one backward per micro-batch) feeding `joint` (one backward per slice).  The
    exchanged gradients must equal the gradients of the concatenated global batch, bucket by bucket."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from caiman_asr_amd.train_utils.distributed import FlatGradReducer

        torch.manual_seed(0)
        enc, joint = torch.nn.Linear(6, 5), torch.nn.Linear(5, 3)
        params = list(enc.parameters()) + list(joint.parameters())
        offsets, total = [], 0
        for p in params:
            offsets.append(total)
            total += (p.numel() + 63) // 64 * 64
        flat_g = torch.zeros(total)
        for p, o in zip(params, offsets):
            p.grad = flat_g[o:o + p.numel()].view(p.shape)
        red = FlatGradReducer(params, offsets, flat_g, bucket_bytes=64 * 4, overlap=False)
        assert len(red.buckets) >= 3
        accum, split, per = 2, 2, 4                      # per rank: 2 micro-batches of 4 rows, joint in 2 slices
        g = torch.Generator().manual_seed(5)
        data = torch.randn(world, accum, per, 6, generator=g)

        def loss_of(rows):
            return joint(torch.tanh(enc(rows))).pow(2).sum(1)

        # reference: the whole global batch at once, mean loss
        flat_g.zero_()
        with red.no_sync():
            loss_of(data.reshape(-1, 6)).mean().backward()
        ref = flat_g.clone()
        flat_g.zero_()
        for mb in range(accum):
            final = mb == accum - 1
            h = torch.tanh(enc(data[rank, mb]))
            h2 = h.detach().requires_grad_(True)
            for s in range(split):
                with red.no_sync():                      # the joint is back-propagated slice by slice
                    sl = slice(s * per // split, (s + 1) * per // split)
                    (joint(h2[sl]).pow(2).sum(1).sum() / (per * accum)).backward()
            if final:
                red.mark_ready(joint.parameters())
                h.backward(h2.grad)                      # hooks fire: enc's buckets go out as they complete
            else:
                with red.no_sync():
                    h.backward(h2.grad)
        red.finish()
        assert torch.allclose(flat_g, ref, atol=1e-6), (flat_g - ref).abs().max()
        # misuse: a second accumulation into a bucket whose collective is already in flight must raise
        flat_g.zero_()
        loss_of(data[rank, 0]).mean().backward()          # hooks on: every bucket is launched
        try:
            loss_of(data[rank, 1]).mean().backward()
            raised = False
        except RuntimeError as e:
            raised = "no_sync" in str(e)
        red.finish()
        assert raised
        # a dropped window (NaN loss: train.py:279-284): buckets may be in flight and parameters marked; reset() drains and
        # forgets them, and the next window runs as if nothing had happened (round-2 advisor finding)
        flat_g.zero_()
        loss_of(data[rank, 0]).mean().backward()          # hooks on: buckets launched, finish() will NOT be called
        assert any(red._launched) and any(red._ready)
        red.reset()
        assert not any(red._launched) and not any(red._ready) and not red._handles and not red._pass_cb_queued
        flat_g.zero_()
        with red.no_sync():
            loss_of(data[rank, 0]).sum().backward()
        red.mark_ready(params)
        red.finish()
        flat_one = flat_g.clone()
        flat_g.zero_()
        with red.no_sync():
            (loss_of(data[0, 0]).sum() + loss_of(data[1, 0]).sum()).mul(0.5).backward()
        assert torch.allclose(flat_one, flat_g, atol=1e-6)     # the mean over the two ranks' windows, nothing stale in it
        out.put((rank, "ok", float(ref.abs().sum())))
    except Exception:  # pragma: no cover
        import traceback

        out.put((rank, "fail", traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def _run2(target):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=100) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    for rank, status, payload in res:
        assert status == "ok", f"rank {rank}: {payload}"
    return res


def test_reducer_step_protocol_batch_split_and_accumulation_world2_gloo():
    res = _run2(_worker_protocol)
    assert res[0][2] == res[1][2]


def test_flat_grad_reducer_world2_gloo():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=100) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    for rank, status, payload in res:
        assert status == "ok", f"rank {rank}: {payload}"
    assert res[0][2] == res[1][2]  # identical reduced gradients on both ranks


def test_single_process_reducer_is_a_noop():
    from caiman_asr_amd.train_utils.distributed import FlatGradReducer

    flat = torch.zeros(128)
    p = torch.nn.Parameter(torch.ones(100))
    p.grad = flat[:100]
    red = FlatGradReducer([p], [0], flat, overlap=False)
    (p * 2).sum().backward()
    red.finish()
    assert torch.all(flat[:100] == 2)
