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


@pytest.mark.timeout(120)
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
