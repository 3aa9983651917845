"""Two data-parallel ranks on ONE GPU (gloo rendezvous; the nccl path differs only in the backend string) running the
REAL RNNT through FlatGradReducer with batch splitting and gradient accumulation: per-rank gradients after the exchange
must equal the single-process gradients of the concatenated global batch (the reference's check:
training/tests/rnnt/test_batch_split.py:155-245, there under torchrun + NCCL with DDP).  Covers the advisor's finding
that the round-1 reducer counted hook firings (joint parameters fire batch_split_factor times, every parameter
grad_accumulation_batches times) and all-reduced partial gradients."""
import json
import os
import socket
from argparse import Namespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
GLOBAL_IDX = [0, 1, 2, 1, 2, 0, 1, 0]     # 8 utterances drawn from the golden batch of 3; 4 per rank


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(tag):
    from caiman_asr_amd.rnnt.model import RNNT

    g = np.load(os.path.join(GOLD, f"rnnt_{tag}.npz"))
    sd = {k[3:]: g[k] for k in g.files if k.startswith("sd.")}
    cfg = dict(json.loads(str(g["cfg"])), custom_lstm=True, joint_apex_transducer="pack", joint_apex_relu_dropout=True)
    m = RNNT(n_classes=int(g["n_classes"]), **cfg)
    m.load_state_dict({k: torch.tensor(v) for k, v in sd.items()})
    return g, m.to("cuda").train()


def _batch(g, idx):
    x = torch.tensor(g["x"][:, idx], device="cuda")
    return x, torch.tensor(g["x_lens"][idx]), torch.tensor(g["y"][idx], device="cuda"), torch.tensor(g["y_lens"][idx])


def _worker(rank, world, port, out):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from caiman_asr_amd import _lib
        from caiman_asr_amd.rnnt.loss import ApexTransducerLoss, LossModifiers
        from caiman_asr_amd.train_utils.core import train_step
        from caiman_asr_amd.train_utils.distributed import FlatGradReducer, broadcast_parameters
        from caiman_asr_amd.train_utils.loop import TrainStepper
        from caiman_asr_amd.train_utils.optimizer import OptimizerWrapper, build_optimizer
        from caiman_asr_amd.train_utils.schedule import ConstantSchedule

        report = {}
        for tag, amp in (("tiny", False), ("mfma", True)):
            g, ref_model = _build(tag)
            V = int(g["n_classes"])
            loss_fn = ApexTransducerLoss(blank_idx=V - 1, eos_idx=None, star_idx=None, packed_input=True)
            mods = LossModifiers(delay_penalty=0.01, eos_penalty=0.0, star_penalty=1.0)
            # single process, the whole global batch in one step
            a1 = Namespace(grad_accumulation_batches=1, batch_split_factor=1, no_amp=not amp, num_gpus=1)
            loss_ref, nan, _ = train_step(ref_model, loss_fn, a1, *_batch(g, GLOBAL_IDX), None, None, mods)
            assert not nan
            ref = {n: p.grad.clone() for n, p in ref_model.named_parameters()}

            g, m = _build(tag)
            opt_args = Namespace(lr=4e-3, weight_decay=1e-2, beta1=0.9, beta2=0.999, clip_norm=1.0, ema=0.999)
            opt = build_optimizer(opt_args, m)
            broadcast_parameters(opt.flat_p)
            # small buckets: several collectives, launched out of order by the hooks
            red = FlatGradReducer(opt._params, opt._offsets, opt.flat_g, bucket_bytes=16 << 10).attach(m).guard_handoffs(opt)
            assert len(red.buckets) >= 3
            captured = {}

            class Capture(OptimizerWrapper):
                def step(self, total_norm=None):
                    if self.reducer is not None:
                        self.reducer.finish()
                    captured.update({n: p.grad.clone() for n, p in m.named_parameters()})
                    self.optimizer.step()

            a2 = Namespace(grad_accumulation_batches=2, batch_split_factor=2, no_amp=not amp, num_gpus=world)
            stepper = TrainStepper(m, loss_fn, a2, Capture(a2, opt, reducer=red), dp_scheduler=ConstantSchedule(0.01))
            mine = GLOBAL_IDX[4 * rank:4 * rank + 4]
            n_res = _lib.lib().caiman_lstm_resident_launches()
            assert stepper.micro_batch(*_batch(g, mine[:2])) is None
            rec = stepper.micro_batch(*_batch(g, mine[2:]))
            assert rec is not None
            torch.cuda.synchronize()
            worst = 0.0
            for n, gr in ref.items():
                scale = gr.abs().max().item() + 1e-6
                err = (captured[n] - gr).abs().max().item() / scale
                worst = max(worst, err)
                assert err <= (3e-2 if amp else 2e-4), (tag, n, err)
            assert opt.last_step_applied.item() == 1
            # every rank holds the same parameters after the step
            gathered = [torch.zeros_like(opt.flat_p) for _ in range(world)]
            dist.all_gather(gathered, opt.flat_p)
            assert torch.equal(gathered[0], gathered[1])
            report[tag] = (worst, int(_lib.lib().caiman_lstm_resident_launches() - n_res))

            # A NaN loss in the LAST micro-batch of a window under batch splitting (one rank's slice only): every rank
            # drops the window (train.py:279-284), nothing of it may be handed to the reducer, and the next window must
            # run as if nothing had happened (round-2 advisor finding: the final backward used to launch the buckets,
            # finish() was never called, and the next final backward raised "bucket already in flight").
            steps_before = int(opt._step.item())
            assert stepper.micro_batch(*_batch(g, mine[:2])) is None
            xb = list(_batch(g, mine[2:]))
            if rank == 1:
                xb[0] = torch.full_like(xb[0], float("nan"))
            assert stepper.micro_batch(*xb) is None, "a NaN window must not reach the optimiser"
            assert stepper.accumulated == 0 and not any(red._launched) and not any(red._ready)
            captured.clear()
            assert stepper.micro_batch(*_batch(g, mine[:2])) is None
            assert stepper.micro_batch(*_batch(g, mine[2:])) is not None
            torch.cuda.synchronize()
            assert int(opt._step.item()) == steps_before + 1
            assert all(torch.isfinite(v).all() for v in captured.values())

            # a hand-off timeout on ONE rank makes EVERY rank drop the step (guard_handoffs + caiman_lstm_resident_poison)
            if tag == "mfma":
                before = opt.flat_p.clone()
                prev = _lib.lib().caiman_lstm_resident_set_failures(1) if rank == 1 else None
                try:
                    stepper.micro_batch(*_batch(g, mine[:2]))
                    assert stepper.micro_batch(*_batch(g, mine[2:])) is not None
                    torch.cuda.synchronize()
                    assert opt.last_step_applied.item() == 0, f"rank {rank} applied a step another rank had to drop"
                    assert torch.equal(opt.flat_p, before)
                finally:
                    if rank == 1:
                        _lib.lib().caiman_lstm_resident_set_failures(prev)
            red.remove()
        out.put((rank, "ok", report))
    except Exception:  # pragma: no cover
        import traceback

        out.put((rank, "fail", traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_two_ranks_real_model_batch_split_and_accumulation_match_single_process():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=540) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, status, payload in res:
        assert status == "ok", f"rank {rank}: {payload}"
    # the bf16 model ran its encoder through the layer pipeline (weight-resident launches) on both ranks
    for rank, _, payload in res:
        assert payload["mfma"][1] > 0, payload


def _nccl_world1_worker(port, out):
    """The RCCL code path on the one GPU there is: a world-size-1 `nccl` process group, the reducer told to behave as if
    distributed, one real train step through encoder_pipe.  (The reference: NCCL DDP,
    training/caiman_asr_train/setup/train.py:190-196, setup/base.py:497-501.)"""
    import time

    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from caiman_asr_amd import _lib
        from caiman_asr_amd.rnnt.loss import ApexTransducerLoss, LossModifiers
        from caiman_asr_amd.train_utils import overlap
        from caiman_asr_amd.train_utils.core import train_step
        from caiman_asr_amd.train_utils.distributed import FlatGradReducer, broadcast_parameters
        from caiman_asr_amd.train_utils.optimizer import OptimizerWrapper, build_optimizer

        lib = _lib.lib()
        warm = torch.ones(1024, device="cuda")
        dist.all_reduce(warm)              # communicator set-up happens here, not inside the timed part below
        torch.cuda.synchronize()

        g, ref_model = _build("mfma")
        V = int(g["n_classes"])
        loss_fn = ApexTransducerLoss(blank_idx=V - 1, eos_idx=None, star_idx=None, packed_input=True)
        mods = LossModifiers(delay_penalty=0.01, eos_penalty=0.0, star_penalty=1.0)
        a1 = Namespace(grad_accumulation_batches=1, batch_split_factor=1, no_amp=False, num_gpus=1)
        batch = _batch(g, GLOBAL_IDX)
        loss_ref, nan, _ = train_step(ref_model, loss_fn, a1, *batch, None, None, mods)
        assert not nan
        ref = {n: p.grad.clone() for n, p in ref_model.named_parameters()}

        g, m = _build("mfma")
        assert m.encoder_pipe
        opt = build_optimizer(Namespace(lr=4e-3, weight_decay=1e-2, beta1=0.9, beta2=0.999, clip_norm=1.0, ema=0.999), m)
        broadcast_parameters(opt.flat_p)
        red = FlatGradReducer(opt._params, opt._offsets, opt.flat_g, bucket_bytes=16 << 10, force_distributed=True,
                              measure_exposed=True).attach(m)
        wrapper = OptimizerWrapper(a1, opt, reducer=red)
        assert red.active and red.world == 1
        assert red._stream_ordered, "the nccl backend must take the stream-ordered branch (h.wait() on the side stream)"
        assert red._guard is not None, "pairing a reducer with an optimiser arms the hand-off guard"
        assert len(red.buckets) >= 3

        # (1) the real step: hooks launch the buckets on the side stream during backward, finish() joins them
        n_res = lib.caiman_lstm_resident_launches()
        loss, nan, _ = train_step(m, loss_fn, a1, *batch, None, None, mods)
        assert not nan and red.launched_total >= 1     # hooks launched buckets during the backward pass
        red.finish()
        torch.cuda.synchronize()
        assert red.launched_total == len(red.buckets)
        assert lib.caiman_lstm_resident_launches() > n_res, "encoder_pipe ran no weight-resident launch"
        assert lib.caiman_lstm_resident_failures() == 0
        assert abs(loss - loss_ref) <= 1e-6 * abs(loss_ref)
        worst = 0.0
        for n, p in m.named_parameters():
            err = (p.grad - ref[n]).abs().max().item() / (ref[n].abs().max().item() + 1e-12)
            worst = max(worst, err)
            assert err <= 1e-6, (n, err)      # a mean over one rank is the identity
        wrapper.optimizer.step()
        torch.cuda.synchronize()
        assert opt.last_step_applied.item() == 1

        # (2) h.wait() inside the side-stream context is a STREAM wait: with ~100 ms of work queued in front of the
        # collectives the host gets through every launch long before the device does
        opt.zero_grad()
        a = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
        torch.cuda.synchronize()
        for _ in range(150):
            a @ a
        t0 = time.perf_counter()
        red.mark_ready(opt._params)            # every bucket but the guarded one is launched from here
        host_s = time.perf_counter() - t0
        pending = not red.comm_stream.query()
        # (3) fence_collectives(): whatever the main stream runs next (a weight-resident LSTM launch in the stacks) is
        # ordered behind the collectives queued so far
        e_coll = torch.cuda.Event(enable_timing=True)
        e_coll.record(red.comm_stream)
        overlap.fence_collectives()
        e_main = torch.cuda.Event(enable_timing=True)
        e_main.record(torch.cuda.current_stream())
        e_main.synchronize()
        fenced = e_coll.query()                # the main stream got past the fence only after the side stream's work
        order_ms = e_coll.elapsed_time(e_main)
        red.finish()
        torch.cuda.synchronize()
        assert host_s < 0.05, f"launching the collectives blocked the host for {host_s * 1e3:.1f} ms"
        assert pending, "the side stream was already idle: the non-blocking check did not see queued work"
        assert fenced and order_ms >= 0.0, (fenced, order_ms)
        # (4) weight-resident grids and a CU-holding "collective" in ONE job, over several steps (round-3 verdict, next 8):
        # the first parameter whose gradient is ready in a step (the stack that finishes its backward first) queues a kernel
        # that holds 32 CUs for 30 ms on the communication stream -- what RCCL's kernel does while it waits for a slow rank.
        # The other stack's backward (resident launches: every workgroup of a layer must be on the chip together) then has to
        # start BEHIND it: fence_collectives() is called at the head of every stack's backward; a spy records a main-stream
        # event right behind each call.  Asserted: every fence that followed the hold completed after the hold's end, no
        # hand-off timed out, the gradients are those of the run without any of this, and the step paid for the hold.
        from caiman_asr_amd.rnnt_ext.custom_lstm import encoder_pipe as _ep  # noqa: F401  (the stacks look `overlap` up at call time)

        HOLD_US, fences, holds = 30000, [], []
        real_fence = overlap.fence_collectives

        def spy_fence():
            real_fence()
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream())
            fences.append((len(holds), ev))

        def hold_on_first_ready(_param):
            if holds and holds[-1][0] == step_no[0]:
                return
            with torch.cuda.stream(red.comm_stream):
                _lib.check(lib.caiman_debug_occupy_cus(32, HOLD_US, _lib.stream()))
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(red.comm_stream)
            holds.append((step_no[0], ev))

        step_no = [0]
        opt.zero_grad()
        train_step(m, loss_fn, a1, *batch, None, None, mods)       # the same step without a hold: the gradients to reproduce
        red.finish()
        torch.cuda.synchronize()
        base = {n: p.grad.clone() for n, p in m.named_parameters()}
        overlap.fence_collectives = spy_fence
        overlap.register_grad_ready_callback(hold_on_first_ready)
        try:
            n_res, step_ms = lib.caiman_lstm_resident_launches(), []
            for k in range(3):
                step_no[0] = k
                opt.zero_grad()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                loss_k, nan, _ = train_step(m, loss_fn, a1, *batch, None, None, mods)
                red.finish()
                torch.cuda.synchronize()
                step_ms.append((time.perf_counter() - t0) * 1e3)
                assert not nan
                for n, p in m.named_parameters():
                    err = (p.grad - base[n]).abs().max().item() / (base[n].abs().max().item() + 1e-12)
                    assert err <= 1e-6, (k, n, err)
            assert len(holds) == 3, len(holds)
            assert lib.caiman_lstm_resident_launches() > n_res and lib.caiman_lstm_resident_failures() == 0
            late = [(h, ev) for h, ev in fences if h >= 1]
            assert late, "no stack started its backward after a hold was queued: nothing was rehearsed"
            for h, ev in late:                       # the fence behind hold number h - 1 ended after that hold did
                assert holds[h - 1][1].elapsed_time(ev) >= 0.0, (h, holds[h - 1][1].elapsed_time(ev))
            assert min(step_ms) >= HOLD_US / 1e3 * 0.9, step_ms   # the hold is on the critical path: it was waited for
            exposed = red.exposed_ms()
        finally:
            overlap.fence_collectives = real_fence
            overlap._grad_ready_callbacks.remove(hold_on_first_ready)
        red.remove()
        out.put(("ok", {"worst_grad_err": worst, "host_launch_ms": host_s * 1e3, "buckets": len(red.buckets),
                        "backend": dist.get_backend(), "hold_step_ms": step_ms, "fences_behind_holds": len(late),
                        "exposed_ms_with_holds": exposed}))
    except Exception:  # pragma: no cover
        import traceback

        out.put(("fail", traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_rccl_path_runs_in_a_world_of_one_on_the_real_model():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    p = ctx.Process(target=_nccl_world1_worker, args=(_free_port(), out))
    p.start()
    status, payload = out.get(timeout=540)
    p.join(timeout=60)
    assert status == "ok", payload
    assert payload["backend"] == "nccl"
    assert payload["fences_behind_holds"] >= 2 and len(payload["hold_step_ms"]) == 3
