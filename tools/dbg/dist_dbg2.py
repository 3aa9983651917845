"""single process, world faked: exercise the protocol bookkeeping without collectives"""
import os, sys
sys.path.insert(0, os.getcwd())
from argparse import Namespace
import torch
from tests import test_gpu_distributed as T
from caiman_asr_amd.train_utils import distributed as D
from caiman_asr_amd.rnnt.loss import ApexTransducerLoss
from caiman_asr_amd.train_utils.loop import TrainStepper
from caiman_asr_amd.train_utils.optimizer import OptimizerWrapper, build_optimizer
from caiman_asr_amd.train_utils.schedule import ConstantSchedule

for tag, amp in (("tiny", False), ("mfma", True)):
    g, m = T._build(tag)
    V = int(g["n_classes"])
    loss_fn = ApexTransducerLoss(blank_idx=V - 1, eos_idx=None, star_idx=None, packed_input=True)
    opt = build_optimizer(Namespace(lr=4e-3, weight_decay=1e-2, beta1=0.9, beta2=0.999, clip_norm=1.0, ema=0.999), m)
    red = D.FlatGradReducer(opt._params, opt._offsets, opt.flat_g, bucket_bytes=16 << 10)
    # fake a 2-rank world for the bookkeeping: hooks on, launches replaced by a print
    red.world = 2
    names = {id(p): n for n, p in m.named_parameters()}
    for p in opt._params:
        red._hooks.append(p.register_post_accumulate_grad_hook(red._on_grad))
    D._hooks.register_grad_ready_callback(red._on_grad)
    def launch(b, red=red):
        if red._launched[b]: return
        red._launched[b] = True
        print(f"   -> launch bucket {b} {red.buckets[b]}")
    red._launch = launch
    orig = red._mark
    def mark(p, red=red, orig=orig):
        b = red.param_bucket.get(id(p))
        print(f"[mark] {names.get(id(p))} bucket {b} launched={red._launched[b]} ready={len(red._ready[b])}/{red.buckets[b][2]}")
        if id(p) in red._ready[b]:
            import traceback; traceback.print_stack(limit=12)
        return orig(p)
    red._mark = mark
    red.attach(m).guard_handoffs(opt)
    a2 = Namespace(grad_accumulation_batches=2, batch_split_factor=2, no_amp=not amp, num_gpus=1)
    stepper = TrainStepper(m, loss_fn, a2, OptimizerWrapper(a2, opt), dp_scheduler=ConstantSchedule(0.01))
    mine = T.GLOBAL_IDX[:4]
    print("== micro 1", tag)
    stepper.micro_batch(*T._batch(g, mine[:2]))
    print("== micro 2", tag)
    try:
        stepper.micro_batch(*T._batch(g, mine[2:]))
    except Exception as e:
        print("EXC", e)
    D._hooks.clear_grad_ready_callbacks()
    for h in red._hooks: h.remove()
