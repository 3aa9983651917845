import os, sys
sys.path.insert(0, os.getcwd())
import torch
from tests import test_gpu_distributed as T
from caiman_asr_amd.rnnt.loss import ApexTransducerLoss, get_packing_meta_data
g, m = T._build("mfma")
V = int(g["n_classes"])
names = {id(p): n for n, p in m.named_parameters()}
for p in m.parameters():
    p.grad = torch.zeros_like(p)
    p.register_post_accumulate_grad_hook(lambda p: print("HOOK", names[id(p)], float(p.grad.abs().sum())))
x, xl, y, yl = T._batch(g, [0, 1])
meta = get_packing_meta_data(xl, yl, 2, device="cuda")
loss_fn = ApexTransducerLoss(blank_idx=V - 1, eos_idx=None, star_idx=None, packed_input=True)
with torch.autocast("cuda", dtype=torch.bfloat16):
    (f, fl), (gg, gl), st = m.enc_pred(x, xl.cuda(), y, yl.cuda())
print("f grad_fn", f.grad_fn, f.grad_fn.next_functions)
node = f.grad_fn
seen = set()
def walk(n, d=0):
    if n is None or n in seen or d > 8: return
    seen.add(n)
    print("  " * d, type(n).__name__, getattr(n, "variable", None) is not None and names.get(id(n.variable)))
    for nx, _ in n.next_functions:
        walk(nx, d + 1)
walk(node)
print("=== backward of f")
import caiman_asr_amd.train_utils.overlap as ov
ov.register_grad_ready_callback(lambda p: print("NOTIFY", names[id(p)], float(p.grad.abs().sum())))
f.float().sum().backward()
torch.cuda.synchronize()
