"""Learning-rate policy: linear warm-up, hold, exponential half-life decay, floor.
Same function signature as training/caiman_asr_train/train_utils/lr.py:16-49."""


def lr_scale(step, warmup_steps, hold_steps, half_life_steps):
    if step < warmup_steps:
        return (step + 1) / (warmup_steps + 1)
    if step < warmup_steps + hold_steps:
        return 1.0
    return 0.5 ** ((step - warmup_steps - hold_steps) / half_life_steps)


def lr_policy(optimizer, initial_lr, min_lr, step, warmup_steps, hold_steps, half_life_steps):
    a = lr_scale(step, warmup_steps, hold_steps, half_life_steps)
    if type(initial_lr) is float:
        initial_lr = [initial_lr]
    assert len(initial_lr) == len(optimizer.param_groups)
    for lr, param_group in zip(initial_lr, optimizer.param_groups):
        param_group["lr"] = max(a * lr, min_lr)
