"""Random state passing (RSP): training on the concatenation of consecutive batches by carrying the encoder /
prediction-network state from one batch into the next, reset at random intervals.

Controller of training/caiman_asr_train/train_utils/rsp.py:17-104 (`rsp_seq_len_freq`, `rsp_delay`); the state
selection helpers of :108-205 live with the state types in caiman_asr_amd/rnnt/state.py and are re-exported here.
The carried states go straight into the layer-pipelined LSTM launches as row 0 of their [T+1, B, H] slabs
(rnnt_ext/custom_lstm/{stack,encoder_pipe}.py): carried run == concatenated run (tests/test_gpu_rsp.py).
"""
import random
from argparse import Namespace
from typing import Optional, Sequence, Tuple

from caiman_asr_amd.rnnt.state import (RNNTState, get_last_nonpadded_states, get_pred_net_state,  # noqa: F401
                                       maybe_get_last_nonpadded)


def is_random_state_passing_on(seq_len_freq: Sequence[float]) -> bool:
    """`rsp_seq_len_freq[i]` is the relative frequency of histories of i+1 batches: RSP is on when any history
    longer than one batch has weight."""
    return any(w > 0 for w in seq_len_freq[1:])


def set_rsp_delay_default(args: Namespace, log=print) -> None:
    """Unset `--rsp_delay`: start once the learning rate has decayed to 1/8 (three half-lives after the hold)."""
    if args.rsp_delay is not None:
        return
    args.rsp_delay = args.warmup_steps + args.hold_steps + 3 * args.half_life_steps
    log(f"--rsp_delay not set. Setting rsp_delay={args.rsp_delay} based on a learning rate schedule heuristic.")
    if args.training_steps < args.rsp_delay + 5000:
        log(f"WARNING: Training too short (steps = {args.training_steps}) to see a benefit from RSP. "
            f"Set --training_steps to >= {args.rsp_delay + 5000}.")


def rsp_config_checks(args: Namespace, cfg: dict) -> None:
    freq = args.rsp_seq_len_freq
    if min(freq) < 0 or max(freq) <= 0:
        raise AssertionError("rsp_seq_len_freq must be non-negative with at least one positive entry")
    if not is_random_state_passing_on(freq):
        return
    rnnt = cfg["rnnt"]
    assert rnnt["custom_lstm"], "State passing requires custom_lstm=True"
    for key in ("enc_batch_norm", "pred_batch_norm"):
        assert not rnnt[key], "State passing hasn't been implemented with batch norm yet"
    set_rsp_delay_default(args)


def generate_batch_history(seq_len_freq: Sequence[float]) -> int:
    """History length in batches, drawn with weights `seq_len_freq` (entry i <-> length i+1)."""
    return 1 + random.choices(range(len(seq_len_freq)), weights=seq_len_freq)[0]


def rsp_end_step(rnnt_state: Optional[RNNTState], loss_nan: bool, step: int, args: Namespace,
                 batches_until_history_reset: int) -> Tuple[Optional[RNNTState], int, bool]:
    """-> (state to carry into the next batch | None, countdown, RSP active).  The state is dropped when the loss
    was NaN (the state itself may be the cause), before `rsp_delay`, and when the drawn history length is used up."""
    active = is_random_state_passing_on(args.rsp_seq_len_freq) and step >= args.rsp_delay
    if active and not loss_nan:
        assert rnnt_state is not None, "random state passing is on but the model returned no state"
    carry = rnnt_state if (active and not loss_nan) else None
    batches_until_history_reset -= 1
    if batches_until_history_reset == 0:
        carry = None
        batches_until_history_reset = generate_batch_history(args.rsp_seq_len_freq)
    return carry, batches_until_history_reset, active
