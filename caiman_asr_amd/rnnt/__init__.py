"""Model-level mirror of training/caiman_asr_train/rnnt (RNNT network, joint, loss wrapper,
state types, greedy decoder) on top of the gfx950 operator layer."""
