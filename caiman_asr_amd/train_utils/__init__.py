"""Train-step layer: flat-arena LAMB + EMA, LR policy, train step, data-parallel gradient
exchange (mirror of training/caiman_asr_train/train_utils)."""
