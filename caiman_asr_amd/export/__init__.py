"""Checkpoint surface (mirror of training/caiman_asr_train/export): file naming, dict keys and load semantics of
the reference's `Checkpointer`, so checkpoints are interchangeable in both directions."""
