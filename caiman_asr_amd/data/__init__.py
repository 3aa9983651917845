"""Feature path: log-mel frontend, normalisation, SpecAugment, frame splicing
(mirror of training/caiman_asr_train/data/features.py and the DALI graph tail)."""
