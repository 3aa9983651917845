"""Golden vectors for the state-reset segmenter from the REFERENCE's own Python:
training/caiman_asr_train/evaluate/state_resets/{core,batch,overlap_processing,timestamp}.py run on seeded inputs
-> tests/golden/state_resets.json (inputs + expected outputs, data only).  TEST INFRASTRUCTURE ONLY; runs in the authoring
container (it imports /root/reference).  Usage: python oracle/gen_golden_state_resets.py"""
import json
import os
import sys

sys.dont_write_bytecode = True
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden  # noqa: E402  (the in-memory stand-ins for beartype / jaxtyping, and the reference on sys.path)


def main():
    gen_golden._install_stubs()
    import types

    import numpy as np
    import torch

    # caiman_asr_train.utils.frame_width imports caiman_asr_train.rnnt.config for a function that is not used here
    # (encoder_output_frame_width); that module pulls in the DALI pipeline (nvidia.dali: absent).  An empty stand-in
    # for it lets the reference's own input_feat_frame_width (plain dict arithmetic) load unchanged.
    sys.modules["caiman_asr_train.rnnt.config"] = types.ModuleType("caiman_asr_train.rnnt.config")

    from caiman_asr_train.evaluate.state_resets.batch import (state_resets_merge_batched_segments,
                                                              state_resets_reshape_batched_feats)
    from caiman_asr_train.evaluate.state_resets.core import get_segmenting_info, get_state_resets_stats
    from caiman_asr_train.evaluate.state_resets.timestamp import FullStamp

    cfg = {"input_train": {"filterbank_features": {"window_stride": 0.01},
                           "frame_splicing": {"frame_stacking": 3, "frame_subsampling": 3}}}
    out = {"cfg": cfg}
    out["plan"] = [[a, o, s, *get_segmenting_info(all_frames=a, overlap_frames=o, segment_frames=s)]
                   for a in range(9, 230, 13) for (s, o) in ((8, 0), (8, 3), (25, 6), (40, 39)) if a >= s]
    out["stats"] = [[seg, ov, *get_state_resets_stats(seg, ov, cfg)] for seg, ov in ((15.0, 3.0), (0.5, 0.0), (1.23, 0.31), (60.0, 2.5))]

    rng = np.random.default_rng(7)
    cases = []
    for T, lens, seg, ov in ((50, [50, 31, 7], 0.3, 0.06), (41, [41], 0.24, 0.0), (90, [17, 90, 64, 90], 0.45, 0.15)):
        B, F = len(lens), 3
        feats = torch.tensor(rng.integers(-50, 50, size=(T, B, F)).astype(np.float32))
        for b, n in enumerate(lens):
            feats[n:, b] = 0
        nf, nl, meta = state_resets_reshape_batched_feats(seg, ov, cfg, feats, torch.tensor(lens, dtype=torch.int32))
        cases.append(dict(feats=feats.tolist(), lens=lens, sr_segment=seg, sr_overlap=ov, out_feats=nf.tolist(),
                          out_lens=[int(x) for x in nl], meta=[list(m) for m in meta]))
    out["reshape"] = cases

    merges = []
    for seed, full, eos, with_probs in ((1, False, None, True), (2, True, None, True), (3, True, 5, True), (4, False, 2, False)):
        r = np.random.default_rng(seed)
        meta = [(int(r.integers(1, 5)), 26, 6) for _ in range(3)]
        pred, stamps, probs = [], [], []
        for n, _, _ in meta:
            for _ in range(n):
                k = int(r.integers(0, 9))
                toks = [int(x) for x in r.integers(0, 8, size=k)]
                t = sorted(int(x) for x in r.integers(0, 13, size=k))
                pred.append(toks)
                stamps.append([[a, a + int(d)] for a, d in zip(t, r.integers(0, 3, size=k))] if full else t)
                probs.append([round(float(x), 4) for x in r.random(k)] if with_probs else [])
        stamp_in = [[FullStamp(a, b) for a, b in s] for s in stamps] if full else [list(s) for s in stamps]
        o_pred, o_t, o_p = state_resets_merge_batched_segments([list(p) for p in pred], stamp_in,
                                                               [list(p) for p in probs] if with_probs else [[] for _ in pred],
                                                               2, meta, eos)
        o_t = [[[t.model, t.user_perceived] if full else t for t in row] for row in o_t]
        merges.append(dict(pred=pred, timestamps=stamps, probs=probs, full=full, eos_idx=eos, with_probs=with_probs,
                           meta=[list(m) for m in meta], enc_time_reduction=2, out_pred=o_pred, out_timestamps=o_t, out_probs=o_p))
    out["merge"] = merges
    path = os.path.join(gen_golden.OUT, "state_resets.json")
    json.dump(out, open(path, "w"))
    print("written", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
