"""Throughput of the data feed: FLAC files -> decode threads -> pinned -> H2D -> log-mel / splice kernels.
Uses copies of the one real recording this repo holds (tests/golden/ref_clip.flac, 8.89 s).
python tools/feed_bench.py [--threads 8] [--batch 32] [--batches 40]"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from caiman_asr_amd.data.frontend import LogMelFrontend  # noqa: E402
from caiman_asr_amd.data.loader import AudioBatchLoader  # noqa: E402
from caiman_asr_amd.data.sampler import SamplerUtt  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--threads", type=int, default=8)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--batches", type=int, default=40)
args = ap.parse_args()
tmp = tempfile.mkdtemp(prefix="feed_bench_")
try:
    n = args.batch * args.batches
    for i in range(64):
        shutil.copy(os.path.join(ROOT, "tests", "golden", "ref_clip.flac"), os.path.join(tmp, f"c{i}.flac"))
    utts = [SamplerUtt(f"c{i % 64}.flac", i, 8.89) for i in range(n)]
    toks = {i: [1, 2, 3] for i in range(n)}
    fe = LogMelFrontend(device="cuda")
    loader = AudioBatchLoader(utts, toks, tmp, args.batch, fe, decode_threads=args.threads, prefetch=3)
    it = iter(loader)
    next(it)                     # warm-up batch (allocations, first kernel launches)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    got = 0
    for feats, f_lens, txt, t_lens in it:
        got += feats.shape[1]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"metric": "data feed (FLAC decode + on-device frontend)", "utterances_per_s": got / dt,
                      "audio_hours_per_s": got * 8.89 / dt / 3600, "decode_threads": args.threads,
                      "host_cpus": len(os.sched_getaffinity(0)), "batch": args.batch, "batches": args.batches - 1}))
finally:
    shutil.rmtree(tmp, ignore_errors=True)
