"""Host side of the beam search alone: N streams driven with canned top-k answers (no GPU).  Times the library's
requests() / feed() per expansion.  python tools/beam_host_bench.py [streams] [frames]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from caiman_asr_amd.rnnt.beam_native import NativeBeamSearch  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 30
V, k = 8704, 4
pieces = ["<unk>"] + [("▁" if i % 3 == 0 else "") + "".join("abcdefghijklmnopqrstuvwxyz"[(i // 26 ** d) % 26] for d in range(3))
                      for i in range(1, V - 1)]
rng = np.random.default_rng(0)
# 4096 canned answers: mostly a confident blank, sometimes a confident token, sometimes a flat row
tab_s = np.empty((4096, k), np.float32)
tab_t = np.empty((4096, k), np.int32)
for i in range(4096):
    u = rng.random()
    if u < 0.75:
        p = np.array([0.9, 0.05, 0.03, 0.02]); t = [V - 1, *rng.integers(1, V - 1, 3)]
    elif u < 0.92:
        p = np.array([0.8, 0.1, 0.06, 0.04]); t = [rng.integers(1, V - 1), V - 1, *rng.integers(1, V - 1, 2)]
    else:
        p = np.array([0.3, 0.28, 0.22, 0.2]); t = [*rng.integers(1, V - 1, 3), V - 1]
    tab_s[i], tab_t[i] = np.log(p), t
tab_b = np.array([tab_s[i][list(tab_t[i]).index(V - 1)] for i in range(4096)], np.float32)
s = NativeBeamSearch(n, pieces, blank_idx=V - 1, max_expansions_per_frame=32)
all_streams = np.arange(n, dtype=np.int32)
t_req = t_feed = 0.0
n_exp = rounds = 0
salt = 0
for f in range(frames):
    s.push_frame(all_streams)
    while True:
        t0 = time.perf_counter()
        stream, frame, y, s_in, s_out = s.requests()
        t1 = time.perf_counter()
        if len(stream) == 0:
            break
        salt += 7
        idx = (stream.astype(np.int64) * 31 + frame * 17 + y + salt) % 4096
        sc, tk, bl = np.ascontiguousarray(tab_s[idx]), np.ascontiguousarray(tab_t[idx]), np.ascontiguousarray(tab_b[idx])
        t2 = time.perf_counter()
        s.feed(sc, tk, bl)
        t3 = time.perf_counter()
        if f >= 5:
            t_req += t1 - t0
            t_feed += t3 - t2
            n_exp += len(stream)
            rounds += 1
    s.take_raw_responses()
print(f"streams {n}: {n_exp / (frames - 5) / n:.2f} expansions per stream-frame, {rounds / (frames - 5):.1f} rounds per frame; "
      f"requests {1e9 * t_req / n_exp:.0f} ns, feed {1e9 * t_feed / n_exp:.0f} ns per expansion "
      f"({1e3 * (t_req + t_feed) / (frames - 5):.2f} ms per frame), threads {os.environ.get('CAIMAN_BEAM_THREADS', 'default')}")
