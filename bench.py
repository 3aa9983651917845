"""bench.py — audio-hours/sec of base (85 M) RNN-T bf16 TRAINING on MI355X.

A "step" is one pass of the hot path over one batch of synthetic LibriSpeech-shaped input:
  on-device SpecAugment + frame splicing -> encoder / prediction LSTMs -> packed joint ->
  joint_fc -> transducer loss -> full backward -> (N>1: RCCL gradient all-reduce overlapped with
  backward) -> LAMB + EMA update.
Metric (BASELINE.json): audio-hours/sec = sum over ranks of sum(feat_lens) * 0.03 s / time / 3600,
the reference's `throughput-audio-secs-per-sec` (training/caiman_asr_train/train.py:379-382) / 3600.

  python bench.py --gpus N --steps K --warmup W
For N > 1 the driver launches it under torch.distributed.run, one rank per GPU over RCCL; started WITHOUT a
rendezvous environment (`python bench.py --gpus N`) it starts those N ranks itself, as child processes and before
this process has touched the GPU, and relays rank 0's line (the reference re-launches itself under torchrun the same
way, training/caiman_asr_train/train_utils/torchrun.py:9-31).  With fewer visible GPUs than ranks the ranks share
devices over gloo: a plumbing rehearsal, labelled as such in the line, never a measurement.
Rank 0 prints ONE JSON line with `roofline` (dominant hand-written kernel, HIP events), `lstm_resident`
(launches of the weight-resident LSTM kernels and their hand-off timeouts: must be 0), at N > 1
`allreduce_exposed_ms` (per step: how long the compute stream stood still waiting for the gradient exchange) and, at
N = 1, `cpu_baseline` (the CPU oracle timed on the host cores on bounded samples) and `decode` (streaming beam decode,
2 000 streams: the second half of BASELINE's metric; `--no-decode` skips it).
`CAIMAN_LSTM_RESIDENT=0` selects the per-timestep LSTM launches instead.
"""
import argparse
import json
import os
import subprocess
import sys
import time
from argparse import Namespace

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

FRAME_SECONDS = 0.03  # 10 ms hop x 3 frame subsampling (training/caiman_asr_train/utils/frame_width.py)
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_PEAK_TFLOPS = 2500.0  # dense bf16, same guide
PMC_FILE = "r04_base_pmc_traffic.json"   # builder-run counter passes of this round (falls back to nothing when absent)
MFMA_BF16_PEAK_TFLOPS = 2500.0

BASE_RNNT = dict(  # training/configs/base-8703sp.yaml:73-94
    in_feats=240, enc_n_hid=1024, enc_pre_rnn_layers=2, enc_post_rnn_layers=6, enc_stack_time_factor=2,
    enc_dropout=0.1, enc_batch_norm=False, enc_freeze=False, pred_n_hid=512, pred_rnn_layers=2, pred_dropout=0.3,
    pred_batch_norm=False, joint_n_hid=768, joint_dropout=0.3, joint_net_lr_factor=0.343,
    joint_apex_transducer="pack", joint_apex_relu_dropout=True, forget_gate_bias=1.0, custom_lstm=True,
    quantize=False, enc_rw_dropout=0.0, pred_rw_dropout=0.0)
N_CLASSES = 8704  # 8703 sentencepieces + blank
# training/configs/large-17407sp.yaml: the deltas of the 196 M model (BASELINE.json configs[3])
LARGE_RNNT = dict(BASE_RNNT, enc_n_hid=1536, pred_n_hid=768, joint_n_hid=1024, joint_net_lr_factor=0.243)
LARGE_N_CLASSES = 17408


def make_batches(n_batches, batch_size, seed, n_mels=80):
    """LibriSpeech-960-like utterances: duration ~ clip(N(12.3, 3.8), 1, 16.7) s,
    ~3.3 tokens/s, sorted into 6 duration buckets like the reference's BucketingSampler; every batch
    is drawn from one bucket.  Returns host tensors: log-mel [B, 80, T], frame lens, tokens, token lens."""
    rng = np.random.default_rng(seed)
    n_utts = 6 * (n_batches // 6 + 2) * batch_size  # every bucket can serve its share of the batches
    dur = np.clip(rng.normal(12.3, 3.8, size=n_utts), 1.0, 16.7)
    order = np.argsort(dur)
    buckets = np.array_split(order, 6)
    batches = []
    per_bucket = [list(rng.permutation(b)) for b in buckets]
    bi = 0
    while len(batches) < n_batches:
        pool = per_bucket[bi % 6]
        bi += 1
        assert len(pool) >= batch_size, "bucket pool exhausted"
        idx = [pool.pop() for _ in range(batch_size)]
        d = dur[idx]
        frames = np.floor(d * 100).astype(np.int64)  # 10 ms hop
        ntok = np.maximum(1, np.round(3.3 * d)).astype(np.int64)
        T = int(frames.max())
        g = torch.Generator().manual_seed(int(seed * 100003 + len(batches)))
        feats = torch.randn(batch_size, n_mels, T, generator=g)
        for i, f in enumerate(frames):
            feats[i, :, f:] = 0
        U = int(ntok.max())
        txt = torch.randint(0, N_CLASSES - 1, (batch_size, U), generator=g)
        batches.append((feats, torch.tensor(frames), txt, torch.tensor(ntok)))
    return batches


def make_pcm(frames, seed):
    """Synthetic 16 kHz PCM [B, max_samples] f32 whose log-mel has exactly `frames` frames per utterance with the frontend's
    geometry (25 ms window, 10 ms hop, 15 ms of initial padding: n_frames = n_samples // 160): what the data loader leaves
    resident in HBM for the frontend kernels (training/caiman_asr_train/data/dali/pipeline.py:359-470 runs them on the
    training GPU too)."""
    samples = frames * 160
    g = torch.Generator().manual_seed(int(seed))
    pcm = 0.1 * torch.randn(len(frames), int(samples.max()), generator=g)
    for i, n in enumerate(samples):
        pcm[i, int(n):] = 0
    return pcm, samples


def _cpu_time(fn, budget_s, max_iters=50):
    fn()   # warm-up
    t0, iters = time.perf_counter(), 0
    while iters < 1 or (time.perf_counter() - t0 < budget_s and iters < max_iters):
        fn()
        iters += 1
    return (time.perf_counter() - t0) / iters, iters


def cpu_baseline(model, threads):
    """The CPU oracle (oracle/model.py + oracle/rnnt_oracle.c: a port, fp32 network / f64 loss) timed on the host
    cores, SURVEY section 8(d): BASELINE config[0] (base-85M, 2 synthetic 1 s utterances) forward + RNN-T loss +
    backward on all of this job's cores (the headline `value`) and, under `variants`, the same on ONE thread (the
    reference pins OMP_NUM_THREADS=1, training/scripts/train.sh:17), forward + loss only, and a B = 4 slice of the GPU
    workload (4 LibriSpeech-shaped utterances of the batch the GPU is timed on)."""
    import math

    from oracle import model as omodel
    from oracle import native as onative

    sd = {k: v.detach().float().cpu().numpy() for k, v in model.state_dict().items()}
    cfg = dict(BASE_RNNT)
    blank = N_CLASSES - 1
    rng = np.random.default_rng(0)
    x = rng.standard_normal((34, 2, 240)).astype(np.float32)
    x_lens = np.array([34, 34])
    y = rng.integers(0, N_CLASSES - 1, size=(2, 5))
    y_lens = np.array([5, 3])

    def fwd_bwd(x=x, x_lens=x_lens, y=y, y_lens=y_lens):
        omodel.loss_and_grads(sd, cfg, x, x_lens, y, y_lens, blank, dtype=torch.float32)

    sd_t = {k: torch.tensor(v) for k, v in sd.items()}

    def fwd_only():
        with torch.no_grad():
            logits, f_lens = omodel.forward(sd_t, cfg, torch.as_tensor(x), torch.as_tensor(x_lens), torch.as_tensor(y),
                                            torch.as_tensor(y_lens))
        onative.transducer_forward(logits.double().numpy(), y.astype(np.int32), f_lens.numpy().astype(np.int32),
                                   y_lens.astype(np.int32), blank, delay_penalty=0.0, eos_penalty=0.0, eos_idx=None,
                                   star_lam=math.log(1.0), star_idx=None)

    audio_s = float(x_lens.sum()) * FRAME_SECONDS
    shape = "base-85M fp32, 2 utterances x 1.02 s (feats [34,2,240], U=[5,3])"
    variants = []
    torch.set_num_threads(threads)
    dt, iters = _cpu_time(fwd_bwd, 8.0)
    out = {"value": audio_s / dt / 3600.0, "unit": "audio-hours/sec", "cores": threads, "kind": "port",
           "sample": f"{shape}, fwd+loss+bwd, {iters} iterations of {dt:.2f} s"}
    dt, iters = _cpu_time(fwd_only, 3.0)
    variants.append({"value": audio_s / dt / 3600.0, "cores": threads, "sample": f"{shape}, fwd+loss only, {iters} x {dt:.2f} s"})
    torch.set_num_threads(1)
    dt, iters = _cpu_time(fwd_bwd, 5.0, max_iters=3)
    variants.append({"value": audio_s / dt / 3600.0, "cores": 1, "sample": f"{shape}, fwd+loss+bwd, {iters} x {dt:.2f} s"})
    dt, iters = _cpu_time(fwd_only, 3.0, max_iters=3)
    variants.append({"value": audio_s / dt / 3600.0, "cores": 1, "sample": f"{shape}, fwd+loss only, {iters} x {dt:.2f} s"})
    torch.set_num_threads(threads)
    try:   # B = 4 slice of the GPU workload: the first four utterances of this run's first batch, ONE pass
        feats, frames, txt, ntok = make_batches(1, 4, seed=1)[0]
        t1 = (frames + 2) // 3
        T1 = int(t1.max())
        xs = torch.zeros(T1, 4, 240)
        for b in range(4):      # frame splicing (stack 3 / subsample 3) on the host, outside the timed pass
            f = feats[b, :, : int(frames[b])]
            pad = (-f.shape[1]) % 3
            f = torch.cat([f, f.new_zeros(80, pad)], 1)
            xs[: int(t1[b]), b] = f.t().reshape(-1, 240)
        t0 = time.perf_counter()
        omodel.loss_and_grads(sd, cfg, xs.numpy(), t1.numpy(), txt.numpy(), ntok.numpy(), blank, dtype=torch.float32)
        dt = time.perf_counter() - t0
        a4 = float(t1.sum()) * FRAME_SECONDS
        variants.append({"value": a4 / dt / 3600.0, "cores": threads,
                         "sample": f"base-85M fp32, B = 4 slice of the GPU workload ({a4:.0f} s of audio, "
                                   f"{int(((t1 + 1) // 2 * (ntok + 1)).sum())} lattice cells), fwd+loss+bwd, 1 x {dt:.1f} s"})
    except Exception as e:   # memory-bound hosts: the slice needs ~10 GB
        variants.append({"error": repr(e), "sample": "B = 4 slice of the GPU workload"})
    out["variants"] = variants
    return out


def launch_ranks(n):
    """`python bench.py --gpus N` without a rendezvous environment: start the N ranks as children of this process
    (torch.distributed.run, rendezvous on 127.0.0.1) and relay rank 0's JSON line.  Called before anything here has
    touched the GPU; the children are ordinary child processes, nothing is exec'ed over an initialised process."""
    import socket

    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    for ln in lines:
        if not ln.lstrip().startswith("{"):
            print(ln, file=sys.stderr)
    result = [ln for ln in lines if ln.lstrip().startswith("{")]
    if result:
        print(result[-1], flush=True)
    return proc.returncode if (proc.returncode or result) else 1


def decode_record(streams=2000, ticks=100, search_runs=5):
    """Second half of BASELINE's metric (configs[4]): streaming beam decode, `streams` concurrent real-time 16 kHz
    streams, beam 4 / temperature 1.4 / <= 8 symbols per frame, from audio.  Runs bench_decode.py as a child process with
    the committed calibration of the synthetic logits (profiles/decode_calibration.json: logit scale and blank bias that
    make seeded random weights emit speech-like token rates; the fit is deterministic, and the record carries the
    measured token rate so that the workload can be checked).  Then MEASURES the capacity: `search_runs` more child runs at
    other stream counts (a bracketing search that starts from the linear extrapolation of the 2 000-stream tick) and
    reports the largest count whose p99 tick stayed under the 60 ms of audio it consumes.  Every run is 100 ticks (6 s of
    audio: the regime the calibration was fitted on -- over longer runs the synthetic weights drift towards emitting blanks
    and the ticks get lighter); the p99 is the nearest-rank percentile (the 99th of 100 ticks, not the worst one)."""
    cal = json.load(open(os.path.join(ROOT, "profiles", "decode_calibration.json")))

    def run(n, n_ticks):
        cmd = [sys.executable, os.path.join(ROOT, "bench_decode.py"), "--decoder", "beam", "--streams", str(n), "--ticks",
               str(n_ticks), "--warmup", "10", "--from-audio", "--scale", repr(cal["logit_scale"]), "--blank-bias",
               repr(cal["blank_bias"])]
        proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
        lines = [ln for ln in proc.stdout.splitlines() if ln.lstrip().startswith("{")]
        if proc.returncode != 0 or not lines:
            return None, (proc.stderr or "no output")[-400:]
        return json.loads(lines[-1]), None

    t0 = time.perf_counter()
    d, err = run(streams, ticks)
    if d is None:
        return {"error": err}
    rec = {"metric": d["metric"], "streams": d["streams"], "real_time": d["real_time"], "tick_audio_ms": d["tick_audio_ms"],
           "tick_latency_ms": d["tick_latency_ms"], "ticks": d["ticks"], "ticks_over_60ms": d["ticks_over_60ms"],
           "tokens_per_encoder_frame": d["tokens_per_encoder_frame"], "beam_width": d["beam_width"],
           "temperature": d["temperature"], "max_symbols_per_step": d["max_symbols_per_step"], "input": d["input"],
           "max_stream_lag_frames": d["max_stream_lag_frames"], "frames_settled_by_cap_frac": d["frames_settled_by_cap_frac"],
           "calibration": "profiles/decode_calibration.json (logit scale / blank bias of the synthetic weights)"}
    # capacity: largest stream count measured real-time (p99 tick < 60 ms), smallest measured not real-time
    ok_n, ok_p99 = (streams, d["tick_latency_ms"]["p99"]) if d["real_time"] else (0, None)
    bad_n = None if d["real_time"] else streams
    tried = [{"streams": streams, "p50_ms": round(d["tick_latency_ms"]["p50"], 2), "p99_ms": round(d["tick_latency_ms"]["p99"], 2),
              "real_time": d["real_time"]}]
    n = min(int(d["max_streams_at_p99_linear_estimate"] * 2 // 1000 * 1000), 4 * streams) if d["real_time"] else streams // 2
    for _ in range(search_runs):
        if n <= ok_n or (bad_n is not None and n >= bad_n) or n < 100:
            break
        dn, err = run(n, ticks)     # as many ticks as the headline run
        if dn is not None and not dn["real_time"] and dn["tick_latency_ms"]["p50"] < 30.0:
            # the median tick uses less than half of its 60 ms and the p99 still missed: that is what a burst of stalled
            # ticks on the box looks like, not saturation -- measure this count once more and keep
            # the better run (both are listed)
            tried.append({"streams": n, "p50_ms": round(dn["tick_latency_ms"]["p50"], 2),
                          "p99_ms": round(dn["tick_latency_ms"]["p99"], 2), "real_time": False, "repeated": True})
            d2, err2 = run(n, ticks)
            if d2 is not None and d2["tick_latency_ms"]["p99"] < dn["tick_latency_ms"]["p99"]:
                dn = d2
        if dn is None:
            tried.append({"streams": n, "error": err[-120:]})
            bad_n = n
        else:
            tried.append({"streams": n, "p50_ms": round(dn["tick_latency_ms"]["p50"], 2),
                          "p99_ms": round(dn["tick_latency_ms"]["p99"], 2), "real_time": dn["real_time"]})
            if dn["real_time"]:
                ok_n, ok_p99 = n, dn["tick_latency_ms"]["p99"]
            else:
                bad_n = n
        n = int((ok_n + bad_n) / 2 // 500 * 500) if bad_n is not None else int(ok_n * 1.5 // 1000 * 1000)
    rec["max_streams_measured"] = {"real_time_at": ok_n, "p99_ms_there": None if ok_p99 is None else round(ok_p99, 2),
                                   "not_real_time_at": bad_n, "runs": tried,
                                   "criterion": "p99 of the 60 ms tick (frontend + encoder + beam search of every stream) < 60 ms"}
    rec["wall_s_including_model_setup"] = time.perf_counter() - t0
    return rec


def feed_beside_step(step, args, dev, first_index):
    """The reference's DALI pipeline runs on the training GPU beside the step (training/caiman_asr_train/data/dali/
    pipeline.py:359-470).  Here: the timed loop's own steps (same synthetic batches, so the two figures compare) run once
    more while AudioBatchLoader decodes FLAC files on 8 host threads and runs the frontend kernels on its side stream; the
    main stream takes one fed batch per step (waits for its `ready` event) and drops it.  -> record for the bench line."""
    import shutil
    import tempfile

    from caiman_asr_amd import _lib
    from caiman_asr_amd.data.frontend import LogMelFrontend
    from caiman_asr_amd.data.loader import AudioBatchLoader
    from caiman_asr_amd.data.sampler import SamplerUtt

    lib = _lib.lib()
    tmp = tempfile.mkdtemp(prefix="bench_feed_")
    try:
        clip = os.path.join(ROOT, "tests", "golden", "ref_clip.flac")    # the one real recording in the repo: 8.89 s
        for i in range(64):
            shutil.copy(clip, os.path.join(tmp, f"c{i}.flac"))
        n_steps = args.steps
        n_utts = args.batch * (n_steps + 6)
        utts = [SamplerUtt(f"c{i % 64}.flac", i, 8.89) for i in range(n_utts)]
        toks = {i: [1, 2, 3] for i in range(n_utts)}
        loader = AudioBatchLoader(utts, toks, tmp, args.batch, LogMelFrontend(device=str(dev)), decode_threads=8, prefetch=3,
                                  device=str(dev))

        def timed(with_feed):
            it = iter(loader) if with_feed else None
            if it is not None:
                next(it)                       # allocations + first launches of the frontend kernels
            torch.cuda.synchronize()
            fails0, in_flight, fed = lib.caiman_lstm_resident_failures(), [], 0
            t0 = time.perf_counter()
            for i in range(n_steps):
                if len(in_flight) >= 2:
                    in_flight.pop(0).synchronize()
                if it is not None:
                    feats, _, _, _ = next(it)  # the main stream now waits for this batch's frontend kernels
                    fed += feats.shape[1]
                step(first_index + i, first_index + i)
                done = torch.cuda.Event()
                done.record(torch.cuda.current_stream())
                in_flight.append(done)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / n_steps * 1e3
            if it is not None:
                it.close()
            return ms, fed, lib.caiman_lstm_resident_failures() - fails0

        alone, _, _ = timed(False)
        beside, fed, timeouts = timed(True)
        return {"ms_per_step_alone": alone, "ms_per_step_with_feed": beside, "slowdown": beside / alone - 1.0,
                "utterances_fed_per_step": fed / n_steps, "fed_audio_seconds_per_step": fed * 8.89 / n_steps,
                "handoff_timeouts_with_feed": int(timeouts), "decode_threads": 8,
                "what": "same steps timed twice in this process, without and with AudioBatchLoader running beside them "
                        "(FLAC decode on host threads, H2D, log-mel + splice kernels on a side stream, one batch taken per step)"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # 60 = ten passes over the six synthetic batches (1.5 s timed): the GPU boxes show a one-off ~20 ms stall in about one
    # run out of three (any step, either tree in an A/B: `step_ms_device` in the JSON line shows it); over 12 steps that
    # is +1.7 ms on the average, over 60 steps +0.3
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="utterances per GPU per step")
    ap.add_argument("--model", choices=["base", "large"], default="base",
                    help="base = the headline config (BASELINE configs[1]); large = the 196 M model of configs[3]")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-decode", action="store_true", help="skip the streaming-decode record (N = 1 only)")
    ap.add_argument("--encoder-pipe", type=int, default=1,
                    help="1: pre_rnn -> StackTime -> post_rnn as one layer pipeline (encoder_pipe.py); 2: with the prediction "
                         "network's steps in the same launches (measured: 38.58 vs 38.55 ms, no gain); 0: stack after stack")
    ap.add_argument("--main-priority", type=int, default=0,
                    help="run the step on a stream of this priority (-1 = above the side streams; 0 = the default stream; measured: no effect)")
    ap.add_argument("--debug-steps", action="store_true", help="sync + log wall time of every step (perturbs timing)")
    ap.add_argument("--sequential-augment", action="store_true",
                    help="SpecAugment, FrameSplicing and PermuteAudio as separate torch modules instead of the fused kernel (A/B)")
    ap.add_argument("--features-resident", action="store_true",
                    help="start the timed step from log-mel features resident in HBM (rounds 1-3) instead of resident 16 kHz PCM: "
                         "leaves caiman_logmel_forward + caiman_mel_normalize out of the clock (A/B)")
    ap.add_argument("--launch-sources", type=int, default=0, metavar="N",
                    help="after the timed loop, run N more steps under torch.profiler (with Python stacks) and write, per kernel "
                         "name and calling source line of this repo, launches and device time per step to stderr: where the "
                         "step's small torch kernels come from")
    ap.add_argument("--no-prewarm", action="store_true",
                    help="skip the allocator pre-warm passes over the largest batches (profile runs: with --steps 6 --warmup 0 the "
                         "process then executes exactly one pass over the six synthetic batches, the mix of the timed loop)")
    ap.add_argument("--feed", action="store_true",
                    help="after the timed loop, time the same steps again WHILE the data feed (AudioBatchLoader: 8 FLAC decode "
                         "threads, side-stream log-mel / normalise / splice kernels) produces one batch per step (N = 1 only)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus)     # nothing in this process has touched the GPU yet
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    rehearsal = None
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        n_dev = torch.cuda.device_count()      # does not initialise the GPU
        # the real path is RCCL, one rank per GPU; with fewer GPUs than ranks (a 1-GPU box) the ranks share devices over
        # gloo so that the N > 1 code path can be rehearsed: same code, different backend string, not a measurement
        backend = os.environ.get("CAIMAN_DIST_BACKEND", "nccl" if n_dev >= world else "gloo")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            local_rank = local_rank % max(n_dev, 1)
            dist.init_process_group(backend)
        if n_dev < world:
            rehearsal = f"{world} ranks sharing {n_dev} GPU(s) over {backend}: plumbing rehearsal, not a measurement"
            # weight-resident LSTM grids of two PROCESSES would each hold part of one chip and wait for the rest
            os.environ["CAIMAN_LSTM_RESIDENT"] = "0"
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from caiman_asr_amd import _lib
    from caiman_asr_amd.data.features import FrameSplicing, SpecAugment, augment_splice_permute
    from caiman_asr_amd.data.frontend import LogMelFrontend, MelFeatNormalizer, NormType, norm_ramp_params
    from caiman_asr_amd.rnnt.loss import ApexTransducerLoss, LossModifiers, get_packing_meta_data
    from caiman_asr_amd.rnnt.model import RNNT
    from caiman_asr_amd.train_utils.distributed import FlatGradReducer, broadcast_parameters
    from caiman_asr_amd.train_utils.lr import lr_policy
    from caiman_asr_amd.train_utils.optimizer import build_optimizer

    _lib.lib()  # fail loudly if the HIP library is missing

    torch.manual_seed(1234)  # identical initial weights on every rank (then broadcast anyway)
    global N_CLASSES
    rnnt_cfg = BASE_RNNT
    if args.model == "large":
        rnnt_cfg, N_CLASSES = LARGE_RNNT, LARGE_N_CLASSES
        args.no_cpu_baseline = True  # the CPU sample is defined on the base config
    model = RNNT(n_classes=N_CLASSES, **rnnt_cfg).to(dev)
    model.train()
    model.joint_fc_nt_backward = os.environ.get("CAIMAN_JOINT_NT", "1") != "0"
    model.encoder_pipe = args.encoder_pipe >= 1
    model.pred_in_encoder_pipe = args.encoder_pipe >= 2
    opt_args = Namespace(lr=4e-3, weight_decay=1e-2, beta1=0.9, beta2=0.999, clip_norm=1.0, ema=0.999)
    optimizer = build_optimizer(opt_args, model)
    initial_lrs = [g["lr"] for g in optimizer.param_groups]
    broadcast_parameters(optimizer.flat_p)
    reducer = None
    if world > 1:
        reducer = FlatGradReducer(optimizer._params, optimizer._offsets, optimizer.flat_g, measure_exposed=True)
        reducer.attach(model).guard_handoffs(optimizer)
    loss_fn = ApexTransducerLoss(blank_idx=N_CLASSES - 1, eos_idx=None, star_idx=None, packed_input=True,
                                 validate_first_n_remaining=0)
    spec = SpecAugment(freq_masks=2, min_freq=0, max_freq=20, time_masks=10, min_time=0, max_time=0.03)
    splice = FrameSplicing(frame_stacking=3, frame_subsampling=3)
    loss_mods = LossModifiers(delay_penalty=0.0, eos_penalty=0.0, star_penalty=0.75)

    n_distinct = min(args.steps + args.warmup, 12)
    host_batches = make_batches(n_distinct, args.batch, seed=1 + rank)
    # inputs resident in HBM before the timed region: 16 kHz PCM (default: the whole north-star step is on the clock, log-mel
    # frontend and mel normalisation included) or, with --features-resident, the log-mel features themselves
    if args.features_resident:
        dev_batches = [(f.to(dev), fl, t.to(dev), tl) for f, fl, t, tl in host_batches]
        frontend = normalizer = None
    else:
        dev_batches = []
        for j, (f, fl, t, tl) in enumerate(host_batches):
            pcm, samples = make_pcm(fl, seed=977 * (1 + rank) + j)
            dev_batches.append(((pcm.to(dev), samples.to(dev).to(torch.int32)), fl, t.to(dev), tl))
        frontend = LogMelFrontend(device=str(dev))
        # blended dataset / utterance statistics on the reference's ramp (mel_normalization.py:85-118; synthetic dataset stats)
        ramp = norm_ramp_params(NormType.BLENDED_STATS, 1632, 18000, 10880)
        normalizer = MelFeatNormalizer(torch.zeros(80), torch.ones(80), ramp[0], ramp[1], starting_ratio=0.25)
        normalizer.means, normalizer.stddevs = normalizer.means.to(dev), normalizer.stddevs.to(dev)

    executed = [0]   # steps run in this process (pre-warm + warm-up + timed [+ feed]): what a profiler's trace holds

    def step(i, global_step):
        executed[0] += 1
        feats, feat_lens_h, txt, txt_lens_h = dev_batches[i % n_distinct]
        lr_policy(optimizer, initial_lrs, 4e-4, global_step, 1632, 18000, 10880)
        feat_lens_d = feat_lens_h.to(dev, non_blocking=True)
        if frontend is not None:       # PCM -> log-mel [B, 80, T] -> normalised: the two frontend kernels, on the step's stream
            pcm, samples_d = feats
            feats, _ = frontend(pcm, samples_d, seed=global_step + 1)
            normalizer.step(global_step)
            feats = normalizer(feats, feat_lens_d)
        if args.sequential_augment:                        # the three feature processors one at a time (A/B)
            x, _ = spec((feats, feat_lens_d))
            x, lens_h = splice((x, feat_lens_h))
            x = x.permute(2, 0, 1).contiguous()
        else:   # SpecAugment masks + frame splicing + PermuteAudio as one kernel (lens on the host: no device sync): [T1, B, 240]
            x, lens_h = augment_splice_permute(spec, splice, feats, feat_lens_d, feat_lens_h)
        meta = get_packing_meta_data(lens_h, txt_lens_h, 2, device=dev)
        lens_d = lens_h.to(dev, non_blocking=True)
        txt_lens_d = txt_lens_h.to(dev, non_blocking=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            logits, logit_lens, _ = model(x, lens_d, txt, txt_lens_d, batch_offset=meta["batch_offset"],
                                          packed_batch=meta["packed_batch"])
            loss = loss_fn(logits, logit_lens, txt, txt_lens_d, meta["batch_offset"], meta["max_f_len"], loss_mods)
        del logits
        loss.backward()   # a NaN loss gives NaN gradients -> the optimiser skips the update on-device
        if reducer is not None:
            reducer.finish()
        optimizer.step(zero_grad=True)
        return loss.detach(), float(lens_h.sum()) * FRAME_SECONDS, meta["packed_batch"], int(txt_lens_h.sum()) + len(txt_lens_h)

    if args.main_priority != 0:
        # The critical path (LSTM step kernels, joint backward) is a chain of short kernels; the side streams carry
        # long GEMMs that would otherwise occupy every CU first.  A higher-priority main stream gets its workgroups
        # dispatched ahead of theirs.
        main_stream = torch.cuda.Stream(device=dev, priority=args.main_priority)
        main_stream.wait_stream(torch.cuda.current_stream())
        plain_step = step

        def step(i, global_step):  # noqa: F811
            with torch.cuda.stream(main_stream):
                return plain_step(i, global_step)

    def log(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Allocator pre-warm (untimed, not one of the W warm-up steps): one pass over the LARGEST batch so the
    # caching allocator owns blocks big enough for every bucket; otherwise the first visit of each bucket
    # inside the timed region pays hundreds of ms of hipMalloc, which is a property of a cold process,
    # not of the training step.
    def lattice_cells(j):   # rows of the packed joint / logits tensor: sum_b T2_b * (U_b + 1)
        frames, ntok = host_batches[j][1], host_batches[j][3]
        t2 = ((frames + 2) // 3 + 1) // 2
        return int((t2 * (ntok + 1)).sum())

    by_cells = max(range(n_distinct), key=lattice_cells)                       # largest logits / gradient tensors
    by_frames = max(range(n_distinct), key=lambda j: host_batches[j][0].numel())  # largest LSTM activations
    for j in ([] if args.no_prewarm else dict.fromkeys((by_cells, by_frames))):
        step(j, 0)
        step(j, 0)   # twice, back to back: the timed loop keeps two steps in flight, so it needs two steps' worth of blocks
    optimizer._step.zero_()
    torch.cuda.synchronize()
    log(f"model + {n_distinct} batches resident; warm-up")
    for i in range(args.warmup):
        last_loss = step(i, i)[0]
        torch.cuda.synchronize()
        log(f"warm-up step {i} done")
        if i == 0:
            loss_fn.t_loss.validate_lengths = False  # inputs validated once; no per-step host syncs
    # A full (generation-2) pass of Python's cycle collector over everything the set-up has allocated takes ~100 ms on this
    # host, and its trigger is a count of allocations: it fell on whichever step reached the count (step 14 of a run at 32
    # timesteps per chunk, step 10 at 24 -- inside or outside the 12 timed steps by accident).  What exists now is long-lived:
    # move it out of the collector's reach, as a training loop does once its model and loaders are built.
    import gc
    gc.collect()
    gc.freeze()
    # The timed loop lets the host run two steps ahead of the device, so the caching allocator must hold up to three steps'
    # activations where the synchronised warm-up steps needed one: without room in its pool a timed step stops for a
    # hipMalloc of several GB (measured: one step of 50 ms instead of 31 in some runs and not in others, +1.1 ms on the
    # 12-step average).  A training run reaches that high-water mark within its first steps; reach it here, before the clock.
    peak = torch.cuda.max_memory_allocated(dev)
    free_b, _total_b = torch.cuda.mem_get_info(dev)
    room = min(int(2.5 * peak), int(0.5 * free_b))
    if room > (1 << 20):
        pool_pad = torch.empty(room, dtype=torch.uint8, device=dev)
        del pool_pad
    barrier()
    if reducer is not None:
        reducer.exposed_ms()   # drop the warm-up steps' events
    if not args.no_kernel_timing:
        _lib.timing.enabled = True
        _lib.timing.sample_every = {"lstm_fwd": 16, "lstm_bwd": 16}   # see _lib._Timing: every bracket would cost 13 %
        _lib.timing.reset()
    audio_s, cells, pred_tokens = 0.0, 0, 0
    in_flight = []   # the host may run at most two steps ahead of the device (a real loop reads the loss now and then);
    #                  unbounded run-ahead makes the allocator hold every queued step's activations at once
    host_wait = host_issue = 0.0   # host blocked on the run-ahead limit | host queuing a step's commands
    step_ends = []                 # one event per step on the device clock: shows a single slow step inside the average
    issue_ms = []                  # host time to queue each step
    segs0 = torch.cuda.memory_stats(dev).get("segment.all.allocated", 0)   # hipMalloc calls of the caching allocator so far
    loop_start = torch.cuda.Event(enable_timing=True)
    loop_start.record(main_stream if args.main_priority != 0 else torch.cuda.current_stream())
    step_ends.append(loop_start)   # so that the first timed step has a predecessor too
    t0 = time.perf_counter()
    for i in range(args.steps):
        ts = time.perf_counter()
        if len(in_flight) >= 2:
            in_flight.pop(0).synchronize()
        tq = time.perf_counter()
        host_wait += tq - ts
        last_loss, a, c, ntok = step(args.warmup + i, args.warmup + i)
        pred_tokens += ntok
        host_issue += time.perf_counter() - tq
        issue_ms.append(round(1e3 * (time.perf_counter() - tq), 2))
        done = torch.cuda.Event(enable_timing=True)
        done.record(main_stream if args.main_priority != 0 else torch.cuda.current_stream())
        in_flight.append(done)
        step_ends.append(done)
        audio_s += a
        cells += c
        if args.debug_steps:
            tl = time.perf_counter()
            torch.cuda.synchronize()
            log(f"step {i}: launch {1e3 * (tl - ts):.1f} ms, total {1e3 * (time.perf_counter() - ts):.1f} ms, "
                f"audio {a:.0f} s, cells {c}")
    # the host polls the last step's event before the closing barrier + synchronize: a thread that sleeps in a blocking
    # synchronise is woken up to several ms late on some boxes (runs whose every step was normal on the device clock came out
    # 0.4-0.6 ms per step slower over 12 steps), and that is the host's wake-up latency, not the step
    while in_flight and not in_flight[-1].query():
        pass
    barrier()
    elapsed = time.perf_counter() - t0
    _lib.timing.enabled = False
    log(f"{args.steps} timed steps in {elapsed:.3f} s")

    exposed_ms = reducer.exposed_ms() / args.steps if reducer is not None else 0.0
    stats = torch.tensor([elapsed, audio_s, exposed_ms], dtype=torch.float64, device=dev)
    if world > 1:
        tmax = stats[0:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        asum = stats[1:2].clone()
        dist.all_reduce(asum, op=dist.ReduceOp.SUM)
        emax = stats[2:3].clone()
        dist.all_reduce(emax, op=dist.ReduceOp.MAX)
        elapsed, audio_total, exposed_ms = float(tmax.item()), float(asum.item()), float(emax.item())
    else:
        audio_total = audio_s

    if rank == 0:
        loss_val = float(last_loss.item())
        out = {
            "metric": "audio-hours/sec (base RNN-T training)", "value": audio_total / elapsed / 3600.0,
            "unit": "audio-hours/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": (f"base-85M RNN-T bf16 training, LibriSpeech-960 shapes, batch {args.batch} per GPU "
                                    + ("(BASELINE.json configs[1])" if args.batch == 32 else
                                       "(per-GPU shape of BASELINE.json configs[2] when 128)")) if args.model == "base" else
                                   "large-196M RNN-T bf16 training + on-GPU SpecAugment (BASELINE.json configs[3] shapes)",
                       "global_batch": args.batch * world, "utterance_seconds": "clip(N(12.3,3.8),1,16.7)",
                       "parallelism": f"dp{world}", "final_loss": loss_val,
                       "audio_seconds_per_step": audio_total / args.steps,
                       "timed_step": ("log-mel frontend + mel normalisation (from 16 kHz PCM resident in HBM) + on-device SpecAugment + "
                                      "frame splicing + fwd + loss + bwd + (gradient all-reduce) + LAMB/EMA"
                                      if frontend is not None else
                                      "on-device SpecAugment + frame splicing + fwd + loss + bwd + (gradient all-reduce) + "
                                      "LAMB/EMA; --features-resident: inputs are log-mel features [B,80,T] resident in HBM, the "
                                      "log-mel frontend is outside the timed step")},
        }
        # how close the step is to being launch-bound: the host thread needs `issue` ms to queue a step's commands and
        # waits `wait_for_device` ms per step for the device to catch up (run-ahead limit of two steps)
        out["host_ms_per_step"] = {"issue": round(host_issue / args.steps * 1e3, 2),
                                   "wait_for_device": round(host_wait / args.steps * 1e3, 2)}
        # device-clock time of every timed step: between the ends of consecutive steps (the first: from the start of the loop)
        out["host_issue_ms"] = issue_ms
        out["allocator_segments_added_in_timed_loop"] = int(torch.cuda.memory_stats(dev).get("segment.all.allocated", 0) - segs0)
        out["step_ms_device"] = [round(step_ends[j - 1].elapsed_time(step_ends[j]), 2) for j in range(1, len(step_ends))]
        # ---- the whole step against the chip (SURVEY section 8(d) formulas; training = 3 x forward for the GEMM terms):
        # encoder 2.748 GFLOP per audio-second, prediction + joint_pred 9.175 MFLOP per token (+ SOS), joint 13.369 MFLOP per
        # lattice cell (large-196M: 6.128 / 20.45 / 35.65).  Algorithmic HBM bytes: the logits cross HBM six times per cell
        # (joint_fc write, LSE read, loss-backward read + write, dX read, dW read: 6 V s) unless a pass is fused away, the
        # optimiser moves 14 x 4 P bytes (three passes over five fp32 arenas); LSTM activations are 2 % of that and left out.
        gf_audio, mf_tok, mf_cell = (2.748, 9.175, 13.369) if args.model == "base" else (6.128, 20.45, 35.65)
        flop_step = 3.0 * (gf_audio * 1e9 * audio_s + mf_tok * 1e6 * pred_tokens + mf_cell * 1e6 * cells) / args.steps
        bytes_step = (6.0 * N_CLASSES * 2 * cells / args.steps) + 14 * 4 * int(optimizer.flat_g.numel())
        step_s = elapsed / args.steps
        out["roofline_step"] = {"flop_per_step": flop_step, "tflops": flop_step / step_s / 1e12, "peak_tflops": MFMA_PEAK_TFLOPS,
                                "mfma_frac": flop_step / step_s / 1e12 / MFMA_PEAK_TFLOPS,
                                "hbm_bytes_per_step": bytes_step, "hbm_gbs": bytes_step / step_s / 1e9, "peak_gbs": HBM_PEAK_GBS,
                                "hbm_frac": bytes_step / step_s / 1e9 / HBM_PEAK_GBS,
                                "per_gpu": True,
                                "note": "algorithmic FLOP and HBM bytes of ONE rank's step (SURVEY 8(d) per-unit figures x the "
                                        "audio-seconds, tokens and lattice cells of the timed steps) over the measured step time; the "
                                        "step is a blend of MFMA-bound GEMMs, HBM-bound loss / optimiser passes and a latency-bound "
                                        "recurrence, so neither fraction can reach 1"}
        if world > 1:
            out["allreduce_exposed_ms"] = exposed_ms    # per step, max over ranks: compute stream idle in reducer.finish()
            # model beside the measurement: ring all-reduce of 4 P bytes over one xGMI link per direction (153 GB/s), of which
            # the weight-gradient tail of the backward pass (the only span in which collectives share the chip with compute:
            # fence_collectives keeps them off while a recurrence runs) hides `tail_ms`
            ring_ms = 2.0 * (world - 1) / world * int(optimizer.flat_g.numel()) * 4 / 153e9 * 1e3
            tail_ms = 3.8 if args.model == "base" else 9.7     # joint_fc weight gradient, profiles/r04*_summary.md
            out["allreduce_exposed_ms_model"] = {"ring_ms": ring_ms, "hidden_under_weight_gradient_tail_ms": tail_ms,
                                                 "exposed_ms": max(0.0, ring_ms - tail_ms), "link_gbs": 153.0}
            out["gradient_exchange"] = {"bytes_per_step": int(optimizer.flat_g.numel()) * 4, "collectives_per_step": len(reducer.buckets),
                                        "backend": dist.get_backend()}
        if rehearsal:
            out["rehearsal"] = rehearsal
        if not args.no_kernel_timing:
            summ = _lib.timing_summary()
            out["kernel_ms_per_step"] = {k: round(v[1] / args.steps, 3) for k, v in summ.items()}
            # Roofline of the dominant hand-written kernel: the backward LSTM recurrence.  With the weight-resident
            # chunk kernel (csrc/lstm.hip, default) one launch = all timesteps of one pipeline tick for every active
            # layer (up to 8): each layer reads its recurrent matrix ONCE per launch plus the per-timestep operands
            # (rnnt_ext/cuda/lstm.py::_step_bytes minus the weights).  With per-timestep launches (mode 0, or shapes
            # the resident kernel does not take) one launch = one timestep and the weights are re-read every launch.
            # Either way the kernel is a chain of dependent timesteps: `us_per_timestep` is the figure that moves.
            if "lstm_bwd" in summ:
                brackets, ms, launches, nbytes, tsteps = summ["lstm_bwd"]
                resident = launches < 1.5 * brackets
                lib_ = _lib.lib()
                split_prev = lib_.caiman_lstm_resident_bwd_split(1)
                split_on = bool(split_prev)
                lib_.caiman_lstm_resident_bwd_split(split_prev)
                kname = (("lstm_bwd_resident2_bt" if args.batch > 32 else "lstm_bwd_resident2") if split_on else "lstm_bwd_resident") \
                    if resident else "lstm_bwd_step_mfma"
                achieved = nbytes / (ms * 1e-3) / 1e9
                # The resident kernel is a chain of dependent timesteps, not a stream: what bounds it is the time per
                # timestep (hand-off between the workgroups of a layer + gather + MFMA), so the headline fraction is the
                # MFMA time of one layer-timestep on the layer's own CUs (8*B*H^2 FLOP on H/32 CUs at 2.5 PF/s / 256)
                # over the measured time per dependent timestep.  The HBM pricing of the algorithmic bytes stays beside
                # it (`hbm`): far below peak BECAUSE the kernel waits, not because it streams badly -- XCD-local
                # placement cut its HBM-side traffic by 30 MB per launch at unchanged time.
                us_ts = ms * 1e3 / tsteps
                H_enc = rnnt_cfg["enc_n_hid"]
                floor_us = 8.0 * args.batch * H_enc * H_enc / ((H_enc / 32) * (MFMA_PEAK_TFLOPS * 1e12 / 256)) * 1e6
                hbm = {"achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                       "algorithmic_bytes_per_launch": nbytes / launches}
                if resident:
                    out["roofline"] = {"kernel": kname + " (all backward timesteps of one pipeline tick, every active LSTM layer, per launch)",
                                       "bound": "latency-chain", "achieved": 1.0 / us_ts, "peak": 1.0 / floor_us,
                                       "unit": "dependent timesteps per us (peak = the layer's own CUs doing nothing but the timestep's MFMAs)",
                                       "frac": floor_us / us_ts, "traffic": None, "hbm": hbm,
                                       "avg_launch_us": ms * 1e3 / launches, "launches": launches,
                                       "algorithmic_bytes_per_launch": nbytes / launches,
                                       "chain": {"us_per_timestep": us_ts, "mfma_floor_us_per_timestep": floor_us,
                                                 "frac_of_floor": floor_us / us_ts},
                                       "note": ("live HIP-event measurement; avg_launch_us is event-to-event over sampled brackets "
                                                "(kernel + the events' own cost).  The kernel is bounded by its chain of dependent "
                                                "timesteps (hand-off of the dG row between the workgroups of a layer), not by HBM and "
                                                "not by the matrix cores: `frac` prices the chain, `hbm` the algorithmic bytes; "
                                                "DESIGN.md section 4")}
                else:
                    out["roofline"] = {"kernel": kname + " (one backward timestep of all pipelined LSTM layers per launch)",
                                       "bound": "hbm", **hbm, "traffic": None,
                                       "avg_launch_us": ms * 1e3 / launches, "launches": launches,
                                       "chain": {"us_per_timestep": us_ts, "mfma_floor_us_per_timestep": floor_us,
                                                 "frac_of_floor": floor_us / us_ts},
                                       "note": "live HIP-event measurement; the per-timestep kernels re-stream the recurrent weights every launch"}
                try:   # builder-run rocprofv3 figures of the same command, committed under profiles/: constants on this box
                    pmc = json.load(open(os.path.join(ROOT, "profiles", PMC_FILE)))
                    out["roofline"]["from_profiles"] = {"source": f"profiles/{PMC_FILE} (builder-run rocprofv3 --pmc / --kernel-trace "
                                                                  "passes, NOT measured in this run)",
                                                        "traffic_bytes_per_launch": pmc[kname]["traffic_bytes_per_launch"],
                                                        "rocprof_kernel_avg_us": pmc[kname].get("kernel_avg_us", pmc[kname].get("rocprof_kernel_avg_us")),
                                                        "same_launch_population": pmc[kname].get("same_population")}
                except Exception:
                    pass
            # the three products of the joint projection (hand-written MFMA GEMMs, csrc/joint_gemm.hip / joint_wgrad.hip: the
            # largest kernels of the step by time): 2 M N K FLOP each over the live event-to-event time of their launches
            # (forward: GEMM + LSE epilogue + the partials reduction; weight gradient: kernel + slab sum + remainder rows)
            jg = {k: summ[k] for k in ("joint_gemm_fwd", "joint_gemm_dx", "joint_gemm_dw") if k in summ}
            if jg:
                flop, ms = sum(v[3] for v in jg.values()), sum(v[1] for v in jg.values())
                out["roofline_joint_gemm"] = {
                    "kernel": "joint_fc_gemm8_kernel (forward + row LSE, input gradient) + joint_wgrad8_kernel (weight gradient)",
                    "bound": "mfma", "achieved": flop / (ms * 1e-3) / 1e12, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": flop / (ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, "traffic": None,
                    "per_product": {k[11:]: {"ms_per_step": v[1] / args.steps, "tflops": v[3] / (v[1] * 1e-3) / 1e12}
                                    for k, v in jg.items()},
                    "note": "live HIP events around each call; rocprofv3 counterparts in profiles/r04_base_summary.md "
                            "(64.3 % / 69.4 % SQ_VALU_MFMA_BUSY)"}
            # The `roofline` record is the DOMINANT kernel's.  Since the hand-written projection GEMM became the default that
            # is joint_fc_gemm8_kernel (two launches per step: forward + row LSE, input gradient; 22.8 % of the step's kernel
            # time in profiles/r04_base_summary.md against 17.2 % for the backward recurrence), MFMA-bound; the recurrence's
            # record, the dominant one of rounds 1-3, stays beside it as `roofline_lstm_bwd`.
            jk = {k: summ[k] for k in ("joint_gemm_fwd", "joint_gemm_dx") if k in summ}
            if len(jk) == 2:
                if "roofline" in out:
                    out["roofline_lstm_bwd"] = out.pop("roofline")
                flop, ms = sum(v[3] for v in jk.values()), sum(v[1] for v in jk.values())
                n_l = sum(v[0] for v in jk.values())
                tfl = flop / (ms * 1e-3) / 1e12
                rec = {"kernel": "joint_fc_gemm8_kernel (C = A . W^T + bias on 256 x 256 tiles, 8-phase MFMA main loop, persistent "
                                 "workgroups; per step one launch with the row log-sum-exp epilogue and one for the input gradient)",
                       "bound": "mfma", "achieved": tfl, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tfl / MFMA_PEAK_TFLOPS,
                       "traffic": None, "avg_launch_us": ms * 1e3 / n_l, "launches": n_l,
                       "algorithmic_flop_per_launch": flop / n_l,
                       "note": ("live HIP events around each call (the forward bracket also holds the 0.17 ms reduction of the LSE "
                                "partials); achieved = 2 M N K of the timed launches / their time; peak = dense bf16 MFMA at the "
                                "guide's 2.5 PFLOP/s")}
                try:
                    pmc = json.load(open(os.path.join(ROOT, "profiles", PMC_FILE)))["joint_fc_gemm8_kernel"]
                    rec["traffic"] = pmc["traffic_bytes_per_launch"]
                    rec["from_profiles"] = {"source": f"profiles/{PMC_FILE} (builder-run rocprofv3 --pmc / --kernel-trace passes, NOT "
                                                      "measured in this run; its launches are those of the profile command's "
                                                      "steps, whose batches differ from the timed loop's mix: compare rates, not "
                                                      "durations)",
                                            "traffic_bytes_per_launch": pmc["traffic_bytes_per_launch"],
                                            "rocprof_kernel_avg_us": pmc.get("kernel_avg_us"),
                                            "mfma_busy_frac": pmc.get("mfma_util"),
                                            "hbm_side_rate_gbs": pmc.get("hbm_side_rate_gbs"),
                                            "same_launch_population": pmc.get("same_population")}
                except Exception:
                    pass
                out["roofline"] = rec
            if "loss_bwd" in summ:
                n_launch, ms = summ["loss_bwd"][0], summ["loss_bwd"][1]
                alg = cells * N_CLASSES * 2 * 2  # V*s read + V*s write per lattice cell
                out["roofline_loss_bwd"] = {"kernel": "loss_bwd_colsum_kernel (+ row descriptors + partial-sum reduction: fused joint_fc bias gradient)", "bound": "hbm",
                                            "achieved": alg / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                            "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                            "avg_launch_ms": ms / n_launch, "launches": n_launch}
        out["steps_executed_in_process"] = executed[0]
        native = _lib.lib()
        out["lstm_resident"] = {"launches": int(native.caiman_lstm_resident_launches()),
                                "handoff_timeouts": int(native.caiman_lstm_resident_failures())}   # must be 0
        if world == 1 and args.launch_sources > 0:
            from torch.profiler import ProfilerActivity, profile

            with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True,
                         experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
                for i in range(args.launch_sources):
                    step(args.warmup + i, args.warmup + i)
                torch.cuda.synchronize()
            # torch operators grouped by their Python call stack: launches and device time per step of every (operator, innermost
            # frame of this repository) pair
            n, agg = args.launch_sources, {}
            for e in prof.key_averages(group_by_stack_n=30):
                dev_us = getattr(e, "device_time_total", 0.0) or getattr(e, "cuda_time_total", 0.0)
                if dev_us <= 0 or not e.key.startswith("aten::"):
                    continue
                # the first three frames of this repository in the order the profiler lists them (innermost first)
                mine = [f for f in e.stack if ("caiman_asr_amd" in f or "bench.py" in f) and "_lib.py" not in f][:3]
                src = " < ".join(f.split("caiman_asr_amd/")[-1].split("repo/")[-1] for f in mine) or "?"
                a = agg.setdefault((e.key, src[:150]), [0, 0.0])
                a[0] += e.count
                a[1] += e.self_device_time_total if hasattr(e, "self_device_time_total") else dev_us
            rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
            for (op, src), (cnt, us) in rows[:110]:
                print(f"[launch sources] {us / n:8.1f} us {cnt / n:6.1f} x  {op:34s} {src}", file=sys.stderr)
        if world == 1 and args.feed:
            try:
                log("feed beside the step")
                out["feed"] = feed_beside_step(step, args, dev, args.warmup)
            except Exception as e:
                out["feed"] = {"error": repr(e)}
        if world == 1 and not args.no_decode and args.model == "base":
            try:
                log("decode record (child process)")
                out["decode"] = decode_record()
            except Exception as e:
                out["decode"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                # the box exposes more logical CPUs than this job's share (16 per GPU): oversubscribing
                # torch's intra-op pool makes the CPU leg crawl
                cores = min(16, len(os.sched_getaffinity(0)))
                log(f"cpu_baseline on {cores} threads")
                out["cpu_baseline"] = cpu_baseline(model, threads=cores)
            except Exception as e:  # the baseline must never take the GPU number down with it
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main())
