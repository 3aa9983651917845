"""Generate golden vectors from the REFERENCE's own Python (CPU) -> tests/golden/*.npz|json.

TEST INFRASTRUCTURE ONLY.  Runs in the authoring container only (it imports
/root/reference, which does not exist on the GPU box); the outputs are small data files
(inputs + expected outputs + random-init weights), never source text.

The reference imports packages this image lacks (beartype, jaxtyping, apex, kenlm, cerberus).
They are type-annotation / third-party-op shims irrelevant to the CPU arithmetic exercised here,
so they are replaced by in-memory no-op stand-ins (nothing is written anywhere):
  beartype.beartype / jaxtyping.jaxtyped -> identity decorators,
  apex TransducerJoint -> unused placeholder (joint_apex_transducer=None selects the reference's
  own `torch_transducer_joint`),
  rnnt_ext.cuda.{logsumexp,transducer_loss} (CUDA-only binaries) -> empty modules, never called.

Usage:  python oracle/gen_golden.py
"""
import itertools
import json
import os
import sys
import types
import typing

sys.dont_write_bytecode = True
REF = "/root/reference/training"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _install_stubs():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    def ident(*a, **k):
        if len(a) == 1 and not k:  # @beartype on a function, class, classmethod or staticmethod object
            return a[0]
        return lambda f: f

    def chunked(iterable, n):   # more_itertools.chunked (absent here): consecutive lists of n items, the last one shorter
        it = iter(iterable)
        while True:
            chunk = list(itertools.islice(it, n))
            if not chunk:
                return
            yield chunk

    stub("more_itertools", chunked=chunked)
    stub("orjson", loads=json.loads, dumps=lambda o, option=None: json.dumps(o).encode(), OPT_INDENT_2=0)  # absent; std json reads the same
    stub("sox")   # only used by the reference to build a manifest from a directory of audio files
    stub("beartype", beartype=ident)
    btt = stub("beartype.typing")
    btt.__dict__.update({k: getattr(typing, k) for k in dir(typing) if not k.startswith("_")})

    class _Sub:
        def __class_getitem__(cls, item):
            return typing.Any

    stub("jaxtyping", jaxtyped=lambda **k: (lambda f: f), Int=_Sub, Float=_Sub, Bool=_Sub, Shaped=_Sub)
    stub("apex")
    stub("apex.contrib")

    class TransducerJoint:  # never called: joint_apex_transducer=None below
        def __init__(self, *a, **k):
            pass

    stub("apex.contrib.transducer", TransducerJoint=TransducerJoint)
    class _Empty:  # placeholder type for annotations / unused validators
        def __init__(self, *a, **k):
            pass

    stub("kenlm", State=_Empty, Model=_Empty, LanguageModel=_Empty)  # n-gram rescoring is not exercised (ngram_info=None)
    class _AcceptAll(_Empty):  # the keyword lists written below are well-formed by construction
        errors = {}

        def validate(self, *a, **k):
            return True

    stub("cerberus", Validator=_AcceptAll)                           # keyword-list schema check
    # the compiled CUDA extensions cannot exist here; only pure-Python helpers of the modules
    # that import them (get_packing_meta_data) are used below.
    sys.path.insert(0, os.path.join(REF, "lib", "src"))
    import rnnt_ext.cuda  # noqa: F401  (the reference's empty package)

    stub("rnnt_ext.cuda.logsumexp")
    stub("rnnt_ext.cuda.transducer_loss")
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "lib", "src"))


def main():
    _install_stubs()
    import numpy as np
    import torch
    import yaml

    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(4)

    from caiman_asr_train.data.features import stack_subsample_frames
    from caiman_asr_train.rnnt.batched_greedy import RNNTBatchedGreedyDecoder
    from caiman_asr_train.rnnt.loss import get_packing_meta_data
    from caiman_asr_train.rnnt.model import RNNT, StackTime
    from caiman_asr_train.train_utils.lr import lr_policy

    # ---- 1. state_dict schema of the shipped configs (shapes only) -----------------
    for name in ("base", "large"):
        schema = json.load(open(f"{REF}/caiman_asr_train/export/model_schema/{name}.json"))
        json.dump(schema, open(os.path.join(OUT, f"schema_{name}.json"), "w"), indent=0)
    for name, n_classes in (("base-8703sp", 8704), ("large-17407sp", 17408)):
        cfg = yaml.safe_load(open(f"{REF}/configs/{name}.yaml"))["rnnt"]
        json.dump(cfg, open(os.path.join(OUT, f"rnnt_cfg_{name.split('-')[0]}.json"), "w"), indent=0)

    # ---- 2. mini models: weights + forward outputs --------------------------------------
    def mini_cfg(enc, pred, joint, in_feats):
        return dict(in_feats=in_feats, enc_n_hid=enc, enc_batch_norm=False, pred_batch_norm=False,
                    enc_pre_rnn_layers=2, enc_post_rnn_layers=3, enc_stack_time_factor=2, enc_dropout=0.0,
                    pred_dropout=0.0, joint_dropout=0.0, pred_n_hid=pred, pred_rnn_layers=2, joint_n_hid=joint,
                    forget_gate_bias=1.0, custom_lstm=False, weights_init_scale=0.5,
                    hidden_hidden_bias_scale=0.0, joint_net_lr_factor=0.343)

    class _Tok:  # detokeniser stand-in: the decoder only calls sentpiece.id_to_piece
        class sentpiece:
            @staticmethod
            def id_to_piece(i):
                return f"<{i}>"

    for tag, (enc, pred, joint, in_feats, V) in {"tiny": (11, 12, 13, 6, 30), "mfma": (64, 64, 32, 24, 30)}.items():
        torch.manual_seed(1234)
        cfg = mini_cfg(enc, pred, joint, in_feats)
        m = RNNT(n_classes=V, **cfg).eval()
        # random-init weights give a degenerate greedy path (one token repeated); widen them so
        # blank / non-blank decisions and token identities vary across frames and utterances.
        with torch.no_grad():
            for p_ in m.parameters():
                p_.normal_(0.0, 0.35)
            m.joint_net[2].bias[V - 1] += (0.6 if tag == "tiny" else 2.0)
        T, B, U = 23, 3, 7
        x = torch.randn(T, B, in_feats)
        x_lens = torch.tensor([23, 17, 9])
        y = torch.randint(0, V - 1, (B, U))
        y_lens = torch.tensor([7, 4, 2])
        with torch.no_grad():
            f, f_lens, _ = m.encode(x, x_lens)
            g, _, _ = m.predict(y)
            logits, out_lens, _ = m(x, x_lens, y, y_lens)
        # state passing: second half given the state of the first half (custom LSTM path is
        # what produces states; with torch.nn.LSTM we emulate through encode on slices)
        sd = {k: v.numpy() for k, v in m.state_dict().items()}
        meta = get_packing_meta_data(x_lens, y_lens, 2)
        # greedy decode through the reference decoder (token ids / frame indices are exact)
        dec = RNNTBatchedGreedyDecoder(model=m, blank_idx=V - 1, eos_strategy=None, max_inputs_per_batch=int(1e7),
                                       tokenizer=_Tok(), max_symbols_per_step=3)
        res = dec.decode(x, x_lens)
        toks, frames, confs = [], [], []
        for per_utt in res:
            tk, fr, cf = [], [], []
            for t in sorted(per_utt):
                hyp = per_utt[t].final.alternatives[0]
                tk += hyp.y_seq
                fr += hyp.timesteps
                cf += hyp.confidence
            toks.append(tk)
            frames.append(fr)
            confs.append(cf)
        np.savez_compressed(
            os.path.join(OUT, f"rnnt_{tag}.npz"), x=x.numpy(), x_lens=x_lens.numpy(), y=y.numpy(),
            y_lens=y_lens.numpy(), f=f.numpy(), f_lens=f_lens.numpy(), g=g.numpy(), logits=logits.numpy(),
            batch_offset=meta["batch_offset"].numpy(), max_f_len=np.array(meta["max_f_len"]),
            cfg=json.dumps(cfg), n_classes=np.array(V),
            greedy=json.dumps(dict(tokens=toks, frames=frames, confidence=confs)),
            **{"sd." + k: v for k, v in sd.items()})
        print(tag, "logits", tuple(logits.shape), "greedy", toks)

    # ---- 2b. beam search (width 4, temperature 1.4, reference defaults) on the "mfma" mini model -------------
    # The reference decoder asserts that <unk> (id 0) is never emitted, so its logit is pushed far down; the
    # sentencepiece pieces it uses for hypothesis merging are stored so that no .model file has to travel.
    from sentencepiece import SentencePieceProcessor as SPP

    from caiman_asr_train.rnnt.beam import RNNTBeamDecoder

    spm = f"{REF}/tests/test_data/librispeech29.model"
    pieces = [SPP(model_file=spm).id_to_piece(i) for i in range(29)]
    g = np.load(os.path.join(OUT, "rnnt_mfma.npz"))
    cfg = json.loads(str(g["cfg"]))
    V = int(g["n_classes"])
    m = RNNT(n_classes=V, **cfg).eval()
    sd = {k[3:]: torch.tensor(g[k]) for k in g.files if k.startswith("sd.")}
    sd["joint_net.2.bias"][0] = -30.0
    m.load_state_dict(sd)
    from caiman_asr_train.rnnt.eos_strategy import EOSBlank, EOSIgnore, EOSPredict

    beam_cases = {
        "default": dict(max_symbols_per_step=8),
        "capped": dict(max_symbols_per_step=2, beam_width=3),
        "wide": dict(beam_width=8, beam_prune_score_thresh=-1, beam_prune_topk_thresh=-1, max_symbols_per_step=4),
        "partials": dict(return_partials=True),
        "forced_finals": dict(final_emission_thresh=0.06, frame_width=0.06),
        "vad": dict(eos_vad_threshold=0.12, frame_width=0.06, beam_width=2, temperature=0.7),
        "sample_cap": dict(max_symbol_per_sample=3),
        "keywords": dict(keywords={"ra": 1.5, "r e": 0.8, "nn": -2.0}),
        "eos_terminal": dict(eos=["predict", 18, 1.0, 0.0], eos_is_terminal=True),
        "eos_blank": dict(eos=["blank", 18]),
        "eos_ignore": dict(eos=["ignore", 12]),
        "eos_predict_beta": dict(eos=["predict", 18, 0.8, 0.3]),
    }
    beam_out = {}
    for tag, kw in beam_cases.items():
        args = dict(kw)
        eos = args.pop("eos", None)
        strategy = None if eos is None else {"predict": EOSPredict, "blank": EOSBlank, "ignore": EOSIgnore}[eos[0]](*eos[1:])
        if "keywords" in args:
            kpath = os.path.join(OUT, "_kw_tmp.json")
            json.dump({"keywords": args.pop("keywords")}, open(kpath, "w"))
            args["keyword_boost_path"] = kpath
        dec = RNNTBeamDecoder(model=m, blank_idx=V - 1, eos_strategy=strategy, sentpiece_model=spm, **args)
        res = dec.decode(torch.tensor(g["x"]), torch.tensor(g["x_lens"]))
        if "keyword_boost_path" in args:
            os.remove(args["keyword_boost_path"])
        per_utt = []
        for r in res:
            tk, ts, cf, frames, parts = [], [], [], [], {}
            for t in sorted(r):
                if r[t].final is not None:
                    a = r[t].final.alternatives[0]
                    tk += a.y_seq
                    ts += a.timesteps
                    cf += a.confidence
                    frames.append(t)
                if r[t].partials is not None:
                    parts[t] = dict(start=r[t].partials.start_frame_idx, dur=r[t].partials.duration_frames,
                                    alts=[[h.y_seq, h.timesteps] for h in r[t].partials.alternatives])
            per_utt.append(dict(tokens=tk, timesteps=ts, confidence=cf, final_frames=frames, last_key=max(r),
                                partials=parts))
        beam_out[tag] = dict(kwargs=kw, utts=per_utt)
        print("beam", tag, [len(u["tokens"]) for u in per_utt], [u["last_key"] for u in per_utt])
    json.dump(dict(pieces=pieces, unk_bias=-30.0, results=beam_out), open(os.path.join(OUT, "beam_mfma.json"), "w"))

    # ---- 2c. inference-only config filter + the reference's own hardware-checkpoint fixture ----------------------
    from caiman_asr_train.export.config_schema import RNNTInferenceConfigSchema
    import yaml   # the reference's rnnt/config.py pulls in DALI; its load() is these two yaml calls (config.py:44-48)

    class _NoAlias(yaml.SafeDumper):
        def ignore_aliases(self, data):
            return True

    inf = {}
    for name in ("base-8703sp", "large-17407sp", "testing-1023sp", "testing-1023sp_run", "librispeech"):
        full = yaml.safe_load(yaml.dump(yaml.safe_load(open(f"{REF}/configs/{name}.yaml")), Dumper=_NoAlias))
        try:
            inf[name] = dict(full=full, inference=RNNTInferenceConfigSchema(**full).model_dump())
        except Exception as e:  # a config the reference itself rejects is a fixture too
            inf[name] = dict(full=full, error=type(e).__name__)
        print("inference config", name, "error" if "error" in inf[name] else "ok")
    hw = torch.load(f"{REF}/tests/test_data/hardware_ckpt.pt", map_location="cpu", weights_only=False)
    hw_meta = dict(keys=sorted(hw), rnnt_config=hw["rnnt_config"], melalpha=hw["melalpha"], epoch=hw["epoch"], step=hw["step"],
                   best_wer=hw["best_wer"], ngram_keys=sorted(hw["ngram"]), ngram_binary_is_bytes=isinstance(hw["ngram"]["binary"], bytes),
                   sentpiece_is_bytes=isinstance(hw["sentpiece_model"], bytes), mel_shape=list(hw["melmeans"].shape),
                   mel_dtype=str(hw["melmeans"].dtype), state_dict={k: list(v.shape) for k, v in hw["state_dict"].items()})
    json.dump(dict(configs=inf, hardware_ckpt=hw_meta), open(os.path.join(OUT, "export.json"), "w"))

    # ---- 2d. samplers: utterance order per rank for seeded synthetic manifests -----------------------------------
    from caiman_asr_train.data.dali import manifest_ratios as mr
    from caiman_asr_train.data.dali import sampler as ref_sampler

    srng = np.random.default_rng(11)
    manifests, label = [], 0
    for m_i, n_utt in enumerate((96, 160, 64)):
        man = {}
        for j in range(n_utt):
            man[f"m{m_i}/utt{j:04d}.flac"] = {"label": label, "duration": float(np.round(srng.uniform(1.0, 16.7), 2))}
            label += 1
        manifests.append(man)
    names = ["m0.json", "m1.json", "m2.json"]
    sampler_cases = {
        "simple_w1": ("SimpleSampler", dict(total_batches=None, batch_size=8, global_batch_size=None, world_size=1), None, None),
        "simple_w4": ("SimpleSampler", dict(total_batches=None, batch_size=8, global_batch_size=None, world_size=4), None, None),
        "sorted_w1": ("SortedSampler", dict(total_batches=None, batch_size=8, global_batch_size=None, world_size=1), None, None),
        "sorted_w4": ("SortedSampler", dict(total_batches=None, batch_size=8, global_batch_size=None, world_size=4), None, None),
        "random_w2": ("RandomSampler", dict(total_batches=200, batch_size=4, global_batch_size=16, world_size=2, resume_step=0), 5, None),
        "bucket_w1": ("BucketingSampler", dict(total_batches=90, batch_size=8, global_batch_size=32, world_size=1, resume_step=0, num_buckets=6), 7, None),
        "bucket_w4": ("BucketingSampler", dict(total_batches=200, batch_size=4, global_batch_size=32, world_size=4, resume_step=0, num_buckets=6), 7, None),
        "bucket_w4_resume": ("BucketingSampler", dict(total_batches=200, batch_size=4, global_batch_size=32, world_size=4, resume_step=3, num_buckets=6), 7, None),
        "bucket_one_epoch_w4": ("BucketingSampler", dict(total_batches=40, batch_size=4, global_batch_size=32, world_size=4, resume_step=0, num_buckets=6), 7, None),
        "bucket_nopess": ("BucketingSampler", dict(total_batches=100, batch_size=8, global_batch_size=16, world_size=2, resume_step=0, num_buckets=3,
                                                         pessimistic_first_batch=False), 9, None),
        "bucket_pess_rand1": ("BucketingSampler", dict(total_batches=100, batch_size=8, global_batch_size=16, world_size=2, resume_step=0, num_buckets=3,
                                                       randomize_n_epochs=1), 9, None),
        "bucket_relative": ("BucketingSampler", dict(total_batches=150, batch_size=4, global_batch_size=16, world_size=2, resume_step=0, num_buckets=4), 3,
                            ("relative", [1.0, 0.5, 2.0])),
        "bucket_absolute": ("BucketingSampler", dict(total_batches=150, batch_size=4, global_batch_size=16, world_size=2, resume_step=0, num_buckets=4), 3,
                            ("absolute", [1.0, 1.0, 1.0])),
        "bucket_canary": ("BucketingSampler", dict(total_batches=150, batch_size=4, global_batch_size=16, world_size=2, resume_step=0, num_buckets=4), 3,
                          ("canary", 0.5)),
    }
    sampler_out = {}
    for tag, (klass, kw, seed, ratios) in sampler_cases.items():
        kw2 = dict(kw)
        if seed is not None:
            kw2["rng"] = np.random.default_rng(seed)
        smp = getattr(ref_sampler, klass)(**kw2)
        ratio_obj = None
        if ratios is not None:
            ratio_obj = {"relative": mr.RelativeManifestRatios, "absolute": mr.AbsoluteManifestRatios,
                         "canary": mr.CanaryManifestRatios}[ratios[0]](ratios[1])
        files, epoch_size = smp.process_output_files([dict(m) for m in manifests], names, ratio_obj)
        sampler_out[tag] = dict(klass=klass, kwargs=kw, seed=seed, ratios=ratios, epoch_size=epoch_size,
                                labels=[u.label for u in files])
        print("sampler", tag, len(files), epoch_size)
    json.dump(dict(manifests=manifests, names=names, cases=sampler_out), open(os.path.join(OUT, "sampler.json"), "w"))

    # ---- 2e. manifest parsing / filtering -------------------------------------------------------------------------
    from caiman_asr_train.data.dali import utils as ref_dutils

    mrng = np.random.default_rng(4)
    entries = []
    for i in range(14):
        n_words = int(mrng.integers(1, 12))
        entries.append({"transcript": " ".join("w%d" % int(mrng.integers(0, 50)) for _ in range(n_words)),
                        "files": [{"fname": f"slow/utt{i}.flac"}, {"fname": f"clips/utt{i}.flac"}],
                        "original_duration": float(np.round(mrng.uniform(0.02, 20.0), 2))})
    man_path = os.path.join(OUT, "_manifest_tmp.json")
    json.dump(entries, open(man_path, "w"))
    pred = ref_dutils.set_predicate(16.7, 40)
    files_all, tr_all = ref_dutils._parse_json(man_path)
    files_f, tr_f = ref_dutils._parse_json(man_path, 7, pred)
    sub_files, sub_tr = ref_dutils._filter_files(dict(files_all), dict(tr_all), 5, 3)
    os.remove(man_path)
    json.dump(dict(entries=entries, parsed=dict(files=files_all, transcripts={str(k): v for k, v in tr_all.items()}),
                   filtered=dict(start_label=7, max_duration=16.7, max_transcript_len=40, files=files_f,
                                 transcripts={str(k): v for k, v in tr_f.items()}),
                   subset=dict(n=5, seed=3, files=sub_files, order=list(sub_files), transcripts={str(k): v for k, v in sub_tr.items()})),
              open(os.path.join(OUT, "manifest.json"), "w"))
    print("manifest", len(files_all), len(files_f), len(sub_files))

    # ---- 3. small pure functions ---------------------------------------------------------
    torch.manual_seed(7)
    x = torch.randn(9, 2, 5)
    lens = torch.tensor([9, 6])
    st = {}
    for fac in (2, 3):
        o, l = StackTime(fac)(x, lens)
        st[f"stack{fac}"] = o.numpy()
        st[f"stack{fac}_lens"] = l.numpy()
    a = torch.randn(2, 4, 10)
    alens = torch.tensor([10, 7])
    for (s, ss) in ((3, 3), (1, 1), (2, 1), (3, 2)):
        o, l = stack_subsample_frames(a, alens, s, ss)
        st[f"splice_{s}_{ss}"] = o.numpy()
        st[f"splice_{s}_{ss}_lens"] = l.numpy()
    np.savez_compressed(os.path.join(OUT, "shape_ops.npz"), x=x.numpy(), lens=lens.numpy(), a=a.numpy(),
                        alens=alens.numpy(), **st)

    class _Opt:
        def __init__(self, n):
            self.param_groups = [dict(lr=0.0) for _ in range(n)]

    rows = []
    init = [4e-3, 4e-3 * 0.343]
    for step in (0, 1, 100, 1631, 1632, 5000, 19631, 19632, 25000, 60000, 200000):
        o = _Opt(2)
        lr_policy(o, init, 4e-4, step, 1632, 18000, 10880)
        rows.append([step] + [g["lr"] for g in o.param_groups])
    json.dump(dict(initial_lr=init, min_lr=4e-4, warmup=1632, hold=18000, half_life=10880, rows=rows),
              open(os.path.join(OUT, "lr_policy.json"), "w"))
    # ---- 4. frontend: the reference's own golden log-mel tensor + the decoded test recording ------
    # training/tests/data/dali/test_data_loader.py:235-258 compares the DALI val pipeline (testing config:
    # window 0.02 s, no initial padding, per-utterance normalisation) on 2 copies of one 8.89 s FLAC with
    # tests/test_data/audio_tensor_batch.pt (atol 2e-4).  Both rows of that tensor are identical; row 0 and
    # the PCM samples (decoded with oracle/flac.py, no audio library in this image) are stored as data.
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import flac

    td = f"{REF}/tests/test_data"
    pcm, sr, bps = flac.decode(f"{td}/gov_DOT_uscourts_DOT_ca9_DOT_04-56618_DOT_2006-02-16_DOT_mp3_00027.flac")
    gold = torch.load(f"{td}/audio_tensor_batch.pt").numpy()
    assert sr == 16000 and bps == 16 and pcm.shape[1] == 1 and np.array_equal(gold[0], gold[1])
    np.savez_compressed(os.path.join(OUT, "frontend_ref.npz"), pcm=pcm[:, 0].astype(np.int16), sample_rate=sr,
                        logmel_norm=gold[0].astype(np.float32), window_size=0.02, window_stride=0.01, n_fft=512)
    # the reference's hardware-checkpoint fixture holds a 1 538-parameter RNN-T written by the reference itself:
    # its state_dict (names / shapes / values) + rnnt config pin the checkpoint schema (data only)
    hw = torch.load(f"{td}/hardware_ckpt.pt", weights_only=True)
    np.savez_compressed(os.path.join(OUT, "ref_ckpt_mini.npz"), rnnt_config=json.dumps(hw["rnnt_config"]["rnnt"]),
                        epoch=hw["epoch"], step=hw["step"], best_wer=hw["best_wer"], version=hw["version"],
                        **{"sd." + k: v.numpy() for k, v in hw["state_dict"].items()})
    for name in ("melmeans", "melvars"):
        np.save(os.path.join(OUT, f"{name}.npy"), torch.load(f"{td}/{name}.pt").numpy())
    print("golden vectors written to", OUT)


if __name__ == "__main__":
    main()
