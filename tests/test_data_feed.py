"""Data feed, CPU: utterance order per rank equals the reference samplers' on seeded synthetic
manifests (tests/golden/sampler.json, oracle/gen_golden.py)."""
import json
import os

import numpy as np
import pytest

from caiman_asr_amd.data import sampler as S

GOLD = os.path.join(os.path.dirname(__file__), "golden")
SAMPLER = json.load(open(os.path.join(GOLD, "sampler.json")))


def _build(case):
    kw = dict(case["kwargs"])
    if case["seed"] is not None:
        kw["rng"] = np.random.default_rng(case["seed"])
    ratios = None
    if case["ratios"] is not None:
        kind, val = case["ratios"]
        ratios = {"relative": S.RelativeManifestRatios, "absolute": S.AbsoluteManifestRatios,
                  "canary": S.CanaryManifestRatios}[kind](val)
    return getattr(S, case["klass"])(**kw), ratios


@pytest.mark.parametrize("tag", sorted(SAMPLER["cases"]))
def test_sampler_order_matches_reference(tag):
    case = SAMPLER["cases"][tag]
    smp, ratios = _build(case)
    files, epoch_size = smp.process_output_files([dict(m) for m in SAMPLER["manifests"]], SAMPLER["names"], ratios)
    assert epoch_size == case["epoch_size"]
    assert [u.label for u in files] == case["labels"]
    # what a rank reads: contiguous shards of whole batches, together the whole list
    smp.make_file_list([dict(m) for m in SAMPLER["manifests"]], SAMPLER["names"], ratios) if case["seed"] is None else None
    W = smp.world_size
    shards = [smp.rank_shard(r, files) for r in range(W)]
    assert [u.label for s in shards for u in s] == case["labels"]
    if len(files) % W == 0:
        assert len({len(s) for s in shards}) == 1


def test_sampler_properties_and_errors():
    man = [dict(m) for m in SAMPLER["manifests"]]
    smp = S.BucketingSampler(total_batches=200, batch_size=4, global_batch_size=32, world_size=4, resume_step=0,
                             rng=np.random.default_rng(0), num_buckets=6)
    smp.make_file_list(man, SAMPLER["names"])
    assert smp.dataset_size == 320 and smp.epoch_size == 320 and len(smp.read_file_list()) == 960
    files = smp._files
    durs = np.array([u.duration for u in files])
    # no file twice within an epoch, for any rank (a rank reads 80 utterances of each epoch)
    for r in range(4):
        shard = smp.rank_shard(r)
        assert len(shard) == 240
        for e in range(3):
            names = [u.file_name for u in shard[80 * e: 80 * (e + 1)]]
            assert len(set(names)) == 80
    # the first step of every rank holds the longest material: worst case first
    first_global = np.concatenate([[u.duration for u in smp.rank_shard(r)[:8]] for r in range(4)])
    assert first_global.max() == durs.max() and first_global.sum() >= np.sort(durs[:320])[-32:].sum() * 0.6
    # buckets: utterances of a batch have similar duration (std far below the corpus std)
    batch_std = np.mean([np.std([u.duration for u in smp.rank_shard(0)[i:i + 4]]) for i in range(80, 240, 4)])
    assert batch_std < 0.5 * durs.std()
    with pytest.raises(AssertionError):
        S.BucketingSampler(total_batches=10, batch_size=4, global_batch_size=6, world_size=1, resume_step=0,
                           rng=np.random.default_rng(0), num_buckets=2)
    with pytest.raises(ValueError, match="randomize_n_epochs"):
        S.BucketingSampler(total_batches=200, batch_size=4, global_batch_size=32, world_size=4, resume_step=0,
                           rng=np.random.default_rng(0), num_buckets=6, randomize_n_epochs=2).make_file_list(man, SAMPLER["names"])
    with pytest.raises(AssertionError, match="smaller than global batch size"):
        S.BucketingSampler(total_batches=10, batch_size=128, global_batch_size=128, world_size=1, resume_step=0,
                           rng=np.random.default_rng(0), num_buckets=2).make_file_list(man, SAMPLER["names"])
    with pytest.raises(ValueError, match="At most one"):
        S.build_manifest_ratios([1.0], [1.0], None)
    assert S.build_manifest_ratios(None, None, None) is None
    assert S.build_json_fracs(S.build_manifest_ratios(None, [2.0, 1.0], None), [10, 20]) == [20.0, 20.0]


# ---- manifests ------------------------------------------------------------------------------------------------------
MANIFEST = json.load(open(os.path.join(GOLD, "manifest.json")))


def test_manifest_parse_and_filter_match_reference(tmp_path):
    from caiman_asr_amd.data.manifest import filter_files, load_manifests, parse_json, set_predicate

    p = tmp_path / "m.json"
    p.write_text(json.dumps(MANIFEST["entries"]))
    files, tr = parse_json(str(p))
    assert files == MANIFEST["parsed"]["files"] and list(files) == list(MANIFEST["parsed"]["files"])
    assert {str(k): v for k, v in tr.items()} == MANIFEST["parsed"]["transcripts"]
    f = MANIFEST["filtered"]
    files_f, tr_f = parse_json(str(p), f["start_label"], set_predicate(f["max_duration"], f["max_transcript_len"]))
    assert files_f == f["files"] and {str(k): v for k, v in tr_f.items()} == f["transcripts"]
    sub = MANIFEST["subset"]
    files_s, tr_s = filter_files(dict(files), dict(tr), sub["n"], sub["seed"])
    assert files_s == sub["files"] and list(files_s) == sub["order"]
    assert {str(k): v for k, v in tr_s.items()} == sub["transcripts"]
    assert filter_files(files, tr, None, 0) == (files, tr)
    # the reference's own two-utterance manifest; several manifests share one label space
    ref_files, ref_tr = parse_json(os.path.join(GOLD, "manifest_short.json"))
    assert [v["label"] for v in ref_files.values()] == [0, 1] and ref_tr[1] == "second clip is repeated"
    many, tr_all = load_manifests([os.path.join(GOLD, "manifest_short.json"), str(p)])
    assert [len(m) for m in many] == [2, 14] and sorted(tr_all) == list(range(16))
    assert min(v["label"] for v in many[1].values()) == 2


# ---- audio decode (host code of the library: runs without a GPU) -----------------------------------------------------------
def _flac_md5(data: bytes) -> str:
    return data[8 + 18: 8 + 34].hex()      # STREAMINFO is the first metadata block; its last 16 bytes


def test_flac_decode_reproduces_the_streams_own_md5():
    """The reference's test recording (libFLAC-encoded, LPC subframes): the decoded PCM hashes to the MD5 the encoder
    stored in STREAMINFO, and equals the samples the independent Python decoder (oracle/flac.py) produced."""
    import hashlib

    from caiman_asr_amd.data.audio import audio_info, decode_audio

    data = open(os.path.join(GOLD, "ref_clip.flac"), "rb").read()
    assert audio_info(data) == (16000, 1, 142240)
    x, sr = decode_audio(data)
    pcm = np.round(x * 32768.0).astype(np.int16)
    assert sr == 16000 and hashlib.md5(pcm.tobytes()).hexdigest() == _flac_md5(data)
    g = np.load(os.path.join(GOLD, "frontend_ref.npz"))
    assert np.array_equal(pcm, g["pcm"].astype(np.int16).reshape(-1))


class _BitWriter:
    def __init__(self):
        self.bits = []

    def put(self, v, n):
        self.bits += [(v >> (n - 1 - i)) & 1 for i in range(n)]

    def signed(self, v, n):
        self.put(v & ((1 << n) - 1), n)

    def unary(self, q):
        self.bits += [0] * q + [1]

    def pad(self):
        self.bits += [0] * (-len(self.bits) % 8)

    def bytes(self):
        assert len(self.bits) % 8 == 0
        return bytes(int("".join(map(str, self.bits[i:i + 8])), 2) for i in range(0, len(self.bits), 8))


def _crc(data, poly, width):
    c, top, mask = 0, 1 << (width - 1), (1 << width) - 1
    for byte in data:
        c ^= byte << (width - 8)
        for _ in range(8):
            c = ((c << 1) ^ poly) & mask if c & top else (c << 1) & mask
    return c


def _encode_flac(channels, bits=16, sr=16000, blocksize=192, mode="fixed2", stereo=None):
    """Tiny FLAC writer for tests: CONSTANT / VERBATIM / FIXED order-2 subframes with Rice coding (k = 4, one
    partition), optional left/side | side/right | mid/side decorrelation."""
    import hashlib

    channels = [np.asarray(c, dtype=np.int64) for c in channels]
    n = len(channels[0])
    inter = np.stack(channels, 1).astype(f"<i{bits // 8}").tobytes() if bits in (16, 32) else b""
    md5 = hashlib.md5(inter).digest() if inter else bytes(16)
    si = _BitWriter()
    si.put(blocksize, 16); si.put(blocksize, 16); si.put(0, 24); si.put(0, 24)
    si.put(sr, 20); si.put(len(channels) - 1, 3); si.put(bits - 1, 5); si.put(n, 36)
    out = b"fLaC" + bytes([0x80]) + (34).to_bytes(3, "big") + si.bytes() + md5

    def subframe(w, s, bps):
        s = [int(v) for v in s]
        if mode == "constant" and len(set(s)) == 1:
            w.put(0, 1); w.put(0, 6); w.put(0, 1); w.signed(s[0], bps)
        elif mode == "verbatim" or len(s) < 3:
            w.put(0, 1); w.put(1, 6); w.put(0, 1)
            for v in s:
                w.signed(v, bps)
        else:
            w.put(0, 1); w.put(8 + 2, 6); w.put(0, 1)
            w.signed(s[0], bps); w.signed(s[1], bps)
            w.put(0, 2); w.put(0, 4); w.put(4, 4)          # Rice, partition order 0, k = 4
            for i in range(2, len(s)):
                r = s[i] - (2 * s[i - 1] - s[i - 2])
                u = (r << 1) ^ (r >> 63)
                w.unary(u >> 4); w.put(u & 15, 4)

    for f, lo in enumerate(range(0, n, blocksize)):
        blk = [c[lo:lo + blocksize] for c in channels]
        m = len(blk[0])
        w = _BitWriter()
        w.put(0x3FFE, 14); w.put(0, 1); w.put(0, 1)
        w.put(1 if m == 192 else 7, 4); w.put(0, 4)
        ch_code = len(channels) - 1 if stereo is None else {"ls": 8, "sr": 9, "ms": 10}[stereo]
        w.put(ch_code, 4); w.put({8: 1, 16: 4, 24: 6}[bits], 3); w.put(0, 1)
        assert f < 128
        w.put(f, 8)
        if m != 192:
            w.put(m - 1, 16)
        hdr = w.bytes()
        w.put(_crc(hdr, 0x07, 8), 8)
        if stereo is None:
            for c in blk:
                subframe(w, c, bits)
        else:
            L, R = blk
            side = L - R
            if stereo == "ls":
                subframe(w, L, bits); subframe(w, side, bits + 1)
            elif stereo == "sr":
                subframe(w, side, bits + 1); subframe(w, R, bits)
            else:
                subframe(w, (L + R) >> 1, bits); subframe(w, side, bits + 1)
        w.pad()
        body = w.bytes()
        out += body + _crc(body, 0x8005, 16).to_bytes(2, "big")
    return out


@pytest.mark.parametrize("mode, stereo, bits", [("fixed2", None, 16), ("verbatim", None, 16), ("constant", None, 16),
                                                ("fixed2", "ls", 16), ("fixed2", "sr", 16), ("fixed2", "ms", 16),
                                                ("fixed2", None, 24), ("verbatim", "ms", 8)])
def test_flac_decode_synthetic_streams(mode, stereo, bits):
    from caiman_asr_amd.data.audio import audio_info, decode_audio

    rng = np.random.default_rng(5)
    n, amp = 500, (1 << (bits - 1)) // 3
    t = np.arange(n)
    L = (amp * np.sin(t / 9.0) + rng.integers(-20, 20, n)).astype(np.int64)
    R = (amp * np.cos(t / 13.0) + rng.integers(-20, 20, n)).astype(np.int64)
    if mode == "constant":
        L[:] = 1234 % (1 << (bits - 2))
    chans = [L] if stereo is None else [L, R]
    data = _encode_flac(chans, bits=bits, mode=mode, stereo=stereo)
    assert audio_info(data) == (16000, len(chans), n)
    x, sr = decode_audio(data)
    want = np.mean(np.stack(chans, 0), 0) / float(1 << (bits - 1))
    assert np.allclose(x, want, atol=1e-7) and len(x) == n
    # a flipped payload bit is caught by the frame CRC
    broken = bytearray(data)
    broken[len(data) - 5] ^= 0x10
    with pytest.raises(RuntimeError, match="CRC|sync|reserved|stream|size|order"):
        decode_audio(bytes(broken))


def _wav(samples, sr, fmt, bits):
    import struct

    x = np.asarray(samples)
    ch = 1 if x.ndim == 1 else x.shape[1]
    if fmt == 3:
        payload = x.astype("<f4" if bits == 32 else "<f8").tobytes()
    elif bits == 8:
        payload = (x + 128).astype(np.uint8).tobytes()
    elif bits == 24:
        payload = b"".join(int(v).to_bytes(3, "little", signed=True) for v in x.reshape(-1))
    else:
        payload = x.astype(f"<i{bits // 8}").tobytes()
    fmt_chunk = struct.pack("<HHIIHH", fmt, ch, sr, sr * ch * bits // 8, ch * bits // 8, bits)
    body = b"WAVE" + b"fmt " + struct.pack("<I", 16) + fmt_chunk + b"LIST" + struct.pack("<I", 3) + b"abc\x00" + \
        b"data" + struct.pack("<I", len(payload)) + payload
    return b"RIFF" + struct.pack("<I", len(body)) + body


def test_wav_decode_formats_and_batch(tmp_path):
    from caiman_asr_amd.data.audio import audio_info, decode_audio, decode_files

    rng = np.random.default_rng(1)
    i16 = rng.integers(-30000, 30000, 1000)
    x, sr = decode_audio(_wav(i16, 16000, 1, 16))
    assert sr == 16000 and np.array_equal(x, (i16 / 32768.0).astype(np.float32))
    st = rng.integers(-30000, 30000, (700, 2))
    x, _ = decode_audio(_wav(st, 8000, 1, 16))
    assert np.allclose(x, st.mean(1) / 32768.0, atol=1e-7) and audio_info(_wav(st, 8000, 1, 16)) == (8000, 2, 700)
    i24 = rng.integers(-(1 << 22), 1 << 22, 300)
    assert np.allclose(decode_audio(_wav(i24, 16000, 1, 24))[0], i24 / float(1 << 23), atol=1e-7)
    i8 = rng.integers(-100, 100, 300)
    assert np.allclose(decode_audio(_wav(i8, 16000, 1, 8))[0], i8 / 128.0, atol=1e-7)
    f32 = rng.uniform(-1, 1, 256).astype(np.float32)
    assert np.array_equal(decode_audio(_wav(f32, 16000, 3, 32))[0], f32)
    with pytest.raises(RuntimeError, match="unknown audio container"):
        decode_audio(b"OggS" + bytes(100))
    with pytest.raises(RuntimeError, match="unsupported WAVE"):
        decode_audio(_wav(i16, 16000, 1, 16).replace(b"\x01\x00\x01\x00", b"\x07\x00\x01\x00", 1))
    # batch: threads, zero padding, lengths, error naming the file
    paths = []
    for k, n in enumerate((1000, 400, 0, 777)):
        p = tmp_path / f"u{k}.wav"
        p.write_bytes(_wav(i16[:n], 16000, 1, 16))
        paths.append(str(p))
    flac = tmp_path / "c.flac"
    flac.write_bytes(open(os.path.join(GOLD, "ref_clip.flac"), "rb").read())
    out = np.full((5, 142240), 7.0, np.float32)
    lens, rates = decode_files(paths + [str(flac)], out, n_threads=3)
    assert lens.tolist() == [1000, 400, 0, 777, 142240] and rates.tolist() == [16000] * 5
    assert np.array_equal(out[1, :400], (i16[:400] / 32768.0).astype(np.float32)) and not out[1, 400:].any()
    assert not out[2].any() and out[4, 1000] != 0
    with pytest.raises(RuntimeError, match="nope.wav"):
        decode_files(paths + [str(tmp_path / "nope.wav")], np.zeros((5, 2000), np.float32))
    with pytest.raises(RuntimeError, match="too small"):
        decode_files(paths[:1], np.zeros((1, 10), np.float32))


def test_tokenizer_roundtrip():
    from caiman_asr_amd.data.tokenizer import Tokenizer

    spm_path = "/root/reference/training/tests/test_data/librispeech29.model"
    if not os.path.exists(spm_path):
        pytest.skip("sentencepiece model of the reference's test data is not on this machine")
    tok = Tokenizer(labels=list(" abcdefghijklmnopqrstuvwxyz'"), sentpiece_model=spm_path)
    ids = tok.tokenize("the cat  sat")
    assert tok.num_labels == 29 and 0 not in ids and tok.detokenize(ids) == "the cat sat" and tok.detokenize(0) == "⁇"
