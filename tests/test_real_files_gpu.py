"""BASELINE configs[0]/[4] shape: real .ogg fixtures through CPU front end -> vpz_decoder_synth (GPU)
against the same packets through the oracle."""
import os

import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SAMPLES = {"1test.ogg": 17318, "2test.ogg": 315790, "3test.ogg": 288094, "issue6test.ogg": 548160}


@pytest.fixture(scope="module")
def ctx():
    import __graft_entry__ as ge
    ge.build()
    from vorbispizza_amd import Context
    c = Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("name", sorted(SAMPLES))
@pytest.mark.parametrize("layout", ["planar", "interleaved"])
def test_fixture_decodes_like_the_oracle(ctx, oracle, name, layout):
    from vorbispizza_amd import Decoder, SynthError, capi
    from vorbispizza_amd.front import OggVorbisFile
    f = OggVorbisFile(os.path.join(GOLDEN, name))
    pk, res, posts, counts = f.decode_packets()
    dec = Decoder(ctx, f.channels, f.block_size0, f.block_size1, floors=f.floors, mappings=f.mappings)
    lay = capi.OUT_PLANAR if layout == "planar" else capi.OUT_INTERLEAVED
    cap = int(f.audio_packets) * 1024 + 2048
    out = np.zeros(f.channels * cap, dtype=np.float32)
    try:
        written = dec.synth_raw(pk, res, posts, counts, out, None, cap, lay, cap, capi.MEM_HOST)
        assert name != "issue6test.ogg"
    except SynthError as e:
        # the trailing empty packet of issue6test.ogg: reported, but everything else was synthesised
        assert name == "issue6test.ogg" and e.status == capi.E_WINDOW_MISMATCH
        written = None
    opk = helpers.packets_for_oracle(f, pk, res, posts, counts)
    ref, pos, _ = helpers.oracle_decode(oracle, f.channels, f.block_size0, f.block_size1, opk,
                                        floors=f.floors, mappings=f.mappings, interleave=(layout == "interleaved"))
    n = SAMPLES[name]
    if written is not None:
        assert int(written[0]) == n
    got = out[: f.channels * cap].reshape(f.channels, cap)[:, :n] if layout == "planar" else \
        out[: n * f.channels].reshape(n, f.channels)
    assert ref.shape == got.shape
    assert np.abs(got - ref).max() <= 1e-5 * max(1.0, float(np.abs(ref).max()))
    assert dec.position(0) == pos == n
    dec.close()


def test_many_streams_of_real_files_in_one_batch(ctx, oracle):
    """configs[4] in miniature: independent copies of the two stereo fixtures as separate streams of
    one decoder group each, decoded in one call per group."""
    from vorbispizza_amd import Decoder, capi
    from vorbispizza_amd.front import OggVorbisFile
    f = OggVorbisFile(os.path.join(GOLDEN, "3test.ogg"))
    n_streams = 4
    parts = [f.decode_packets(stream_id=s) for s in range(n_streams)]
    res_off = np.cumsum([0] + [p[1].size for p in parts])
    pk = np.concatenate([p[0] for p in parts])
    for s in range(n_streams):
        sel = pk["stream"] == s
        pk["residue_offset"][sel] += res_off[s]
    res = np.concatenate([p[1] for p in parts])
    posts = np.concatenate([p[2] for p in parts])
    counts = np.concatenate([p[3] for p in parts])
    dec = Decoder(ctx, 2, 256, 2048, floors=f.floors, mappings=f.mappings, n_streams=n_streams)
    outs = dec.synth(pk, res, posts, counts, out_layout=capi.OUT_INTERLEAVED)
    assert all(o.shape == (288094, 2) for o in outs)
    for s in range(1, n_streams):
        assert np.array_equal(outs[0], outs[s])
    opk = helpers.packets_for_oracle(f, *parts[0])
    ref, _, _ = helpers.oracle_decode(oracle, 2, 256, 2048, opk, floors=f.floors, mappings=f.mappings, interleave=True)
    assert np.abs(outs[0] - ref).max() <= 1e-5 * max(1.0, float(np.abs(ref).max()))
    dec.close()


@pytest.mark.parametrize("name", ["mono_floor1_res1", "stereo_coupled_res2", "three_channels_two_submaps", "stereo_floor0"])
def test_synthetic_stream_decodes_like_the_oracle(ctx, oracle, name):
    """Streams from the spec-based writer (tests/vorbis_writer.py): Floor0, residue 0, two submaps, block
    sizes outside 256/2048 -- container to PCM through the front end and the GPU, against the oracle."""
    import synthetic_streams as ss
    from vorbispizza_amd import Decoder, capi
    from vorbispizza_amd.front import OggVorbisFile
    stream, rng = ss.ALL[name]()
    ogg, _ = stream.build(rng, 40)
    f = OggVorbisFile(ogg)
    pk, res, posts, counts = f.decode_packets()
    dec = Decoder(ctx, f.channels, f.block_size0, f.block_size1, floors=f.floors, mappings=f.mappings)
    if f.floor0_data is not None:
        dec.set_floor0_data(*f.floor0_data)
    outs = dec.synth(pk, res, posts, counts, out_layout=capi.OUT_PLANAR)
    opk = helpers.packets_for_oracle(f, pk, res, posts, counts)
    ref, pos, _ = helpers.oracle_decode(oracle, f.channels, f.block_size0, f.block_size1, opk,
                                        floors=f.floors, mappings=f.mappings)
    assert outs[0].shape == ref.shape and ref.shape[1] == f.last_granule
    assert np.isfinite(ref).all()
    assert np.abs(outs[0] - ref).max() <= 1e-5 * max(1.0, float(np.abs(ref).max()))
    assert dec.position(0) == pos
    dec.close()


def test_synthetic_stream_through_the_reader_mirror(ctx, oracle):
    import synthetic_streams as ss
    from vorbispizza_amd.front import OggVorbisFile, VorbisReader
    stream, rng = ss.three_channels_two_submaps(seed=21)
    ogg, _ = stream.build(rng, 30)
    f = OggVorbisFile(ogg)
    pk, res, posts, counts = f.decode_packets()
    ref, _, _ = helpers.oracle_decode(oracle, 3, f.block_size0, f.block_size1,
                                      helpers.packets_for_oracle(f, pk, res, posts, counts),
                                      floors=f.floors, mappings=f.mappings, interleave=True, clip=True)
    r = VorbisReader(ctx, ogg, batch_packets=7)
    got = []
    buf = np.zeros(3 * 700, dtype=np.float32)
    while True:
        n = r.ReadSamples(buf)
        if n == 0:
            break
        got.append(buf[: n * 3].reshape(n, 3).copy())
    got = np.concatenate(got)
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= 1e-5
    r.Dispose()


@pytest.mark.parametrize("seed", range(40, 64))
def test_random_synthetic_setups_decode_like_the_oracle(ctx, oracle, seed):
    """Random setups from the spec-based writer (1-3 channels, any block-size pair 64..4096, residue 0 / 1 / 2,
    one or two submaps, up to two coupling steps): container to PCM, planar and interleaved, against the oracle."""
    import synthetic_streams as ss
    from vorbispizza_amd import Decoder, capi
    from vorbispizza_amd.front import OggVorbisFile
    stream, rng = ss.random_stream(seed)
    ogg, _ = stream.build(rng, 16, packets_per_page=int(rng.integers(1, 6)))
    f = OggVorbisFile(ogg)
    pk, res, posts, counts = f.decode_packets()
    opk = helpers.packets_for_oracle(f, pk, res, posts, counts)
    ref, pos, _ = helpers.oracle_decode(oracle, f.channels, f.block_size0, f.block_size1, opk,
                                        floors=f.floors, mappings=f.mappings)
    # (the counted length, not the writer's granule: a long first block followed by a short one loses its flat part
    # in the reference -- `_prevPacketStart = rightStart` for the first packet, StreamDecoder.cs:679)
    assert ref.shape[1] == f.total_samples and np.isfinite(ref).all()
    scale = max(1.0, float(np.abs(ref).max()))
    dec = Decoder(ctx, f.channels, f.block_size0, f.block_size1, floors=f.floors, mappings=f.mappings)
    if f.floor0_data is not None:
        dec.set_floor0_data(*f.floor0_data)
    planar = dec.synth(pk, res, posts, counts, out_layout=capi.OUT_PLANAR)[0]
    assert planar.shape == ref.shape and dec.position(0) == pos
    assert np.abs(planar - ref).max() <= 1e-5 * scale
    dec.reset(0)
    dec.set_position(0)
    if f.floor0_data is not None:
        dec.set_floor0_data(*f.floor0_data)
    inter = dec.synth(pk, res, posts, counts, out_layout=capi.OUT_INTERLEAVED)[0]
    assert np.array_equal(inter, planar.T)
    dec.close()


def test_configs4_share_of_one_gpu_all_128_streams_both_layouts(ctx, oracle):
    """BASELINE configs[4] at the size one GPU gets out of eight: 64 x 3test.ogg + 64 x issue6test.ogg as 128
    independent streams, two decoder groups (one per setup header, StreamDecoder.cs:45-49 state per stream), every
    one of the 128 outputs against the oracle, interleaved and planar."""
    from vorbispizza_amd import Decoder, SynthError, capi
    from vorbispizza_amd.front import OggVorbisFile
    copies = 64
    for name in ("3test.ogg", "issue6test.ogg"):
        f = OggVorbisFile(os.path.join(GOLDEN, name))
        pk, res, posts, counts = f.decode_packets()
        opk = helpers.packets_for_oracle(f, pk, res, posts, counts)
        ref, ref_pos, _ = helpers.oracle_decode(oracle, f.channels, f.block_size0, f.block_size1, opk, floors=f.floors,
                                                mappings=f.mappings, interleave=True)
        total, C_ = ref.shape
        tol = 1e-5 * max(1.0, float(np.abs(ref).max()))
        n = len(pk)
        pk_all = np.tile(pk, copies)
        pk_all["stream"] = np.repeat(np.arange(copies, dtype=np.int32), n)
        # every stream reads its own copy of the residue, as separately decoded files would
        res_all = np.tile(res, copies)
        pk_all["residue_offset"] += np.repeat(np.arange(copies, dtype=np.int64) * res.size, n)
        posts_all, counts_all = np.tile(posts, (copies, 1)), np.tile(counts, copies)
        cap = total + 2048
        for layout in (capi.OUT_INTERLEAVED, capi.OUT_PLANAR):
            dec = Decoder(ctx, C_, f.block_size0, f.block_size1, floors=f.floors, mappings=f.mappings, n_streams=copies)
            out = np.zeros(copies * cap * C_, dtype=np.float32)
            offs = np.arange(copies, dtype=np.int64) * cap * C_
            try:
                w = dec.synth_raw(pk_all, res_all, posts_all, counts_all, out, offs, cap, layout, cap, capi.MEM_HOST)
            except SynthError as e:  # issue6test.ogg's last packet (the reference's OverlapBuffers throws on it)
                assert e.status == capi.E_WINDOW_MISMATCH and name == "issue6test.ogg"
                w = dec.last_packet_samples(len(pk_all)).reshape(copies, n).sum(axis=1)
            assert (np.asarray(w) == total).all()
            assert [dec.position(s) for s in range(copies)] == [ref_pos] * copies
            blocks = out.reshape(copies, cap * C_)
            worst = 0.0
            for s in range(copies):
                got = (blocks[s, : total * C_].reshape(total, C_) if layout == capi.OUT_INTERLEAVED
                       else blocks[s].reshape(C_, cap)[:, :total].T)
                worst = max(worst, float(np.abs(got - ref).max()))
            assert worst <= tol, (name, layout, worst)
            dec.close()


def test_streams_with_different_setup_headers_in_one_decoder(ctx):
    """sharding.merge_setups: 3test.ogg and issue6test.ogg streams (different floors, same block sizes) decoded by ONE
    decoder over the union of their floors and mappings -- what bench.py's configs[4] lines run -- give, stream for
    stream, the bits the per-file decoders give (which the tests above hold against the oracle)."""
    from vorbispizza_amd import Decoder, SynthError, capi, sharding
    from vorbispizza_amd.front import OggVorbisFile
    copies = 3
    parts, separate = [], []
    for name in ("3test.ogg", "issue6test.ogg"):
        f = OggVorbisFile(os.path.join(GOLDEN, name))
        pk, res, posts, counts = f.decode_packets()
        n, total = len(pk), SAMPLES[name]
        pk_all = np.tile(pk, copies)
        pk_all["stream"] = np.repeat(np.arange(copies, dtype=np.int32), n)
        pk_all["residue_offset"] += np.repeat(np.arange(copies, dtype=np.int64) * res.size, n)
        parts.append((f, pk_all, np.tile(res, copies), np.tile(posts, (copies, 1)), np.tile(counts, copies), total))
        cap = total + 2048
        dec = Decoder(ctx, f.channels, f.block_size0, f.block_size1, floors=f.floors, mappings=f.mappings, n_streams=copies)
        out = np.zeros(copies * cap * f.channels, dtype=np.float32)
        offs = np.arange(copies, dtype=np.int64) * cap * f.channels
        try:
            dec.synth_raw(parts[-1][1], parts[-1][2], parts[-1][3], parts[-1][4], out, offs, cap, capi.OUT_INTERLEAVED, cap,
                          capi.MEM_HOST)
        except SynthError as e:
            assert e.status == capi.E_WINDOW_MISMATCH
        separate += [out.reshape(copies, cap * f.channels)[s, : total * f.channels].copy() for s in range(copies)]
        dec.close()
    f0 = parts[0][0]
    floors, mappings, bases = sharding.merge_setups([(p[0].floors, p[0].mappings) for p in parts])
    assert len(floors) <= sum(len(p[0].floors) for p in parts) and bases == [0, len(parts[0][0].mappings)]
    pks, s0, r0 = [], 0, 0
    for (f, pk, res, posts, counts, total), base in zip(parts, bases):
        pk = pk.copy()
        pk["stream"] += s0
        pk["residue_offset"] += r0
        pk["mapping"] += base
        pks.append(pk)
        s0 += copies
        r0 += res.size
    n_all = 2 * copies
    cap = max(p[5] for p in parts) + 2048
    dec = Decoder(ctx, f0.channels, f0.block_size0, f0.block_size1, floors=floors, mappings=mappings, n_streams=n_all)
    out = np.zeros(n_all * cap * f0.channels, dtype=np.float32)
    offs = np.arange(n_all, dtype=np.int64) * cap * f0.channels
    try:
        dec.synth_raw(np.concatenate(pks), np.concatenate([p[2] for p in parts]), np.concatenate([p[3] for p in parts]),
                      np.concatenate([p[4] for p in parts]), out, offs, cap, capi.OUT_INTERLEAVED, cap, capi.MEM_HOST)
    except SynthError as e:
        assert e.status == capi.E_WINDOW_MISMATCH
    totals = [p[5] for p in parts for _ in range(copies)]
    for s in range(n_all):
        got = out.reshape(n_all, cap * f0.channels)[s, : totals[s] * f0.channels]
        assert np.array_equal(got.view(np.uint32), separate[s].view(np.uint32)), s
    dec.close()
