"""CPU checks of the oracle's stream driver used by the GPU parity tests (tests/helpers.py)."""
import os
import numpy as np

import helpers
from helpers import PKT_BLOCK_FLAG, PKT_EOS, PKT_NEXT_FLAG, PKT_NO_FLOOR, PKT_NOT_DECODED, PKT_PREV_FLAG


def _pk(flags, spec, channels):
    out = []
    for f in range(len(flags)):
        half = 1024 if flags[f] & 1 else 128
        out.append({"flags": int(flags[f]) | PKT_NO_FLOOR, "residue": spec[f, :, :half].reshape(-1), "granule": -1})
    return out


def test_packet_driver_equals_batch_driver(oracle):
    frames = 60
    flags = helpers.markov_block_flags(frames, seed=3)
    assert (flags & 1).min() == 0 and (flags & 1).max() == 1
    spec = helpers.gaussian_spectra((frames, 2, 1024), seed=4)
    a, pos, _ = helpers.oracle_decode(oracle, 2, 256, 2048, _pk(flags, spec, 2))
    b = oracle.synth_stream_planar(2, 256, 2048, flags & 7, spec)
    assert a.shape == b.shape and np.array_equal(a, b)
    # per-packet sample counts follow Mode.cs:30-66: total = sum over packets 1.. of SampleCount
    total = 0
    for f in range(1, frames):
        i = oracle.packet_info(256, 2048, flags[f] & 1, bool(flags[f] & 2), bool(flags[f] & 4))
        total += i.SampleCount
    assert a.shape[1] == total == pos


def test_eos_trim_and_drain(oracle):
    L = PKT_BLOCK_FLAG | PKT_PREV_FLAG | PKT_NEXT_FLAG
    spec = helpers.gaussian_spectra((5, 1, 1024), seed=1)
    pk = _pk(np.array([L] * 5, dtype=np.uint8), spec, 1)
    pk[2]["granule"] = 2048
    pk[3]["flags"] |= PKT_EOS
    pk[3]["granule"] = 2048 + 100
    out, pos, _ = helpers.oracle_decode(oracle, 1, 256, 2048, pk)
    assert out.shape == (1, 2148) and pos == 2148
    pk2 = _pk(np.array([L] * 3, dtype=np.uint8), spec, 1) + [{"flags": PKT_NOT_DECODED | PKT_EOS}]
    out2, _, _ = helpers.oracle_decode(oracle, 1, 256, 2048, pk2)
    assert out2.shape == (1, 3072)
    # the drained tail is the raw second half of the last IMDCT block, un-windowed (quirk q4)
    raw = oracle.mdct_reverse(spec[2, 0][None, :], 2048)[0]
    assert np.array_equal(out2[0, 2048:], raw[1024:])


def test_floor_and_coupling_driver_runs(oracle):
    rng = np.random.default_rng(0)
    floors = [(helpers.SHORT_XLIST, 2), (helpers.LONG_XLIST, 2)]
    mappings = [{"coupling": [(0, 1)], "channel_floor": [0, 0]}, {"coupling": [(0, 1)], "channel_floor": [1, 1]}]
    flags = helpers.markov_block_flags(10, seed=2)
    pks = []
    for f in range(10):
        bf = flags[f] & 1
        half = 1024 if bf else 128
        posts, counts = helpers.random_posts(rng, helpers.LONG_XLIST if bf else helpers.SHORT_XLIST, 2, 2, 0.2)
        pks.append({"flags": int(flags[f]), "mapping": int(bf),
                    "residue": rng.standard_normal(2 * half).astype(np.float32), "posts": posts,
                    "post_count": counts})
    out, _, _ = helpers.oracle_decode(oracle, 2, 256, 2048, pks, floors=floors, mappings=mappings)
    assert out.shape[0] == 2 and out.shape[1] > 0 and np.isfinite(out).all()
    # unwrapped posts stay inside the dB table for these generators
    f1 = oracle.floor1_init(helpers.LONG_XLIST, 2)
    for _ in range(200):
        posts, counts = helpers.random_posts(rng, helpers.LONG_XLIST, 2, 1)
        fy, _ = oracle.floor1_unwrap(f1, posts[0], int(counts[0]))
        assert fy[:29].min() >= 0 and fy[:29].max() * 2 <= 255


def test_the_c_stream_driver_equals_the_packet_by_packet_drive(oracle):
    """oracle.FlooredStream (orc_synth_stream_floored: one stream's packets through the restated Mapping.DecodePacket tail +
    StreamDecoder in C, what bench.py's CPU baselines of the fused workloads time) gives the bits of helpers.oracle_decode,
    which drives the same restated functions packet by packet: on the reference's stereo and mono fixtures (Residue2 vector,
    coupling, EOS trim, the skipped trailing packet of issue6test.ogg) and on a synthetic 6-channel stream."""
    import synthetic_streams as ss
    from vorbispizza_amd.front import OggVorbisFile
    cases = [open(os.path.join(os.path.dirname(__file__), "golden", n), "rb").read() for n in ("3test.ogg", "issue6test.ogg", "2test.ogg")]
    stream, rng = ss.ALL["six_channels_51"]()
    cases.append(bytes(stream.build(rng, 24)[0]))
    for raw in cases:
        f = OggVorbisFile(raw)
        pk, res, posts, counts = f.decode_packets()
        ref, _, _ = helpers.oracle_decode(oracle, f.channels, f.block_size0, f.block_size1,
                                          helpers.packets_for_oracle(f, pk, res, posts, counts), floors=f.floors, mappings=f.mappings)
        fs = oracle.FlooredStream(f.channels, f.block_size0, f.block_size1, pk, res, posts, counts, floors=f.floors, mappings=f.mappings)
        assert fs.run() == ref.shape[1] > 0
        assert np.array_equal(fs.pcm[:, :fs.total].view(np.uint32), ref.view(np.uint32))
