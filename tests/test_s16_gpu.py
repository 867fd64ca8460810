"""The 16-bit output layouts (VPZ_OUT_INTERLEAVED_S16 / VPZ_OUT_PLANAR_S16): every sample must be exactly what the
reference's own test computes from the float sample, `(int)(x * 32768f)` clamped to the short range
(NVorbis.Tests/AssetTest.cs:131-132) -- 0 LSB against the conversion of the library's own float output (same
kernel arithmetic, so bit-exact), and within the reference's acceptance band against the oracle."""
import os

import numpy as np
import pytest

import helpers
from helpers import PKT_NO_FLOOR

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ctx():
    import __graft_entry__ as ge
    ge.build()
    from vorbispizza_amd import Context
    c = Context(0)
    yield c
    c.close()


def to_s16(x):  # AssetTest.cs:131-132
    return np.clip((x.astype(np.float32) * np.float32(32768.0)).astype(np.int64), -32768, 32767).astype(np.int16)


@pytest.mark.parametrize("channels", [1, 2, 3, 6])
@pytest.mark.parametrize("clip", [False, True])
def test_s16_is_the_reference_conversion_of_the_float_output(ctx, channels, clip):
    """Loud spectra (|PCM| well beyond 1): the clamp and, with ClipSamples, the clip in front of it both matter.
    Mixed 256 / 2048 blocks, every emission path of the kernel (steady state, transitions, scalar tail)."""
    from vorbispizza_amd import Decoder, capi, make_packets
    frames = 60
    flags = helpers.markov_block_flags(frames, seed=channels)
    spec = helpers.gaussian_spectra((frames, channels, 1024), seed=50 + channels, sigma=2.0 ** -5)
    pk = make_packets(frames)
    res, off = [], 0
    for f in range(frames):
        half = 1024 if flags[f] & 1 else 128
        pk[f]["flags"], pk[f]["granule"], pk[f]["residue_offset"] = flags[f] | PKT_NO_FLOOR, -1, off
        res.append(spec[f, :, :half].reshape(-1))
        off += channels * half
    pk[frames - 1]["flags"] |= capi.PKT_EOS          # EOS trim to an odd length: the scalar store path
    pk[frames - 1]["granule"] = 3001
    res = np.concatenate(res)
    for lay_f, lay_s in ((capi.OUT_INTERLEAVED, capi.OUT_INTERLEAVED_S16), (capi.OUT_PLANAR, capi.OUT_PLANAR_S16)):
        outs = []
        for layout in (lay_f, lay_s):
            dec = Decoder(ctx, channels, 256, 2048, clip_samples=clip)
            outs.append(dec.synth(pk, res, out_layout=layout)[0])
            dec.close()
        f32, s16 = outs
        assert s16.dtype == np.int16 and s16.shape == f32.shape and f32.size > 0
        if not clip:
            assert np.abs(f32).max() > 1.5                   # the clamp is exercised
        else:
            assert np.abs(f32).max() == np.float32(0.99999994)   # ... and so is the clip in front of it
        assert np.array_equal(s16, to_s16(f32)), (channels, clip, lay_s)
        if clip:
            assert np.abs(s16.astype(np.int32)).max() <= 32767


@pytest.mark.parametrize("name", ["2test.ogg", "3test.ogg"])
def test_reader_delivers_s16_like_the_asset_test_computes_it(ctx, oracle, name):
    """The VorbisReader mirror in s16 mode against the oracle's float PCM converted the reference's way."""
    from vorbispizza_amd.front import OggVorbisFile, VorbisReader
    path = os.path.join(GOLDEN, name)
    f = OggVorbisFile(path)
    pk, res, posts, counts = f.decode_packets()
    ref, _, _ = helpers.oracle_decode(oracle, f.channels, f.block_size0, f.block_size1,
                                      helpers.packets_for_oracle(f, pk, res, posts, counts), floors=f.floors,
                                      mappings=f.mappings, clip=True, interleave=True)
    rdr = VorbisReader(ctx, path, s16=True)
    buf = np.zeros(2048 * 8, dtype=np.int16)
    chunks = []
    while True:
        n = rdr.ReadSamples(buf)
        if n == 0:
            break
        chunks.append(buf[: n * rdr.Channels].reshape(n, rdr.Channels).copy())
    got = np.concatenate(chunks)
    assert got.shape == ref.shape
    # the float paths differ by <= 1e-5: a sample may sit across a truncation boundary, never further
    assert np.abs(got.astype(np.int32) - to_s16(ref).astype(np.int32)).max() <= 1
    assert (got == to_s16(ref)).mean() > 0.99
    # a float read on an s16 reader is refused
    from vorbispizza_amd import SynthError
    with pytest.raises(SynthError):
        rdr.ReadSamples(np.zeros(64, dtype=np.float32))
    rdr.Dispose()


def test_s16_through_the_any_size_path(ctx):
    """Block sizes outside {256 .. 2048} take the three-pass path; its store converts the same way."""
    from vorbispizza_amd import Decoder, capi, make_packets
    frames, channels = 12, 2
    flags = np.full(frames, 7, dtype=np.uint8)
    spec = helpers.gaussian_spectra((frames, channels, 2048), seed=4, sigma=2.0 ** -6)
    pk = make_packets(frames)
    pk["flags"] = flags | PKT_NO_FLOOR
    pk["granule"] = -1
    pk["residue_offset"] = np.arange(frames) * channels * 2048
    outs = []
    for layout in (capi.OUT_INTERLEAVED, capi.OUT_INTERLEAVED_S16):
        dec = Decoder(ctx, channels, 128, 4096)
        outs.append(dec.synth(pk, spec.reshape(-1), out_layout=layout)[0])
        dec.close()
    assert np.array_equal(outs[1], to_s16(outs[0])) and np.abs(outs[0]).max() > 1.0
