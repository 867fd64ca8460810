"""Error behaviour of the C ABI: every misuse returns a negative status (never a crash, never a throw into the
host), leaves the decoder state as it was, and explains itself through vpz_context_last_error.  The reference's
counterparts are .NET exceptions (ArgumentException / InvalidDataException / ArgumentOutOfRangeException)."""
import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import __graft_entry__ as ge
    ge.build()
    from vorbispizza_amd import Context
    c = Context(0)
    yield c
    c.close()


def _batch(frames=6, channels=2):
    from vorbispizza_amd import capi, make_packets
    pk = make_packets(frames)
    pk["flags"] = capi.PKT_BLOCK_FLAG | capi.PKT_PREV_FLAG | capi.PKT_NEXT_FLAG | capi.PKT_NO_FLOOR
    pk["granule"] = -1
    pk["residue_offset"] = np.arange(frames, dtype=np.int64) * 1024 * channels
    res = helpers.gaussian_spectra((frames, channels, 1024), seed=9).reshape(-1)
    return pk, res


def _status(fn):
    from vorbispizza_amd import SynthError
    with pytest.raises(SynthError) as e:
        fn()
    return e.value.status


def test_create_rejects_bad_configurations(ctx):
    from vorbispizza_amd import Decoder, capi
    assert _status(lambda: Decoder(ctx, 0, 256, 2048)) == capi.E_INVALID_ARG            # channels
    assert _status(lambda: Decoder(ctx, 2, 2048, 256)) == capi.E_INVALID_ARG            # size0 > size1
    assert _status(lambda: Decoder(ctx, 2, 100, 2048)) == capi.E_UNSUPPORTED            # not a power of two
    assert _status(lambda: Decoder(ctx, 2, 32, 2048)) == capi.E_UNSUPPORTED             # below 64
    assert _status(lambda: Decoder(ctx, 2, 256, 16384)) == capi.E_UNSUPPORTED           # above 8192
    # coupling a channel with itself / out of range (Mapping.cs:41 throws InvalidDataException)
    bad = [{"coupling": [(1, 1)], "channel_floor": [0, 0]}]
    assert _status(lambda: Decoder(ctx, 2, 256, 2048, floors=[(helpers.LONG_XLIST, 2)], mappings=bad)) == capi.E_INVALID_ARG
    bad = [{"coupling": [(0, 2)], "channel_floor": [0, 0]}]
    assert _status(lambda: Decoder(ctx, 2, 256, 2048, floors=[(helpers.LONG_XLIST, 2)], mappings=bad)) == capi.E_INVALID_ARG
    # floor index out of range, duplicate X (Floor1.cs:141), multiplier out of range
    bad = [{"coupling": [], "channel_floor": [3, 0]}]
    assert _status(lambda: Decoder(ctx, 2, 256, 2048, floors=[(helpers.LONG_XLIST, 2)], mappings=bad)) == capi.E_INVALID_ARG
    ok_map = [{"coupling": [], "channel_floor": [0, 0]}]
    assert _status(lambda: Decoder(ctx, 2, 256, 2048, floors=[([0, 128, 5, 5], 2)], mappings=ok_map)) == capi.E_INVALID_ARG
    assert _status(lambda: Decoder(ctx, 2, 256, 2048, floors=[([0, 128, 5], 7)], mappings=ok_map)) == capi.E_INVALID_ARG
    f0 = {"order": 0, "rate": 44100, "bark_map_size": 64, "amp_bits": 6, "amp_ofs": 40}
    assert _status(lambda: Decoder(ctx, 2, 256, 2048, floors=[f0], mappings=ok_map)) == capi.E_INVALID_ARG
    # Floor1.cs:96-97 builds _xList[0] = 0, _xList[1] = 1 << rangeBits: the render relies on the post at x = 0,
    # and `Posts = new int[64]` (Floor1.cs:17) cannot hold a 65th post
    assert _status(lambda: Decoder(ctx, 2, 256, 2048, floors=[([3, 128, 5], 2)], mappings=ok_map)) == capi.E_INVALID_ARG
    assert _status(lambda: Decoder(ctx, 2, 256, 2048, floors=[([0, 0x7FFF + 1, 5], 2)], mappings=ok_map)) == capi.E_INVALID_ARG
    x65 = [0, 1024] + list(range(1, 64))
    assert len(x65) == 65
    assert _status(lambda: Decoder(ctx, 2, 256, 2048, floors=[(x65, 2)], mappings=ok_map)) == capi.E_INVALID_ARG
    Decoder(ctx, 2, 256, 2048, floors=[(x65[:64], 2)], mappings=ok_map).close()


def test_synth_rejects_bad_calls_and_keeps_its_state(ctx):
    from vorbispizza_amd import Decoder, capi
    dec = Decoder(ctx, 2, 256, 2048)
    pk, res = _batch()
    cap = 8192
    out = np.zeros(2 * cap, dtype=np.float32)

    def call(packets=pk, residue=res, capacity=cap, layout=capi.OUT_PLANAR, stride=cap, mem=capi.MEM_HOST, posts=None,
             counts=None, offs=None, residue_floats=None, n_records=None):
        return dec.synth_raw(packets, residue, posts, counts, out, offs, capacity, layout, stride, mem,
                             residue_floats=residue_floats, n_records=n_records)

    assert _status(lambda: call(mem=7)) == capi.E_INVALID_ARG
    assert _status(lambda: call(layout=5)) == capi.E_INVALID_ARG
    assert _status(lambda: call(capacity=100)) == capi.E_CAPACITY
    assert "capacity" in ctx.last_error().lower()
    assert _status(lambda: call(stride=10)) == capi.E_INVALID_ARG                       # planar stride < capacity
    # per-stream bounds (vpz_decoder_set_stream_capacities) tighten the call's: one value per stream of the decoder
    assert _status(lambda: dec.set_stream_capacities([cap, cap])) == capi.E_INVALID_ARG
    assert _status(lambda: dec.set_stream_capacities([-1])) == capi.E_INVALID_ARG
    dec.set_stream_capacities([5 * 1024 - 1])
    assert _status(call) == capi.E_CAPACITY
    dec.set_stream_capacities([5 * 1024])                                               # (what the batch produces: enough)
    bad = pk.copy()
    bad["stream"][2] = 3
    assert _status(lambda: call(packets=bad)) == capi.E_INVALID_ARG                     # stream index
    bad = pk.copy()
    bad["residue_offset"][1] = -4
    assert _status(lambda: call(packets=bad)) == capi.E_INVALID_ARG
    bad = pk.copy()
    bad["flags"] &= ~np.uint8(capi.PKT_NO_FLOOR)                                        # floor wanted, none configured
    assert _status(lambda: call(packets=bad)) == capi.E_INVALID_ARG
    # ABI v3: the extents of the input buffers are part of the call -- a packet whose residue reaches beyond them is
    # refused instead of read (the reference's Span<float> would throw, Mapping.cs:98)
    assert _status(lambda: call(residue_floats=res.size - 1)) == capi.E_INVALID_ARG
    assert "residue_floats" in ctx.last_error()
    assert _status(lambda: call(residue_floats=-1)) == capi.E_INVALID_ARG
    bad = pk.copy()
    bad["residue_offset"][len(pk) - 1] += 4
    assert _status(lambda: call(packets=bad)) == capi.E_INVALID_ARG                     # one packet past the end
    assert int(call(residue_floats=res.size)[0]) == 5 * 1024                            # exactly enough is enough
    dec.reset(-1)
    dec.set_position(0)
    # nothing above touched the stream state: the good call still decodes from the start
    w = call()
    assert int(w[0]) == 5 * 1024 and dec.position(0) == 5 * 1024
    dec.set_stream_capacities(None)
    ref = Decoder(ctx, 2, 256, 2048)
    want = ref.synth(pk, res)[0]
    assert np.array_equal(out.reshape(2, cap)[:, :5 * 1024], want)
    assert _status(lambda: dec.set_position(0, stream=4)) == capi.E_INVALID_ARG
    assert _status(lambda: dec.reset(9)) == capi.E_INVALID_ARG
    ref.close()
    dec.close()


def test_floor0_needs_its_data_and_mapping_index_is_checked(ctx):
    from vorbispizza_amd import Decoder, capi
    f0 = {"order": 8, "rate": 44100, "bark_map_size": 64, "amp_bits": 6, "amp_ofs": 40}
    maps = [{"coupling": [], "channel_floor": [0, 0]}]
    dec = Decoder(ctx, 2, 256, 2048, floors=[f0], mappings=maps)
    pk, res = _batch()
    pk["flags"] &= ~np.uint8(capi.PKT_NO_FLOOR)
    posts = np.zeros((len(pk) * 2, 64), dtype=np.int16)
    counts = np.ones(len(pk) * 2, dtype=np.uint8)
    out = np.zeros(2 * 8192, dtype=np.float32)
    st = _status(lambda: dec.synth_raw(pk, res, posts, counts, out, None, 8192, capi.OUT_PLANAR, 8192, capi.MEM_HOST))
    assert st == capi.E_INVALID_ARG and "type-0 floors need" in ctx.last_error()
    bad = pk.copy()
    bad["mapping"][0] = 5
    st = _status(lambda: dec.synth_raw(bad, res, posts, counts, out, None, 8192, capi.OUT_PLANAR, 8192, capi.MEM_HOST))
    assert st == capi.E_INVALID_ARG
    # fewer post records than packets * channels (ABI v3)
    st = _status(lambda: dec.synth_raw(pk, res, posts, counts, out, None, 8192, capi.OUT_PLANAR, 8192, capi.MEM_HOST,
                                       n_records=len(pk) * 2 - 1))
    assert st == capi.E_INVALID_ARG and "post records" in ctx.last_error()
    dec.close()


def test_imdct_batch_argument_checks(ctx):
    from vorbispizza_amd import SynthError, capi
    x = np.zeros((4, 500), dtype=np.float32)
    with pytest.raises((SynthError, ValueError, AssertionError)):
        ctx.imdct_batch(x, 1000)      # not a power of two
    y = ctx.imdct_batch(np.zeros((0, 1024), dtype=np.float32), 2048)
    assert y.shape == (0, 2048)
