"""ABI v4: the residue's support (`vpz_mapping_config.residue_begin / residue_end`, what Residue0.cs:122-125 clamps every
decode to).  A decoder that is told the support must give the bits of one that is not -- on every route (stereo fast path,
group mode, separate coupling pass, general block sizes) and both input layouts -- as long as the residue IS zero beyond
it, which the setup header guarantees; what lies in the buffer beyond the support is not looked at by group mode."""
import numpy as np
import pytest

import helpers
from test_dual_gpu import ROUTES, same_bits
from test_host_paths_gpu import env, run, stream_major_batch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import __graft_entry__ as ge
    ge.build()
    from vorbispizza_amd import Context
    c = Context(0)
    yield c
    c.close()


def zero_beyond(pk, res, channels, end_short, end_long, interleaved, fill=0.0):
    """res with every packet's bins >= end (per channel) set to `fill`"""
    out = res.copy()
    for p in pk:
        long_ = bool(p["flags"] & 1)
        half = 1024 if long_ else 128
        e = end_long if long_ else end_short
        off = int(p["residue_offset"])
        v = out[off: off + channels * half]
        if interleaved:
            v.reshape(half, channels)[e:, :] = fill
        else:
            v.reshape(channels, half)[:, e:] = fill
    return out


@pytest.mark.parametrize("channels,interleaved", [(2, True), (2, False), (6, True), (6, False), (3, True), (10, True)])
@pytest.mark.parametrize("ends", [(128, 410), (40, 512), (128, 600), (16, 100)])
def test_declared_support_gives_the_bits_of_the_full_vector(ctx, channels, interleaved, ends):
    from vorbispizza_amd import capi
    n_streams, frames = 7, 48
    pk, res, posts, counts = stream_major_batch(n_streams, frames, channels, seed=4400 + channels + ends[1], floor=True,
                                                interleaved=interleaved, p_ls=0.15, p_sl=0.3, silent_prob=0.1)
    pk["mapping"] = pk["flags"] & 1
    res = zero_beyond(pk, res, channels, ends[0], ends[1], interleaved)
    floors = [(helpers.SHORT_XLIST, 2), (helpers.LONG_XLIST, 2)]
    steps = [(0, 1)] if channels < 4 else [(0, 1), (2, 3)]
    plain = [{"coupling": steps, "channel_floor": [0] * channels}, {"coupling": steps, "channel_floor": [1] * channels}]
    told = [dict(m, residue_begin=(0, 0), residue_end=ends) for m in plain]
    routes = ROUTES if channels == 2 else ROUTES[1:]
    for layout in (capi.OUT_PLANAR, capi.OUT_INTERLEAVED):
        for name, kv in routes:
            with env(**dict(kv, VPZ_PAR_MIN_PACKETS=1, VPZ_HOST_THREADS=3)):
                ref = run(ctx, pk, res, posts, counts, n_streams, channels, floors, plain, layout=layout, splits=2)
                got = run(ctx, pk, res, posts, counts, n_streams, channels, floors, told, layout=layout, splits=2)
            same_bits(got, ref, "%s, layout %d, ends %r" % (name, layout, ends))
            assert np.abs(ref[0].astype(np.float64)).max() > 0


@pytest.mark.parametrize("channels", [2, 6])
def test_group_mode_does_not_look_beyond_the_support(ctx, channels):
    """Group mode neither loads nor stages the upper half of a 2048 block whose support ends in the lower one: NaNs there
    do not reach the PCM.  (The stereo fast path reads the whole vector -- it saves the arithmetic, not the loads -- so it
    is held to the zero-filled vector only.)"""
    import os
    from vorbispizza_amd import capi
    if os.environ.get("VPZ_NO_SUPPORT", "0") not in ("", "0"):
        pytest.skip("the decoder is told to ignore the declared support (tools/suite_under_switches.sh)")
    n_streams, frames, ends = 5, 40, (128, 400)
    pk, res, posts, counts = stream_major_batch(n_streams, frames, channels, seed=91 + channels, floor=True, interleaved=True,
                                                p_ls=0.1, p_sl=0.3, silent_prob=0.0)
    pk["mapping"] = pk["flags"] & 1
    floors = [(helpers.SHORT_XLIST, 2), (helpers.LONG_XLIST, 2)]
    steps = [(0, 1)] if channels < 4 else [(0, 1), (2, 3)]
    told = [{"coupling": steps, "channel_floor": [f] * channels, "residue_begin": (0, 0), "residue_end": ends} for f in (0, 1)]
    clean = zero_beyond(pk, res, channels, ends[0], ends[1], True)
    # poison only what group mode promises not to read: bins >= 512 of long blocks
    dirty = clean.copy()
    for p in pk:
        if p["flags"] & 1:
            off = int(p["residue_offset"])
            dirty[off: off + channels * 1024].reshape(1024, channels)[512:, :] = np.nan
    with env(VPZ_NO_DUAL=1, VPZ_NO_GROUP=None, VPZ_PAR_MIN_PACKETS=1, VPZ_HOST_THREADS=3):
        ref = run(ctx, pk, clean, posts, counts, n_streams, channels, floors, told, layout=capi.OUT_PLANAR, splits=2)
        got = run(ctx, pk, dirty, posts, counts, n_streams, channels, floors, told, layout=capi.OUT_PLANAR, splits=2)
    same_bits(got, ref, "poisoned upper half")
    assert np.isfinite(got[0]).all()


def test_support_against_the_oracle(ctx, oracle):
    """... and the told decoder against the restated reference on the same (zero-beyond-the-end) packets: 6 channels,
    the configs[3] shape."""
    from vorbispizza_amd import capi
    channels, n_streams, frames, ends = 6, 2, 24, (128, 410)
    pk, res, posts, counts = stream_major_batch(n_streams, frames, channels, seed=606, floor=True, interleaved=True,
                                                p_ls=0.1, p_sl=0.3, silent_prob=0.1)
    pk["mapping"] = pk["flags"] & 1
    res = zero_beyond(pk, res, channels, ends[0], ends[1], True)
    floors = [(helpers.SHORT_XLIST, 2), (helpers.LONG_XLIST, 2)]
    steps = [(0, 1), (2, 3)]
    told = [{"coupling": steps, "channel_floor": [f] * channels, "residue_begin": (0, 0), "residue_end": ends} for f in (0, 1)]
    got = run(ctx, pk, res, posts, counts, n_streams, channels, floors, told, layout=capi.OUT_PLANAR, splits=2)
    cap = frames * 1024 + 64
    for s_ in range(n_streams):
        opk = []
        for i in range(s_ * frames, (s_ + 1) * frames):
            half = 1024 if pk["flags"][i] & 1 else 128
            off = int(pk["residue_offset"][i])
            opk.append({"flags": int(pk["flags"][i]), "granule": -1, "mapping": int(pk["mapping"][i]),
                        "residue": res[off: off + channels * half], "posts": posts[i * channels:(i + 1) * channels],
                        "post_count": counts[i * channels:(i + 1) * channels]})
        ref, _, _ = helpers.oracle_decode(oracle, channels, 256, 2048, opk, floors=floors, mappings=told)
        pcm = got[0][s_ * channels * cap:(s_ + 1) * channels * cap].reshape(channels, cap)[:, :ref.shape[1]]
        assert got[1][s_] == ref.shape[1] and ref.shape[1] > 0
        assert np.abs(pcm - ref).max() <= 1e-5 * max(1.0, float(np.abs(ref).max()))


def test_general_block_sizes_take_the_support_too(ctx):
    """512 / 1024 blocks (the general instantiation): only the arithmetic is skipped there."""
    from vorbispizza_amd import capi
    channels, n_streams, frames = 2, 4, 40
    xl = [helpers.SHORT_XLIST, [x for x in helpers.LONG_XLIST if x < 512]]
    pk, res, posts, counts = stream_major_batch(n_streams, frames, channels, seed=515, floor=True, interleaved=True, size0=512,
                                                size1=1024, xlists=xl, p_ls=0.2, p_sl=0.3)
    pk["mapping"] = pk["flags"] & 1
    out = res.copy()
    for p in pk:
        half = 512 if p["flags"] & 1 else 256
        e = 200 if p["flags"] & 1 else 100
        off = int(p["residue_offset"])
        out[off: off + channels * half].reshape(half, channels)[e:, :] = 0
    floors = [(xl[0], 2), (xl[1], 2)]
    plain = [{"coupling": [(0, 1)], "channel_floor": [0, 0]}, {"coupling": [(0, 1)], "channel_floor": [1, 1]}]
    told = [dict(m, residue_begin=(0, 0), residue_end=(100, 200)) for m in plain]
    ref = run(ctx, pk, out, posts, counts, n_streams, channels, floors, plain, size0=512, size1=1024)
    got = run(ctx, pk, out, posts, counts, n_streams, channels, floors, told, size0=512, size1=1024)
    same_bits(got, ref, "512 / 1024")


def test_inconsistent_support_is_refused(ctx):
    from vorbispizza_amd import Decoder, capi
    floors = [(helpers.SHORT_XLIST, 2)]
    for begin, end in (((0, 700), (128, 600)), ((-1, 0), (128, 600)), ((0, 0), (-5, 600)), ((200, 0), (100, 600))):
        with pytest.raises(capi.SynthError) as e:
            Decoder(ctx, 2, 256, 2048, floors=floors,
                    mappings=[{"coupling": [], "channel_floor": [0, 0], "residue_begin": begin, "residue_end": end}])
        assert e.value.status == capi.E_INVALID_ARG
    # not stated (0) and beyond the block (clamped) are fine
    Decoder(ctx, 2, 256, 2048, floors=floors,
            mappings=[{"coupling": [], "channel_floor": [0, 0], "residue_begin": (0, 0), "residue_end": (0, 5000)}]).close()


def test_real_files_report_their_support_and_decode_the_same_with_it(ctx):
    """The front end fills the support in from the setup header (Residue0.cs:43-44 `_begin`, `_end`); every non-zero
    residue value of the fixtures lies inside it, and the decoder that is told gives the bits of the one that is not."""
    import os
    from vorbispizza_amd import Decoder, capi
    from vorbispizza_amd.front import OggVorbisFile
    for name in ("3test.ogg", "issue6test.ogg", "2test.ogg"):
        f = OggVorbisFile(os.path.join(os.path.dirname(__file__), "golden", name))
        pk, res, posts, counts = f.decode_packets()
        C_ = f.channels
        for p in pk:
            if p["flags"] & capi.PKT_NOT_DECODED:
                continue
            long_ = bool(p["flags"] & 1)
            half = (f.block_size1 if long_ else f.block_size0) // 2
            m = f.mappings[p["mapping"]]
            b, e = m["residue_begin"][long_], m["residue_end"][long_]
            assert 0 <= b < e <= half
            v = res[int(p["residue_offset"]): int(p["residue_offset"]) + C_ * half]
            v = v.reshape(half, C_) if p["flags"] & capi.PKT_INTERLEAVED else v.reshape(C_, half).T
            assert not v[e:].any() and not v[:b].any(), name
        outs = []
        for told in (True, False):
            mp = f.mappings if told else [{k: v for k, v in m.items() if not k.startswith("residue_")} for m in f.mappings]
            dec = Decoder(ctx, C_, f.block_size0, f.block_size1, floors=f.floors, mappings=mp)
            outs.append(dec.synth(pk, res, posts, counts, on_mismatch="ignore")[0])  # (issue6test.ogg's trailing packet)
            dec.close()
        assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32)), name
