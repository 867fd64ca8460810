"""The oracle against the committed golden vectors (tests/golden/*.npz, written by tools/make_golden_fixtures.py):
SURVEY.md 8c's fixture list.  The vectors freeze the oracle as it was when test_spec_crosscheck_cpu.py tied it to the
specification-derived synthesis; a later edit that moves one bit of one sample fails here.  Independent closed forms
are checked on the same data where they exist (float64 cosine sum, coupling truth table, PacketInfo table)."""
import os

import numpy as np
import pytest

import helpers
import spec_synthesis as spec

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def test_imdct_vectors(oracle):
    v = load("imdct_vectors.npz")
    for n in (256, 2048):
        x, y = v["spectra_%d" % n], v["pcm_%d" % n]
        assert np.array_equal(bits(oracle.mdct_reverse(x, n)), bits(y))
        want = spec.imdct(x)  # float64 cosine sum
        assert np.abs(y - want).max() <= 5e-7 * np.abs(want).max()


def test_window_ola_sequence(oracle):
    v = load("window_ola_sequence.npz")
    flags, spectra, pcm = v["flags"], v["spectra"], v["pcm"]
    assert np.array_equal(bits(oracle.synth_stream_planar(2, 256, 2048, flags, spectra)), bits(pcm))
    # every geometry of Mode.cs:30-66 is in the sequence: (LeftStart, LeftEnd, RightStart, RightEnd)
    geo = {tuple(r[2:]) for r in v["packet_info"]}
    assert geo == {(0, 1024, 1024, 2048), (0, 1024, 1472, 1600), (0, 128, 128, 256), (448, 576, 1472, 1600),
                   (448, 576, 1024, 2048)}
    # ... and the emitted length is the sum of SampleCount = RightStart - LeftStart of all but the first packet
    assert pcm.shape[1] == int((v["packet_info"][1:, 4] - v["packet_info"][1:, 2]).sum())
    # the specification-derived synthesis on the same spectra (floor-less: the spectra are used as they are)
    chunks, prev = [], None
    for f in range(len(flags)):
        bf = bool(flags[f] & 1)
        n = 2048 if bf else 256
        y = spec.imdct(spectra[f, :, :n // 2]) * spec.window(n, 256, bf, bool(flags[f] & 2), bool(flags[f] & 4))[None]
        if prev is not None:
            n_prev = prev.shape[1]
            out = np.zeros((2, n_prev // 4 + n // 4))
            t = np.arange(out.shape[1])
            ip, ic = n_prev // 2 + t, t + n_prev // 2 - (n_prev * 3 // 4 - n // 4)
            out[:, ip < n_prev] += prev[:, ip[ip < n_prev]]
            ok = (ic >= 0) & (ic < n)
            out[:, ok] += y[:, ic[ok]]
            chunks.append(out)
        prev = y
    want = np.concatenate(chunks, axis=1)
    # the reference drops the flat part of a first long block that is followed by a block it overlaps shorter
    # (StreamDecoder.cs:679); here the first block is long with a long successor, so the lengths agree
    assert want.shape == pcm.shape and np.abs(want - pcm).max() <= 1e-6


def test_coupling_quadrants(oracle):
    v = load("coupling_quadrants.npz")
    m, a = v["magnitude"], v["angle"]
    for form, km, ka in ((True, "out_magnitude_vector", "out_angle_vector"), (False, "out_magnitude_scalar", "out_angle_scalar")):
        gm, ga = oracle.apply_coupling(m, a, vector_form=form)
        assert np.array_equal(bits(gm), bits(v[km])) and np.array_equal(bits(ga), bits(v[ka]))
    # Vorbis I 4.3.5, value-wise (the two reference branches differ only in the sign of a zero)
    wm, wa = spec.inverse_coupling(m.astype(np.float64), a.astype(np.float64))
    for km, ka in (("out_magnitude_vector", "out_angle_vector"), ("out_magnitude_scalar", "out_angle_scalar")):
        assert np.array_equal(v[km].astype(np.float64), wm) and np.array_equal(v[ka].astype(np.float64), wa)


def test_floor1_of_3test(oracle):
    v = load("floor1_3test_long.npz")
    xlist, mult = [int(x) for x in v["x_list"]], int(v["multiplier"])
    assert len(xlist) == 29
    f = oracle.floor1_init(xlist, mult)
    for r in range(len(v["raw_posts"])):
        fy, fl, cur = oracle.floor1_indices(f, v["raw_posts"][r].astype(np.int32), 29, 1024)
        assert np.array_equal(fy[:29], v["final_y"][r]) and np.array_equal(fl[:29], v["step_flags"][r])
        assert np.array_equal(cur, v["table_index"][r])
        # the specification's floor synthesis gives the same curve
        want = spec.floor1_curve(xlist, mult, v["raw_posts"][r], 1024)
        assert np.abs(v["inverse_db_table"][np.clip(cur, 0, 255)] / want - 1.0).max() < 1e-6
    assert np.array_equal(bits(oracle.inverse_db_table()), bits(v["inverse_db_table"]))


@pytest.mark.parametrize("name", ["1test", "2test", "3test", "issue6test"])
def test_fixture_pcm_heads(oracle, name):
    import __graft_entry__ as ge
    ge.build()
    from vorbispizza_amd.front import OggVorbisFile
    v = load("fixture_pcm_heads.npz")
    f = OggVorbisFile(os.path.join(GOLDEN, name + ".ogg"))
    pk, res, posts, counts = f.decode_packets()
    ref, pos, clipped = helpers.oracle_decode(oracle, f.channels, f.block_size0, f.block_size1,
                                              helpers.packets_for_oracle(f, pk, res, posts, counts),
                                              floors=f.floors, mappings=f.mappings, clip=True)
    meta = v[name + "_meta"]
    assert list(meta[:6]) == [f.channels, f.sample_rate, f.audio_packets, ref.shape[1], pos, int(clipped)]
    assert np.array_equal(bits(ref[:, :4096]), bits(v[name + "_pcm"]))
    mid = int(meta[6])
    assert np.array_equal(bits(ref[:, mid: mid + 2048]), bits(v[name + "_mid"]))
    if name != "1test":  # (1test.ogg is 17 318 samples of digital silence)
        assert np.abs(v[name + "_mid"]).max() > 0.01
