"""The C++ front end against an independent, specification-based stream writer (tests/vorbis_writer.py):
every symbol of every synthetic packet is random, and the writer's own model says what the entropy stage must
recover.  Covers what the reference's fixtures do not (tests/synthetic_streams.py)."""
import numpy as np
import pytest

import helpers
import synthetic_streams as ss
import vorbis_writer as vw


@pytest.fixture(scope="module")
def front():
    import __graft_entry__ as ge
    ge.build()
    from vorbispizza_amd import front as f
    return f


def test_codeword_assignment_matches_spec_example():
    # Vorbis I 3.2.1: lengths 2 4 4 4 4 2 3 3 -> 00 0100 0101 0110 0111 10 110 111
    b = vw.Codebook(1, [2, 4, 4, 4, 4, 2, 3, 3])
    got = ["".join(map(str, b.codes[i])) for i in range(8)]
    assert got == ["00", "0100", "0101", "0110", "0111", "10", "110", "111"]


def test_lookup1_values_and_float_pack():
    assert vw.lookup1_values(81, 4) == 3 and vw.lookup1_values(80, 4) == 2 and vw.lookup1_values(256, 8) == 2
    bits, val = vw.float32_pack(5, 788 - 2, negative=True)
    assert val == np.float32(-1.25) and bits >> 31 == 1


def _check(front, stream, exps, ogg):
    f = front.OggVorbisFile(ogg)
    C = stream.channels
    assert (f.channels, f.sample_rate, f.block_size0, f.block_size1) == (C, stream.rate, stream.bs0, stream.bs1)
    assert f.audio_packets == len(exps) and f.info.bad_crc_pages == 0
    assert f.info.codebook_count == len(stream.books)
    for i, fl in enumerate(stream.floors):
        if fl.type == 0:
            assert f.floors[i] == {"order": fl.order, "rate": fl.rate, "bark_map_size": fl.bark_map_size,
                                   "amp_bits": fl.amp_bits, "amp_ofs": fl.amp_ofs}
        else:
            assert f.floors[i] == (fl.x_list, fl.multiplier)
    assert f.residue_types == [r.type for r in stream.residues]
    for i, mp in enumerate(stream.mappings):
        assert f.mappings[i]["coupling"] == mp.coupling
        assert f.mappings[i]["channel_floor"] == [mp.submap_floor[mp.mux[c]] for c in range(C)]
    pk, res, posts, counts = f.decode_packets()
    assert not (pk["flags"] & helpers.PKT_NOT_DECODED).any()
    for i, e in enumerate(exps):
        fl = int(pk["flags"][i])
        assert (fl & 1) == e["blockflag"], i
        if e["blockflag"]:
            assert ((fl >> 1) & 1, (fl >> 2) & 1) == (e["prev"], e["next"]), i
        assert pk["mapping"][i] == e["mapping"]
        half = (stream.bs1 if e["blockflag"] else stream.bs0) // 2
        off = int(pk["residue_offset"][i])
        got = res[off: off + C * half]
        got = got.reshape(half, C).T if fl & helpers.PKT_INTERLEAVED else got.reshape(C, half)
        np.testing.assert_array_equal(counts[i * C:(i + 1) * C], e["post_count"], err_msg="packet %d" % i)
        np.testing.assert_array_equal(posts[i * C:(i + 1) * C], e["posts"], err_msg="packet %d" % i)
        np.testing.assert_array_equal(got, e["residue"], err_msg="packet %d" % i)
        if f.floor0_data is not None:
            amp, coeff = f.floor0_data
            for c in range(C):
                if e["f0_coeff"][c] is not None:
                    assert amp[i * C + c] == e["f0_amp"][c]
                    k = len(e["f0_coeff"][c])
                    np.testing.assert_array_equal(coeff[i * C + c, :k], e["f0_coeff"][c])
    assert pk["flags"][-1] & helpers.PKT_EOS
    return f


@pytest.mark.parametrize("name", sorted(ss.ALL))
def test_front_end_recovers_what_the_writer_encoded(front, name):
    stream, rng = ss.ALL[name]()
    ogg, exps = stream.build(rng, 24)
    _check(front, stream, exps, ogg)


def test_packets_continued_across_pages(front):
    # 2 lacing values per page force every packet over several pages (continued-packet flag, granule -1
    # on pages that complete nothing)
    stream, rng = ss.stereo_coupled_res2(seed=11)
    ogg, exps = stream.build(rng, 10, packets_per_page=4, max_segments=2)
    continued, pos = 0, 0
    while pos < len(ogg):
        assert ogg[pos:pos + 4] == b"OggS"
        nseg = ogg[pos + 26]
        continued += ogg[pos + 5] & 1
        pos += 27 + nseg + sum(ogg[pos + 27: pos + 27 + nseg])
    assert continued >= 5
    _check(front, stream, exps, ogg)


def test_granules_follow_the_page_that_completes_the_packet(front):
    stream, rng = ss.mono_floor1_res1(seed=5)
    ogg, exps = stream.build(rng, 12, packets_per_page=1)
    f = front.OggVorbisFile(ogg)
    pk, _, _, _ = f.decode_packets()
    total, want = 0, []
    for i, e in enumerate(exps):
        if i:
            p = exps[i - 1]["blockflag"]
            total += ((stream.bs1 if p else stream.bs0) + (stream.bs1 if e["blockflag"] else stream.bs0)) // 4
        want.append(total)
    assert list(pk["granule"]) == want and f.last_granule == want[-1]


@pytest.mark.parametrize("seed", range(40, 64))
def test_random_setups(front, seed):
    stream, rng = ss.random_stream(seed)
    ogg, exps = stream.build(rng, 14, packets_per_page=int(rng.integers(1, 6)))
    _check(front, stream, exps, ogg)
