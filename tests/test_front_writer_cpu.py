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


# ---- malformed setup headers the reference rejects with a managed exception (round-1 advisor findings) ----
def test_mux_equal_to_submap_count_is_rejected_at_open(front):
    # Mapping.cs:60 lets mux == submapCount through, :89-93 then indexes _submapFloor out of range and the
    # constructor dies; the front end must refuse the header instead of reading past its vector
    stream, rng = ss.three_channels_two_submaps()
    for mp in stream.mappings:
        mp.mux = [0, 0, 2]
    # the writer's own packet model cannot use submap 2: only the headers matter here
    ogg = vw.ogg_mux(stream.headers(), [0, 0, 0])
    with pytest.raises(front.FrontError) as e:
        front.OggVorbisFile(ogg)
    assert "mux" in str(e.value)


def test_residue_value_book_without_dimensions_is_rejected_at_open(front):
    # dimensions == 0 with lookup type 2 parses (0 multiplicands); Residue0.cs:213 would divide by it
    stream, rng = ss.mono_floor1_res1()
    res = stream.residues[0]
    victim = next(b for row in res.books for b in row if b is not None)
    old = stream.books[victim]
    stream.books[victim] = vw.Codebook(0, old.lengths, 2, minv=vw.float32_pack(0, 788), delta=vw.float32_pack(1, 788),
                                       value_bits=4, mults=[], sparse=old.sparse, ordered=old.ordered)
    ogg = vw.ogg_mux(stream.headers(), [0, 0, 0])
    with pytest.raises(front.FrontError) as e:
        front.OggVorbisFile(ogg)
    assert "dimensions" in str(e.value)


def test_a_packet_that_throws_costs_only_itself(front):
    # "Unused mode index." (StreamDecoder.cs:727-730): three modes need 2 mode bits, index 3 does not exist.
    # The reference's exception consumes that packet and nothing else.
    stream, rng = ss.mono_floor1_res1(seed=9)
    ogg, exps = stream.build(rng, 12, packets_per_page=1)
    # find the page of audio packet 5 (3 header packets sit on the first pages) and set its mode bits to 3
    pages, pos = [], 0
    while pos < len(ogg):
        nseg = ogg[pos + 26]
        body = sum(ogg[pos + 27: pos + 27 + nseg])
        pages.append((pos, 27 + nseg, body))
        pos += 27 + nseg + body
    good = front.OggVorbisFile(ogg)
    pk_good, res_good, posts_good, counts_good = good.decode_packets()
    # audio packets are one per page here; the header packets may share pages, so count from the end
    ppos, hdr, body = pages[len(pages) - 12 + 5]
    raw = bytearray(ogg)
    raw[ppos + hdr] |= 0b110        # bit 0 = packet type (0), bits 1..2 = mode index -> 3
    vw_crc = vw._crc
    raw[ppos + 22: ppos + 26] = b"\0\0\0\0"
    c = vw_crc(bytes(raw[ppos: ppos + hdr + body]))
    raw[ppos + 22: ppos + 26] = int(c).to_bytes(4, "little")
    bad = front.OggVorbisFile(bytes(raw))
    pk, res, posts, counts = bad.decode_packets()
    assert bad.decode_failures() == (1, 5) and "mode" in bad.last_error().lower()
    assert pk["flags"][5] == helpers.PKT_NOT_DECODED
    keep = np.arange(12) != 5
    np.testing.assert_array_equal(pk["flags"][keep], pk_good["flags"][keep])
    C = stream.channels
    np.testing.assert_array_equal(posts.reshape(12, -1)[keep], posts_good.reshape(12, -1)[keep])
    for i in np.nonzero(keep)[0]:
        half = (stream.bs1 if pk["flags"][i] & 1 else stream.bs0) // 2
        a, b = int(pk["residue_offset"][i]), int(pk_good["residue_offset"][i])
        np.testing.assert_array_equal(res[a: a + C * half], res_good[b: b + C * half])
