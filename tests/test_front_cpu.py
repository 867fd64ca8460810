"""The C++ CPU front end (Ogg demux + Vorbis setup + entropy decode) on the reference's
own fixtures (tests/golden/*.ogg are copies of /root/reference/TestFiles/*.ogg -- data, not code).
Known answers: channels, rates, block sizes, packet counts, floor / residue /
coupling layout, and the decoded sample count, which must equal the stream's granule span."""
import hashlib
import os
import time

import numpy as np
import pytest

import helpers

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# file -> (sha256[:16], channels, audio packets, short, long, samples, floor posts, residue types, coupling)
FACTS = {
    "1test.ogg": ("9e11f3008b507327", 1, 25, 8, 17, 17318, [(19, 2), (29, 2)], [1, 1], []),
    "2test.ogg": ("ec0b573cb2f09dfa", 1, 310, 1, 309, 315790, [(6, 4), (29, 2)], [1, 1], []),
    "3test.ogg": ("9da311d45f63d0b1", 2, 366, 95, 271, 288094, [(19, 2), (29, 2)], [2, 2], [(0, 1)]),
    # last granule 548223, first audio page starts 63 samples in; the trailing empty packet (page 17) is
    # the 606th packet and hits the reference's OverlapBuffers exception (StreamDecoder.cs:777-778)
    "issue6test.ogg": ("3b863d0503d22b6e", 2, 606, 80, 526, 548160, [(13, 2), (29, 2)], [2, 2], [(0, 1)]),
}


@pytest.fixture(scope="module")
def front():
    import __graft_entry__ as ge
    ge.build()
    from vorbispizza_amd import front as f
    return f


@pytest.mark.parametrize("name", sorted(FACTS))
def test_fixture_header_facts(front, name):
    sha, ch, n_pk, n_short, n_long, samples, floors, res_types, coupling = FACTS[name]
    path = os.path.join(GOLDEN, name)
    assert hashlib.sha256(open(path, "rb").read()).hexdigest()[:16] == sha
    f = front.OggVorbisFile(path)
    assert (f.channels, f.sample_rate, f.block_size0, f.block_size1) == (ch, 44100, 256, 2048)
    assert f.audio_packets == n_pk
    assert [(len(x), m) for x, m in f.floors] == floors
    assert f.residue_types == res_types
    assert all(m["coupling"] == coupling for m in f.mappings) and len(f.mappings) == 2
    assert f.info.bad_crc_pages == 0
    pk, res, posts, counts = f.decode_packets()
    longs = int((pk["flags"] & 1).sum())
    assert (n_pk - longs, longs) == (n_short, n_long)
    assert not (pk["flags"] & helpers.PKT_NOT_DECODED).any()
    assert int((pk["flags"] & helpers.PKT_EOS != 0).sum()) == 1 and pk["flags"][-1] & helpers.PKT_EOS
    # Residue2 streams hand the interleaved vector over, residue 1 streams planar
    assert bool((pk["flags"] & helpers.PKT_INTERLEAVED).any()) == (res_types[0] == 2)
    # floor posts stay inside their ranges
    assert posts.min() >= 0 and counts.max() <= 29


@pytest.mark.parametrize("name", sorted(FACTS))
def test_decoded_sample_count_equals_granule_span(front, oracle, name):
    samples = FACTS[name][5]
    f = front.OggVorbisFile(os.path.join(GOLDEN, name))
    pk, res, posts, counts = f.decode_packets()
    opk = helpers.packets_for_oracle(f, pk, res, posts, counts)
    pcm, pos, _ = helpers.oracle_decode(oracle, f.channels, f.block_size0, f.block_size1, opk,
                                        floors=f.floors, mappings=f.mappings)
    assert pcm.shape == (f.channels, samples)
    assert pos == samples
    assert helpers.oracle_decode.last_mismatches == (1 if name == "issue6test.ogg" else 0)
    assert np.isfinite(pcm).all()
    assert 0.05 < np.abs(pcm).max() < 1.5          # real audio, not garbage from a bad Huffman decode
    # decoded audio is smooth: a wrong codeword anywhere shows up as broadband noise
    d = np.diff(pcm, axis=1)
    assert np.sqrt((d ** 2).mean()) < 0.6 * np.sqrt((pcm ** 2).mean()) + 1e-3


def test_truncated_and_corrupt_input(front):
    data = open(os.path.join(GOLDEN, "1test.ogg"), "rb").read()
    with pytest.raises(front.FrontError):
        front.OggVorbisFile(data[:40])                       # no complete header packets
    bad = bytearray(data)
    bad[-100] ^= 0x55                                        # CRC failure on the audio page -> page dropped
    f = front.OggVorbisFile(bytes(bad))
    assert f.info.bad_crc_pages >= 1 and f.audio_packets < 25
    with pytest.raises(front.FrontError):
        front.OggVorbisFile(b"not an ogg file at all" * 10)


def test_block_sizes_outside_the_format_are_refused_at_open(front):
    # StreamDecoder.cs:226 accepts any two exponents; Vorbis I allows 64 ... 8192, short <= long, and the synthesis library
    # refuses the rest at vpz_decoder_create -- so does the front end, before a block of one sample reaches its residue decode
    import synthetic_streams as ss
    import vorbis_writer as vw
    for logs in ((0, 8), (5, 11), (9, 8), (8, 14)):
        stream, _ = ss.mono_floor1_res1()
        stream.bs_logs = logs
        with pytest.raises(front.FrontError):
            front.OggVorbisFile(bytes(vw.ogg_mux(stream.headers(), [0, 0, 0])))
    stream, _ = ss.mono_floor1_res1()
    front.OggVorbisFile(bytes(vw.ogg_mux(stream.headers(), [0, 0, 0])))  # (the unchanged headers open)


def test_an_ordered_codebook_that_runs_out_of_bits_is_refused_not_counted_to_the_end_of_int(front):
    # Codebook.cs:60-66 with a packet that ends inside the length list: every count reads as zero, `len` goes up for ever
    # (found by the sanitizer campaign as a signed overflow after 2^31 rounds)
    import time
    import synthetic_streams as ss
    import vorbis_writer as vw
    stream, _ = ss.mono_floor1_res1()
    ident, comment, _ = stream.headers()
    w = vw.BitWriter()
    for c in b"\x05vorbis":
        w.write(c, 8)
    w.write(0, 8)            # one codebook
    w.write(0x564342, 24)
    w.write(1, 16)           # dimensions
    w.write(100, 24)         # entries
    w.write(1, 1)            # ordered
    w.write(0, 5)            # first length 1 -- and the packet ends here
    t0 = time.time()
    with pytest.raises(front.FrontError):
        front.OggVorbisFile(bytes(vw.ogg_mux([ident, comment, w.bytes()], [0, 0, 0])))
    assert time.time() - t0 < 1.0


def test_a_comment_header_over_many_pages_changes_nothing(front):
    # album art: a 300 KB comment packet laced over five pages in front of the setup header (PacketProvider.cs:427-560)
    import struct
    import synthetic_streams as ss
    stream, rng = ss.ALL["stereo_coupled_res2"]()
    plain, _ = stream.build(rng, 20)
    stream, rng = ss.ALL["stereo_coupled_res2"]()
    headers = stream.headers
    big = b"\x03vorbis" + struct.pack("<I", 4) + b"test" + struct.pack("<II", 1, 300000) + b"A" * 300000 + b"\x01"
    stream.headers = lambda: [headers()[0], big, headers()[2]]
    padded, _ = stream.build(rng, 20)
    assert len(padded) > len(plain) + 300000
    a, b = front.OggVorbisFile(bytes(plain)), front.OggVorbisFile(bytes(padded))
    pa, pb = a.decode_packets(), b.decode_packets()
    assert (a.audio_packets, int(a.total_samples)) == (b.audio_packets, int(b.total_samples)) == (20, 13184)
    assert np.array_equal(pa[0]["flags"], pb[0]["flags"]) and np.array_equal(pa[1], pb[1]) and np.array_equal(pa[2], pb[2])


def test_decode_into_shared_batch_buffers_from_threads(front):
    """Several handles decoded from several threads straight into slices of one batch buffer give what
    decode_packets gives for each stream alone (stream ids and residue offsets rebased)."""
    from concurrent.futures import ThreadPoolExecutor
    from vorbispizza_amd import capi
    data = open(os.path.join(GOLDEN, "3test.ogg"), "rb").read()
    ref_f = front.OggVorbisFile(data)
    pk0, res0, posts0, counts0 = ref_f.decode_packets()
    n, C_, rf, copies = ref_f.audio_packets, ref_f.channels, ref_f.info.residue_floats, 6
    pk = capi.make_packets(n * copies)
    res = np.zeros(rf * copies, dtype=np.float32)
    posts = np.zeros((n * copies * C_, 64), dtype=np.int16)
    counts = np.zeros(n * copies * C_, dtype=np.uint8)

    def job(k):
        f = front.OggVorbisFile(data)
        f.decode_into(pk[k * n:(k + 1) * n], res[k * rf:(k + 1) * rf], posts[k * n * C_:(k + 1) * n * C_],
                      counts[k * n * C_:(k + 1) * n * C_], stream_id=k, residue_base=k * rf)
        f.close()

    with ThreadPoolExecutor(max_workers=4) as pool:
        list(pool.map(job, range(copies)))
    for k in range(copies):
        sl = pk[k * n:(k + 1) * n]
        assert (sl["stream"] == k).all()
        assert np.array_equal(sl["residue_offset"], pk0["residue_offset"] + k * rf)
        assert np.array_equal(sl["flags"], pk0["flags"]) and np.array_equal(sl["granule"], pk0["granule"])
        assert np.array_equal(res[k * rf:(k + 1) * rf], res0)
        assert np.array_equal(posts[k * n * C_:(k + 1) * n * C_], posts0)
        assert np.array_equal(counts[k * n * C_:(k + 1) * n * C_], counts0)


@pytest.mark.parametrize("name", sorted(FACTS))
def test_seek_table_matches_the_packet_geometry(front, name):
    """PacketProvider.SeekTo with preRoll 1: positions are counted from the per-packet SampleCount (the first
    audio packet only primes the overlap); the answer is the packet BEFORE the one whose span holds the target."""
    f = front.OggVorbisFile(os.path.join(GOLDEN, name))
    pk, _, _, _ = f.decode_packets()
    counts = []
    for fl in pk["flags"]:
        if fl & helpers.PKT_NOT_DECODED:
            counts.append(0)
            continue
        bf, pf, nf = fl & 1, bool(fl & 2), bool(fl & 4)
        size = 2048 if bf else 256
        left = 0 if (not bf or pf) else (size - 256) // 4
        right = size // 2 if (not bf or nf) else (size * 3 - 256) // 4
        counts.append(right - left)
    cum = np.concatenate([[0], np.cumsum(counts[1:])])
    assert f.total_samples == min(int(cum[-1]), f.last_granule)
    rng = np.random.default_rng(1)
    for g in [0, 1, int(cum[1]) - 1, int(cum[1]), int(cum[-1]) - 1, int(cum[-1])] + [int(v) for v in rng.integers(0, cum[-1], 50)]:
        first, roll = f.seek(g)
        k = first + 1
        assert 0 <= first < len(pk) - 1
        assert cum[k - 1] <= g and (g < cum[k] or (g == cum[-1] and k == len(pk) - 1))
        assert roll == g - cum[k - 1]
    with pytest.raises(front.FrontError):
        f.seek(int(cum[-1]) + 1)
    with pytest.raises(front.FrontError):
        f.seek(-1)


# ---- chained streams and resync ----
def _pages(ogg):
    out, pos = [], 0
    while pos < len(ogg):
        assert ogg[pos:pos + 4] == b"OggS"
        nseg = ogg[pos + 26]
        body = sum(ogg[pos + 27: pos + 27 + nseg])
        out.append((pos, 27 + nseg + body))
        pos += 27 + nseg + body
    return out


def test_chained_container_opens_stream_by_stream(front):
    """VorbisReader.FindNextStream / SwitchStreams (VorbisReader.cs:191-217): a chained file is a sequence of logical
    streams, each with its own setup headers; stream i of the chain decodes exactly like the file it came from."""
    a = open(os.path.join(GOLDEN, "3test.ogg"), "rb").read()
    b = open(os.path.join(GOLDEN, "1test.ogg"), "rb").read()
    chain = a + b + a
    for idx, single in enumerate((a, b, a)):
        want = front.OggVorbisFile(single)
        got = front.OggVorbisFile(chain, stream_index=idx)
        assert (got.channels, got.sample_rate, got.audio_packets, got.last_granule) == \
               (want.channels, want.sample_rate, want.audio_packets, want.last_granule)
        assert got.info.stream_serial == want.info.stream_serial
        for x, y in zip(got.decode_packets(), want.decode_packets()):
            assert np.array_equal(x, y)
    with pytest.raises(front.FrontError):
        front.OggVorbisFile(chain, stream_index=3)


def test_a_dropped_page_marks_the_next_packet_as_resync(front):
    """A page with a bad checksum is skipped (PageReaderBase.cs:286-361); the page found after it is a resync page and
    its packets carry IsResync (StreamDecoder.cs:718-722 clears _hasPosition for them).  A gap in the page sequence
    numbers does the same (StreamPageReader.cs:87-96)."""
    from vorbispizza_amd import capi
    ogg = bytearray(open(os.path.join(GOLDEN, "3test.ogg"), "rb").read())
    clean = front.OggVorbisFile(bytes(ogg))
    pk0 = clean.decode_packets()[0]
    assert not (pk0["flags"] & capi.PKT_RESYNC).any()
    pages = _pages(ogg)
    victim = len(pages) // 2
    pos, length = pages[victim]
    ogg[pos + length - 1] ^= 0x55                      # body byte: the checksum no longer matches
    f = front.OggVorbisFile(bytes(ogg))
    assert f.info.bad_crc_pages >= 1 and f.audio_packets < clean.audio_packets
    pk = f.decode_packets()[0]
    marked = np.nonzero(pk["flags"] & capi.PKT_RESYNC)[0]
    assert len(marked) >= 1
    # every packet completed on the page after the dropped one is flagged, nothing else
    nseg = ogg[pages[victim + 1][0] + 26]
    segs = ogg[pages[victim + 1][0] + 27: pages[victim + 1][0] + 27 + nseg]
    completed = sum(1 for v in segs if v < 255)
    assert 1 <= len(marked) <= completed and (np.diff(marked) == 1).all()
    # cutting the page out altogether leaves a sequence gap instead of skipped bytes: same flags
    cut = bytes(ogg[:pos]) + bytes(ogg[pos + length:])
    g = front.OggVorbisFile(cut)
    assert g.info.bad_crc_pages == 0
    assert np.array_equal(g.decode_packets()[0]["flags"], pk["flags"])


def test_decode_many_equals_one_stream_at_a_time():
    """vpzh_decode_many (a library of files entropy-decoded on the host library's own threads, straight into batch arrays)
    writes for every stream exactly what vpzh_decode_all writes for it alone -- at its slices, with its stream id and with
    residue offsets relative to the origin the later synth call will see."""
    from vorbispizza_amd import capi, front
    datas = {name: open(os.path.join(GOLDEN, name), "rb").read() for name in ("3test.ogg", "issue6test.ogg")}
    singles = {}
    for name, raw in datas.items():
        f = front.OggVorbisFile(raw)
        singles[name] = (f.audio_packets, f.channels, f.info.residue_floats) + tuple(f.decode_packets())
    order = ["3test.ogg", "issue6test.ogg", "3test.ogg", "3test.ogg", "issue6test.ogg"]
    C_ = 2
    pbase, rbase, np_, nr = [], [], 0, 0
    lead_p, lead_r = 7, 1000  # (the batch arrays hold other streams in front: the call writes at the bases it is given)
    np_, nr = lead_p, lead_r
    for name in order:
        n, _, rf = singles[name][:3]
        pbase.append(np_)
        rbase.append(nr)
        np_ += n
        nr += rf
    pk = capi.make_packets(np_)
    res = np.full(nr, 7.0, dtype=np.float32)
    posts = np.full((np_ * C_, 64), -3, dtype=np.int16)
    counts = np.full(np_ * C_, 200, dtype=np.uint8)
    arrays = [np.frombuffer(datas[name], dtype=np.uint8) for name in order]
    done = np.zeros(len(order), dtype=np.int32)
    failed = front.decode_many(arrays, pbase, rbase, pk, res, posts, counts, threads=3, stream_id0=10, residue_origin=lead_r,
                               done=done)
    assert failed == 0 and (done == 1).all()  # (vpzh_decode_many_progress: every stream reported complete)
    for k, name in enumerate(order):
        n, _, rf, spk, sres, sposts, scounts = singles[name]
        a = pk[pbase[k]: pbase[k] + n]
        assert (a["stream"] == 10 + k).all()
        for field in ("flags", "mapping", "granule"):
            assert np.array_equal(a[field], spk[field]), (name, field)
        assert np.array_equal(a["residue_offset"], spk["residue_offset"] + (rbase[k] - lead_r))
        assert np.array_equal(res[rbase[k]: rbase[k] + rf], sres)
        assert np.array_equal(posts[pbase[k] * C_: (pbase[k] + n) * C_], sposts)
        assert np.array_equal(counts[pbase[k] * C_: (pbase[k] + n) * C_], scounts)
    assert (res[:lead_r] == 7.0).all() and (counts[: lead_p * C_] == 200).all()  # nothing in front was touched
    # a container that does not fit the room its slice has is refused, and nothing of it is written
    with pytest.raises(front.FrontError):
        front.decode_many(arrays[:1], pbase[:1], rbase[:1], pk, res, posts, counts, threads=1,
                          packet_room=[singles[order[0]][0] - 1], residue_room=[singles[order[0]][2]])
    # ... and says so in its progress entry, while the streams next to it are decoded
    done = np.zeros(2, dtype=np.int32)
    with pytest.raises(front.FrontError):
        front.decode_many(arrays[:2], pbase[:2], rbase[:2], pk, res, posts, counts, threads=2, done=done,
                          packet_room=[singles[order[0]][0] - 1, singles[order[1]][0]],
                          residue_room=[singles[order[0]][2], singles[order[1]][2]])
    assert list(done) == [-1, 1]


def test_decode_many_refuses_a_container_with_another_channel_count():
    """The batch's post records are laid out for ONE channel count: a 6-channel file that slips into a stereo batch -- with few
    enough packets and residue to pass both room checks -- would write its records past the caller's arrays (untrusted file
    contents).  It is refused before anything is written; the streams next to it are decoded."""
    import synthetic_streams as ss
    from vorbispizza_amd import capi, front
    stereo = open(os.path.join(GOLDEN, "3test.ogg"), "rb").read()
    f = front.OggVorbisFile(stereo)
    n, rf = f.audio_packets, f.info.residue_floats
    stream, rng = ss.ALL["six_channels_51"]()
    six, _ = stream.build(rng, 6)
    g = front.OggVorbisFile(six)
    assert g.channels == 6 and g.audio_packets <= n and g.info.residue_floats <= rf  # (it WOULD pass the room checks)
    C_ = 2
    pk = capi.make_packets(3 * n)
    res = np.full(3 * rf, 7.0, dtype=np.float32)
    guard = 4096
    posts = np.full((3 * n * C_ + guard, 64), -3, dtype=np.int16)
    counts = np.full(3 * n * C_ + guard, 200, dtype=np.uint8)
    arrays = [np.frombuffer(stereo, dtype=np.uint8), np.frombuffer(bytes(six), dtype=np.uint8), np.frombuffer(stereo, dtype=np.uint8)]
    done = np.zeros(3, dtype=np.int32)
    with pytest.raises(front.FrontError):
        front.decode_many(arrays, [0, n, 2 * n], [0, rf, 2 * rf], pk, res, posts[: 3 * n * C_], counts[: 3 * n * C_], threads=2,
                          done=done, channels=C_)
    assert list(done) == [1, -1, 1]
    assert (res[rf: 2 * rf] == 7.0).all() and (counts[n * C_: 2 * n * C_] == 200).all() and (posts[n * C_: 2 * n * C_] == -3).all()
    assert (counts[3 * n * C_:] == 200).all() and (posts[3 * n * C_:] == -3).all()  # nothing beyond the arrays
    assert (pk["stream"][:n] == 0).all() and (pk["stream"][2 * n:] == 2).all()
    # a channel count out of range is an argument error
    with pytest.raises(front.FrontError):
        front.decode_many(arrays[:1], [0], [0], pk, res, posts, counts, threads=1, channels=0)
    assert front.lib().vpzh_default_threads() >= 1


def test_decode_many_progress_lets_another_thread_take_finished_streams():
    """The pipeline bench.py's end-to-end leg runs: one thread inside vpzh_decode_many_progress, another one watching the
    flags and picking streams up in order as they complete -- what it picks up is final."""
    import threading
    from vorbispizza_amd import capi, front
    raw = np.frombuffer(open(os.path.join(GOLDEN, "1test.ogg"), "rb").read(), dtype=np.uint8)
    f = front.OggVorbisFile(raw.tobytes())
    n, C_, rf = f.audio_packets, f.channels, f.info.residue_floats
    want = f.decode_packets()
    streams = 12
    pk = capi.make_packets(n * streams)
    res = np.zeros(rf * streams, dtype=np.float32)
    posts = np.zeros((n * streams * C_, 64), dtype=np.int16)
    counts = np.zeros(n * streams * C_, dtype=np.uint8)
    done = np.zeros(streams, dtype=np.int32)
    t = threading.Thread(target=front.decode_many, args=([raw] * streams, [k * n for k in range(streams)], [k * rf for k in range(streams)],
                                                         pk, res, posts, counts), kwargs=dict(threads=3, done=done))
    t.start()
    seen = []
    for k in range(streams):
        while done[k] == 0:
            time.sleep(1e-4)
        assert done[k] == 1
        seen.append(bool(np.array_equal(res[k * rf:(k + 1) * rf], want[1]) and
                         np.array_equal(counts[k * n * C_:(k + 1) * n * C_], want[3])))
    t.join()
    assert all(seen)


def _decode_digest(names, rounds):
    from vorbispizza_amd import front
    h = hashlib.sha256()
    for _ in range(rounds):
        for name in names:
            f = front.OggVorbisFile(os.path.join(GOLDEN, name))
            pk, res, posts, counts = f.decode_packets()
            for a in (pk, res, posts, counts):
                h.update(np.ascontiguousarray(a).tobytes())
            h.update(repr((f.channels, f.block_size0, f.block_size1, f.floors, f.mappings)).encode())
    return h.hexdigest()


def test_setup_cache_hits_decode_exactly_like_a_fresh_unpack():
    """A stream whose identification + setup headers were seen before copies their unpacked form (codebook tables, floors,
    residues, mappings, modes) instead of unpacking them again: the second and third opening of every fixture (hits) give the
    bytes a process with the cache switched off (VPZH_NO_SETUP_CACHE=1) gives."""
    import subprocess
    import sys
    names = ["1test.ogg", "2test.ogg", "3test.ogg", "issue6test.ogg"]
    here = _decode_digest(names, 3)
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_front_cpu as t; "
            "print(t._decode_digest(%r, 3))" % (os.path.dirname(GOLDEN), os.path.dirname(os.path.dirname(GOLDEN)), names))
    env = dict(os.environ, VPZH_NO_SETUP_CACHE="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip().splitlines()[-1] == here
