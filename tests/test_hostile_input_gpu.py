"""Containers whose AUDIO pages were damaged after encoding (checksums made valid again, so that the damage reaches the
bit-level decode): they open, their packets decode to whatever the bits say -- floor posts and residue values no encoder
writes -- and the device path must take that like any other input: no fault, every stream's outcome its own, and the same
PCM bit for bit whichever way the job is partitioned and through the plain ABI one file at a time.  (What the reference
does with such packets is its own table-index exceptions, Floor1.cs:407-473: there is nothing to compare values with --
the properties are determinism and isolation.)"""
import os
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _pages(raw):
    """(start, header length, body length, packets completed) of every page"""
    out, pos = [], 0
    while pos + 27 <= len(raw) and raw[pos:pos + 4] == b"OggS":
        nseg = raw[pos + 26]
        lac = raw[pos + 27: pos + 27 + nseg]
        out.append((pos, 27 + nseg, sum(lac), sum(1 for v in lac if v < 255)))
        pos += 27 + nseg + sum(lac)
    return out


def damage_audio(raw, seed, hits):
    import vorbis_writer as vw
    rng = np.random.default_rng(seed)
    pages, done, first_audio = _pages(raw), 0, None
    for i, (_, _, _, completed) in enumerate(pages):
        if done >= 3:
            first_audio = i
            break
        done += completed
    assert first_audio is not None
    data = bytearray(raw)
    touched = set()
    for _ in range(hits):
        i = int(rng.integers(first_audio, len(pages)))
        start, hdr, body, _ = pages[i]
        if body == 0:
            continue
        at = start + hdr + int(rng.integers(0, body))
        kind = int(rng.integers(0, 3))
        if kind == 0:
            data[at] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:
            data[at] = int(rng.integers(0, 256))
        else:
            n = min(int(rng.integers(1, 24)), start + hdr + body - at)
            data[at:at + n] = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        touched.add(i)
    for i in touched:
        start, hdr, body, _ = pages[i]
        page = bytes(data[start:start + 22]) + b"\0\0\0\0" + bytes(data[start + 26:start + hdr + body])
        data[start + 22:start + 26] = struct.pack("<I", vw._crc(page))
    return bytes(data)


@pytest.fixture(scope="module")
def ctx():
    from vorbispizza_amd import Context
    c = Context(0)
    yield c
    c.close()


def hostile_library():
    import synthetic_streams as ss
    rounds = int(os.environ.get("VPZ_HOSTILE_ROUNDS", "1"))  # (a longer campaign by hand: more seeds of the same recipe)
    raws = []
    for name in ("1test.ogg", "2test.ogg", "3test.ogg"):
        clean = open(os.path.join(GOLDEN, name), "rb").read()
        raws.append(clean)
        for r in range(rounds):
            for seed, hits in ((1, 1), (2, 4), (3, 16), (4, 64)):
                raws.append(damage_audio(clean, seed + 100 * r, hits * (1 + r % 4)))
    for name in ("stereo_floor0", "six_channels_51", "three_channels_two_submaps", "mono_floor1_res1", "ten_channels"):
        stream, rng = ss.ALL[name]()
        ogg, _ = stream.build(rng, 40)
        raws.append(bytes(ogg))
        for r in range(rounds):
            for seed, hits in ((5, 2), (6, 12)):
                raws.append(damage_audio(bytes(ogg), seed + 100 * r, hits * (1 + r % 4)))
    return raws


def test_damaged_audio_pages_decode_the_same_way_however_the_job_is_cut(ctx):
    from test_multi_gpu import run_dispatcher, single_stream_pcm
    raws = hostile_library()
    runs = [run_dispatcher([0], raws, host_threads=3, streams_per_call=5, capacity_slack=int(os.environ.get("VPZ_HOSTILE_SLACK", "2048"))),
            run_dispatcher([0, 0, 0], raws, host_threads=6, streams_per_call=2),
            run_dispatcher([0, 0], raws, host_threads=4, streams_per_call=16, float_residue=1)]
    pcm0, offs, res0, _, infos = runs[0]
    assert (res0["status"] == 0).all()  # (they all opened: the damage is in the audio pages; bad packets are skipped ones)
    assert int(res0["samples"].sum()) > 0
    print("%d containers, %d samples, %d skipped packets" % (len(raws), int(res0["samples"].sum()), int(res0["skipped_packets"].sum())))
    for pcm, _, res, _, _ in runs[1:]:
        for field in ("status", "samples", "packets", "skipped_packets", "channels"):
            assert np.array_equal(res[field], res0[field]), field
        assert np.array_equal(pcm.view(np.uint32), pcm0.view(np.uint32))
    # ... and the plain ABI, one file at a time, gives those very samples
    for k, r in enumerate(raws):
        ref = single_stream_pcm(ctx, r)
        C_ = infos[r][0]
        assert res0["samples"][k] == ref.shape[0], k
        got = pcm0[offs[k]: offs[k] + ref.shape[0] * C_].reshape(-1, C_)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), k
    # 16-bit PCM (`(int)(x * 32768f)` clamped, AssetTest.cs:131-132) of values no encoder produces: the same in both again
    pcm16, offs16, res16, _, _ = run_dispatcher([0, 0], raws, s16=True, host_threads=4, streams_per_call=6)
    assert np.array_equal(res16["samples"], res0["samples"])
    for k, r in enumerate(raws[:: max(1, len(raws) // 40)]):
        k = k * max(1, len(raws) // 40)
        ref = single_stream_pcm(ctx, r, s16=True)
        C_ = infos[r][0]
        got = pcm16[offs16[k]: offs16[k] + ref.shape[0] * C_].reshape(-1, C_)
        assert np.array_equal(got, ref), k
    # the context is as good as before: an undamaged file decodes to what it always did
    clean = open(os.path.join(GOLDEN, "3test.ogg"), "rb").read()
    again = run_dispatcher([0], [clean], host_threads=1)
    ref = single_stream_pcm(ctx, clean)
    assert np.array_equal(again[0][: ref.size].view(np.uint32), ref.reshape(-1).view(np.uint32))


def _read_all(reader_cls, ctx, raw, batch, buf_samples, synth_error):
    """the ReadSamples loop of a host that goes on after a throwing Read (tests/test_reader_gpu.py)"""
    rd = reader_cls(ctx, raw, clip_samples=False, batch_packets=batch)
    C_ = rd.Channels
    buf = np.zeros(buf_samples * C_, dtype=np.float32)
    chunks, thrown = [], 0
    while True:
        try:
            n = rd.ReadSamples(buf)
        except synth_error:
            thrown += 1
            assert thrown < 100000
            continue
        if n == 0:
            break
        chunks.append(buf[:n * C_].copy())
    rd.Dispose()
    got = np.concatenate(chunks).reshape(-1, C_) if chunks else np.zeros((0, C_), dtype=np.float32)
    return got, thrown


def test_the_reader_gives_damaged_streams_the_samples_of_the_batch_decode(ctx):
    """VorbisReader.ReadSamples over the same damaged containers, in packet batches and buffers of odd sizes: a Read whose
    packet fails the window check throws (StreamDecoder.cs:777-778) and the stream goes on -- the samples delivered are those
    of the one-call decode, whatever the batching"""
    from test_multi_gpu import single_stream_pcm
    from vorbispizza_amd import capi
    from vorbispizza_amd.front import VorbisReader
    raws = [r for i, r in enumerate(hostile_library()) if i % 3 != 0][:14 * int(os.environ.get("VPZ_HOSTILE_ROUNDS", "1"))]
    for k, raw in enumerate(raws):
        ref = single_stream_pcm(ctx, raw)
        outs = [_read_all(VorbisReader, ctx, raw, batch, buf, capi.SynthError) for batch, buf in ((7, 1000), (64, 4096), (1, 333))]
        for got, thrown in outs:
            assert got.shape == ref.shape, (k, got.shape, ref.shape)
            assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), k
        assert len({t for _, t in outs}) == 1, (k, [t for _, t in outs])


def test_a_stream_that_outgrows_its_announced_length_costs_only_itself(ctx):
    """damage that takes the end-of-stream trim away (StreamDecoder.cs:658-666 trims when the marked packet is read): the
    stream produces more samples than PacketProvider.GetGranuleCount announced (:35-49).  With areas of exactly the announced
    sizes the synth call that holds it fails -- and every other member gets a call of its own: four neighbours of the same
    setup decode to their usual bits, the one that does not fit is VPZM_E_CAPACITY, nothing of it is written"""
    from test_multi_gpu import run_dispatcher, single_stream_pcm
    from vorbispizza_amd import multi
    clean = open(os.path.join(GOLDEN, "1test.ogg"), "rb").read()
    grown = damage_audio(clean, 501, 2)
    raws = [clean, clean, grown, clean, clean]
    pcm, offs, res, _, infos = run_dispatcher([0], raws, capacity_slack=0, host_threads=2, streams_per_call=8)
    produced = single_stream_pcm(ctx, grown).shape[0]
    assert produced > infos[grown][1] == infos[clean][1]
    assert list(res["status"]) == [0, 0, multi.E_CAPACITY, 0, 0]
    ref = single_stream_pcm(ctx, clean)
    for k in (0, 1, 3, 4):
        assert res["samples"][k] == ref.shape[0]
        assert np.array_equal(pcm[offs[k]: offs[k] + ref.size].view(np.uint32), ref.reshape(-1).view(np.uint32)), k
    assert (pcm[offs[2]: offs[3]] == np.float32(7.0)).all()
    # with room for what it produces it is a stream like any other
    pcm, offs, res, _, _ = run_dispatcher([0], raws, capacity_slack=produced - infos[grown][1], host_threads=2, streams_per_call=8)
    assert (res["status"] == 0).all() and res["samples"][2] == produced
