"""Damaged SETUP headers through the device path (round-4 review, item 6): 576 containers whose setup packet was changed in one
to three bytes after encoding and which the front end still opens (tests/hostile_setups.py, seeds committed in
tests/golden/hostile_setup_seeds.json) -- floors, residues, mappings and modes no encoder wrote.  Each goes through
vpz_decoder_create + a synth call (the plain ABI) and through the dispatcher, several per call.

What the reference does with such a stream is an InvalidDataException out of StreamDecoder.cs:262-321 / Floor1.cs:39-155 /
Mapping.cs:19-95 where ITS checks catch it, and whatever the numbers give where they do not (NVorbis.Tests/AssetTest.cs:201-213
holds its decoder to "no crash" on such input).  Here every case must end
  * in a status -- VPZ_E_* out of vpz_decoder_create / vpz_decoder_synth, VPZM_E_* from the dispatcher, the same either way --, or
  * in PCM, and then: the oracle driven from the same parsed setup and the same decoded packets gives that PCM (<= 1e-5 x the
    peak; only where every table index the setup produces is one the reference's table has -- beyond that the reference
    throws, the kernels clamp and the C oracle may not be asked), and the dispatcher gives it bit for bit in any partition,
with no HIP fault, no hang (the suite's timeout) and the undamaged neighbours of the same calls untouched."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from vorbispizza_amd import Context
    c = Context(0)
    yield c
    c.close()


def parsed(raw):
    """what the front end makes of a container: the file object and its packets' batch arrays, or the exception"""
    from vorbispizza_amd.front import FrontError, OggVorbisFile
    try:
        f = OggVorbisFile(raw)
        return f, f.decode_packets()
    except FrontError as e:
        return None, e


def indices_in_table(orc, f, pk, posts, counts, limit=400):
    """True if every inverse-dB-table index the type-1 floors of the first `limit` packets render lies in the reference's
    table (Floor1.cs:383, 395 index it unchecked: 0..255) -- the envelope inside which the C oracle may be asked"""
    C_ = f.channels
    inits = {}
    for i in range(min(len(pk), limit)):
        flags = int(pk["flags"][i])
        if flags & 0x10:
            continue
        n = (f.block_size1 if flags & 1 else f.block_size0) // 2
        m = f.mappings[int(pk["mapping"][i])]
        for c in range(C_):
            fl = m["channel_floor"][c]
            cnt = int(counts[i * C_ + c])
            if cnt == 0 or isinstance(f.floors[fl], dict):
                continue
            if fl not in inits:
                inits[fl] = orc.floor1_init(*f.floors[fl])
            _, _, idx = orc.floor1_indices(inits[fl], posts[i * C_ + c].astype(np.int32), cnt, n)
            if idx.min() < 0 or idx.max() > 255:
                return False
    return True


def test_damaged_setup_headers_end_in_a_status_or_in_the_oracles_pcm(ctx):
    import helpers
    import hostile_setups as hs
    import oracle
    from test_multi_gpu import single_stream_pcm
    from vorbispizza_amd import capi
    cases = hs.committed_cases()
    assert len(cases) >= 500
    n_status, n_pcm, n_oracle, n_front = 0, 0, 0, 0
    outcomes = {}
    for name, seed, raw in cases:
        f, got = parsed(raw)
        if f is None:
            n_front += 1  # (opens, but its packets cannot be decoded: a front-end status)
            outcomes[(name, seed)] = ("front", None)
            continue
        pk, res, posts, counts = got
        try:
            pcm = single_stream_pcm(ctx, raw)
        except capi.SynthError as e:
            assert e.status in (capi.E_INVALID_ARG, capi.E_UNSUPPORTED, capi.E_NOMEM, capi.E_CAPACITY), (name, seed, e)
            n_status += 1
            outcomes[(name, seed)] = ("status", e.status)
            f.close()
            continue
        n_pcm += 1
        outcomes[(name, seed)] = ("pcm", pcm)
        # the oracle, from the same parsed setup and the same packets (first 400: enough to meet every floor and mapping)
        lim = min(len(pk), 400)
        if lim and not any(isinstance(fl, dict) for fl in f.floors) and indices_in_table(oracle, f, pk, posts, counts, lim):
            ref, _, _ = helpers.oracle_decode(oracle, f.channels, f.block_size0, f.block_size1,
                                              helpers.packets_for_oracle(f, pk[:lim], res, posts, counts),
                                              floors=f.floors, mappings=f.mappings, interleave=True)
            t = min(ref.shape[0], pcm.shape[0])
            a, b = pcm[:t].astype(np.float64), ref[:t].astype(np.float64)
            fin = np.isfinite(b)
            assert np.array_equal(np.isfinite(a), fin), (name, seed)
            peak = max(1.0, float(np.abs(b[fin]).max()) if fin.any() else 1.0)
            assert float(np.abs(a[fin] - b[fin]).max() if fin.any() else 0.0) <= 1e-5 * peak, (name, seed)
            n_oracle += 1
        f.close()
    print("hostile setups: %d cases -- %d PCM (%d of them held to the oracle), %d statuses from the back end, %d from the front end's "
          "packet decode" % (len(cases), n_pcm, n_oracle, n_status, n_front))
    assert n_pcm >= 300 and n_oracle >= 150
    # the context is as good as before
    clean = open(os.path.join(hs.GOLDEN, "3test.ogg"), "rb").read()
    a = single_stream_pcm(ctx, clean)
    b = single_stream_pcm(ctx, clean)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    test_damaged_setup_headers_end_in_a_status_or_in_the_oracles_pcm.outcomes = outcomes


def test_the_dispatcher_gives_every_damaged_setup_its_own_outcome(ctx):
    """the same containers, many per vpzm_decode_library call, undamaged files between them, two partitions: a stream's
    outcome is its own (a setup the back end refuses is VPZM_E_SETUP for its streams alone), PCM bit-equal to the plain ABI's"""
    import hostile_setups as hs
    from test_multi_gpu import run_dispatcher, single_stream_pcm
    from vorbispizza_amd import capi, multi
    cases = hs.committed_cases()
    src = hs.sources()
    raws, kinds = [], []
    for i, (name, seed, raw) in enumerate(cases):
        if i % 6 == 0:
            raws.append(src[name])  # an undamaged neighbour
            kinds.append(None)
        raws.append(raw)
        kinds.append((name, seed))
    runs = [run_dispatcher([0], raws, host_threads=6, streams_per_call=8, capacity_slack=8192),
            run_dispatcher([0, 0, 0], raws, host_threads=6, streams_per_call=3, capacity_slack=8192)]
    pcm0, offs, res0, _, infos = runs[0]
    pcm1, _, res1, _, _ = runs[1]
    for field in ("status", "samples", "packets", "skipped_packets", "channels"):
        assert np.array_equal(res0[field], res1[field]), field
    ok = res0["status"] == 0
    for k in np.nonzero(ok)[0]:
        n = int(res0["samples"][k]) * int(res0["channels"][k])
        assert np.array_equal(pcm0[offs[k]: offs[k] + n].view(np.uint32), pcm1[offs[k]: offs[k] + n].view(np.uint32)), k
    assert set(int(s) for s in res0["status"]) <= {0, multi.E_OPEN, multi.E_SETUP, multi.E_SYNTH, multi.E_CAPACITY}
    clean_pcm = {}
    n_checked = 0
    for k, kind in enumerate(kinds):
        if kind is None:  # the undamaged neighbours: their usual bits
            assert res0["status"][k] == 0, k
            if raws[k] not in clean_pcm:
                clean_pcm[raws[k]] = single_stream_pcm(ctx, raws[k])
            ref = clean_pcm[raws[k]]
            assert res0["samples"][k] == ref.shape[0]
            assert np.array_equal(pcm0[offs[k]: offs[k] + ref.size].view(np.uint32), ref.reshape(-1).view(np.uint32)), k
        elif k % 5 == 0:  # a fifth of the damaged ones against the plain ABI (the first test holds all of them to the oracle)
            try:
                ref = single_stream_pcm(ctx, raws[k])
            except capi.SynthError:
                assert res0["status"][k] in (multi.E_SETUP, multi.E_SYNTH, multi.E_CAPACITY), (kind, int(res0["status"][k]))
                continue
            except Exception:
                continue  # (a front-end failure of the packet decode: the dispatcher's status for it is its own)
            if res0["status"][k] == 0:
                assert res0["samples"][k] == ref.shape[0], kind
                assert np.array_equal(pcm0[offs[k]: offs[k] + ref.size].view(np.uint32), ref.reshape(-1).view(np.uint32)), kind
                n_checked += 1
    print("dispatcher: %d containers, statuses %s, %d damaged ones bit-equal to the plain ABI" % (
        len(raws), {int(s): int((res0["status"] == s).sum()) for s in set(res0["status"])}, n_checked))
    assert n_checked >= 40


def test_setups_at_the_edges_of_what_the_headers_can_say(ctx):
    """Containers built with the specification-based writer (tests/hostile_setups.crafted): 64 posts with multiplier 4, 65 posts
    (one more than the reference's `Posts = new int[64]`, Floor1.cs:17: refused, VPZM_E_SETUP for that stream alone), X lists that
    crowd both ends of the block, residue ranges beyond the block and begin > end (Residue0.cs:122-125 clamps them), twelve modes
    on twenty-four floors, eight channels coupled in a ring (Mapping.cs:166-172 in reverse order: group mode's levels), forty
    channels in sixteen submaps with a chain of 39 coupling steps -- each against the oracle, alone and all in one library."""
    import helpers
    import hostile_setups as hs
    import oracle
    from test_multi_gpu import run_dispatcher, single_stream_pcm
    from vorbispizza_amd import multi
    from vorbispizza_amd.front import FrontError, OggVorbisFile
    made = hs.crafted()
    pcms = {}
    for name, (raw, expect) in made.items():
        f, got = parsed(raw)
        if expect == "refused":
            assert f is None and isinstance(got, FrontError), name
            continue
        assert f is not None, (name, got)
        pk, res, posts, counts = got
        pcm = single_stream_pcm(ctx, raw)
        assert indices_in_table(oracle, f, pk, posts, counts, len(pk)), name
        ref, _, _ = helpers.oracle_decode(oracle, f.channels, f.block_size0, f.block_size1,
                                          helpers.packets_for_oracle(f, pk, res, posts, counts),
                                          floors=f.floors, mappings=f.mappings, interleave=True)
        assert pcm.shape == ref.shape and pcm.shape[0] > 0, (name, pcm.shape, ref.shape)
        peak = max(1.0, float(np.abs(ref).max()))
        assert float(np.abs(pcm.astype(np.float64) - ref).max()) <= 1e-5 * peak, name
        pcms[name] = pcm
        f.close()
    # ... and side by side in one library, the refused one among them
    names = list(made)
    raws = [made[n][0] for n in names]
    from vorbispizza_amd import multi as _m
    caps, sizes, chans = [], [], []
    for n in names:
        if made[n][1] == "refused":
            caps.append(65536)  # (room for whatever its 20 packets announce: its status must be about the setup, not the area)
            chans.append(2)
        else:
            caps.append(pcms[n].shape[0] + 64)
            chans.append(pcms[n].shape[1])
        sizes.append(caps[-1] * chans[-1])
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
    for groups in (1, 3):
        pcm = np.full(int(sum(sizes)), np.float32(7.0), dtype=np.float32)
        d = _m.Dispatcher([0] * groups, host_threads=4, streams_per_call=4)
        try:
            results, _ = d.decode_library([np.frombuffer(r, dtype=np.uint8) for r in raws], pcm, offs, np.array(caps, dtype=np.int64))
        finally:
            d.close()
        for k, n in enumerate(names):
            if made[n][1] == "refused":
                assert results["status"][k] == multi.E_SETUP, (n, int(results["status"][k]))
                assert (pcm[offs[k]: offs[k] + sizes[k]] == np.float32(7.0)).all()
            else:
                assert results["status"][k] == 0, (n, int(results["status"][k]))
                ref = pcms[n]
                assert results["samples"][k] == ref.shape[0]
                assert np.array_equal(pcm[offs[k]: offs[k] + ref.size].view(np.uint32), ref.reshape(-1).view(np.uint32)), n
