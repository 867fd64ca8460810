"""Randomised differential test of the per-stream state machine (`ReadNextPacket` / `Read`, StreamDecoder.cs:418-498,
640-694) between the library and the oracle: arbitrary block / window flag sequences (consistent or not: mismatches
are part of the game), undecodable packets, resync packets (the position becomes unknown until the next granule), EOS anywhere with or without a granule, granule position pick-up after a
reset, arbitrary batch splits, two block-size pairs.  PCM within 1e-5, everything integer exact."""
import numpy as np
import pytest

import helpers
from helpers import (PKT_BLOCK_FLAG, PKT_EOS, PKT_NEXT_FLAG, PKT_NO_FLOOR, PKT_NOT_DECODED, PKT_PREV_FLAG, PKT_RESYNC)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import __graft_entry__ as ge
    ge.build()
    from vorbispizza_amd import Context
    c = Context(0)
    yield c
    c.close()


def random_stream(rng, frames, size0, size1):
    flags = np.zeros(frames, dtype=np.uint8)
    gran = np.full(frames, -1, dtype=np.int64)
    consistent = rng.random() < 0.7
    bf = (rng.random(frames) < 0.6).astype(np.uint8)
    for f in range(frames):
        if consistent:
            prev = bf[f - 1] if f else 1
            nxt = bf[f + 1] if f + 1 < frames else 1
        else:
            prev, nxt = rng.integers(0, 2, size=2)
        flags[f] = bf[f] * (PKT_BLOCK_FLAG | prev * PKT_PREV_FLAG | nxt * PKT_NEXT_FLAG)
        if rng.random() < 0.06:
            flags[f] |= PKT_NOT_DECODED
        if rng.random() < 0.05:
            flags[f] |= PKT_RESYNC      # lost sync in front of this packet: the position is picked up again (:718-722)
        if rng.random() < 0.15:
            gran[f] = int(rng.integers(0, frames * size1 // 2))
    if rng.random() < 0.7:  # an EOS somewhere in the second half, sometimes on an undecodable packet
        e = int(rng.integers(frames // 2, frames))
        flags[e] |= PKT_EOS
        if rng.random() < 0.6:
            gran[e] = int(rng.integers(0, frames * size1 // 2))
    return flags, gran


@pytest.mark.parametrize("size0,size1", [(256, 2048), (512, 1024)])
@pytest.mark.parametrize("seed", range(12))
def test_random_scenarios_match_the_oracle(ctx, oracle, size0, size1, seed):
    from vorbispizza_amd import Decoder, SynthError, capi, make_packets
    rng = np.random.default_rng(1000 * size0 + seed)
    channels, frames = int(rng.integers(1, 4)), int(rng.integers(20, 60))
    flags, gran = random_stream(rng, frames, size0, size1)
    pk = make_packets(frames)
    opk, chunks, off = [], [], 0
    for f in range(frames):
        half = (size1 if flags[f] & 1 else size0) // 2
        x = helpers.gaussian_spectra((channels, half), seed=seed * 100 + f)
        pk[f]["flags"], pk[f]["granule"], pk[f]["residue_offset"] = flags[f] | PKT_NO_FLOOR, gran[f], off
        off += x.size
        chunks.append(x.reshape(-1))
        opk.append({"flags": int(flags[f]) | PKT_NO_FLOOR, "granule": int(gran[f]), "residue": x.reshape(-1)})
    res = np.concatenate(chunks)
    # the oracle's view; a reset in the middle of some scenarios exercises the granule pick-up (:459-463)
    reset_at = int(rng.integers(5, frames - 5)) if rng.random() < 0.4 else None
    if reset_at is None:
        ref, pos, _ = helpers.oracle_decode(oracle, channels, size0, size1, opk)
        ref_parts, positions = [ref], [pos]
    else:
        a, pa, _, st = helpers.oracle_decode(oracle, channels, size0, size1, opk[:reset_at], keep_state=True)
        oracle.lib().orc_stream_reset(st)  # ResetDecoder keeps the position value and clears _hasPosition (:357-369)
        b, pb, _ = helpers.oracle_decode(oracle, channels, size0, size1, opk[reset_at:], state=st)
        ref_parts, positions = [a, b], [pa, pb]
    # the library: random batch splits, a reset between the two parts
    dec = Decoder(ctx, channels, size0, size1)
    got_parts = []
    bounds = [0, frames] if reset_at is None else [0, reset_at, frames]
    for part in range(len(bounds) - 1):
        lo, hi = bounds[part], bounds[part + 1]
        if part:
            dec.reset(0)
        cuts = sorted(set([lo, hi] + [int(v) for v in rng.integers(lo, hi + 1, size=3)]))
        outs = []
        for a, b in zip(cuts[:-1], cuts[1:]):
            sub = pk[a:b].copy()
            cap = (b - a) * size1 + 8
            out = np.zeros(channels * cap, dtype=np.float32)
            try:
                w = dec.synth_raw(sub, res, None, None, out, None, cap, capi.OUT_PLANAR, cap, capi.MEM_HOST)
            except SynthError as e:
                assert e.status == capi.E_WINDOW_MISMATCH
                w = [int(dec.last_packet_samples(b - a).sum())]
            outs.append(out.reshape(channels, cap)[:, : int(w[0])])
        got_parts.append(np.concatenate(outs, axis=1) if outs else np.zeros((channels, 0), np.float32))
        assert dec.position(0) == positions[part], (part, dec.position(0), positions[part])
    for got, ref in zip(got_parts, ref_parts):
        assert got.shape == ref.shape
        if ref.size:
            assert np.abs(got - ref).max() <= 1e-5
    dec.close()


def test_a_call_that_only_drains(ctx, oracle):
    """A batch whose only frame is the un-windowed drain of the previous block (an undecodable EOS packet,
    StreamDecoder.cs:451-455): the kernel runs with nothing to load -- its unconditional prefetches must not touch the
    (empty) residue -- and emits the rest of the previous block."""
    from vorbispizza_amd import Decoder, capi, make_packets
    channels, size0, size1 = 2, 256, 2048
    spectra = helpers.gaussian_spectra((3, channels, 1024), seed=5)
    pk = make_packets(4)
    opk = []
    for f in range(3):
        pk[f]["flags"] = PKT_BLOCK_FLAG | PKT_PREV_FLAG | PKT_NEXT_FLAG | PKT_NO_FLOOR
        pk[f]["granule"] = -1
        pk[f]["residue_offset"] = f * channels * 1024
        opk.append({"flags": int(pk[f]["flags"]), "granule": -1, "residue": spectra[f].reshape(-1)})
    pk[3]["flags"] = PKT_NOT_DECODED | PKT_EOS
    pk[3]["granule"] = -1
    opk.append({"flags": int(pk[3]["flags"]), "granule": -1, "residue": np.zeros(0, np.float32)})
    ref, pos, _ = helpers.oracle_decode(oracle, channels, size0, size1, opk)
    dec = Decoder(ctx, channels, size0, size1)
    cap = 4 * 1024
    out_a = np.zeros(channels * cap, dtype=np.float32)
    w_a = dec.synth_raw(pk[:3].copy(), spectra.reshape(-1), None, None, out_a, None, cap, capi.OUT_PLANAR, cap, capi.MEM_HOST)
    out_b = np.zeros(channels * cap, dtype=np.float32)
    w_b = dec.synth_raw(pk[3:].copy(), np.zeros(4, np.float32), None, None, out_b, None, cap, capi.OUT_PLANAR, cap, capi.MEM_HOST)
    got = np.concatenate([out_a.reshape(channels, cap)[:, : int(w_a[0])], out_b.reshape(channels, cap)[:, : int(w_b[0])]], axis=1)
    assert int(w_b[0]) > 0 and got.shape == ref.shape
    assert np.abs(got - ref).max() <= 1e-5
    assert dec.position(0) == pos
    dec.close()
