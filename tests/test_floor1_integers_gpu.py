"""Bit-exactness of the Floor1 INTEGER work that runs on the GPU (north_star: "the Huffman/indexing stage bit-exact").

`Floor1.UnwrapPosts` (Floor1.cs:270-353), the choice of the posts a line is drawn to (:236-252) and the DDA of
`RenderLineMulti` (:372-397) run in floor1_unwrap_kernel and in render_floor_indices -- the device function the
fused synthesis kernel calls.  The test-only entry vpz_debug_floor1_indices (include/vorbispizza_synth_debug.h)
reads their integers back: finalY * multiplier, the step flags and the inverse-dB table index of EVERY bin, and
this file compares them index for index with the oracle's restatement over > 10^5 random records:
x_count 2..64, multipliers 1..4, half block sizes 32..4096, X lists with posts BEYOND n (quirk q2: the reference
passes min(hx, n) into the slope, Floor1.cs:248), silent channels, room-clamp cases."""
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


def random_xlist(rng, n, count, beyond):
    """X list as Floor1's constructor builds it: 0, 1 << rangeBits, then distinct values in between -- with
    `beyond` the range is twice the half block size, so posts at and beyond n exist (a stream that pairs a floor
    with a shorter block than its range was made for; the reference accepts it)."""
    top = 2 * n if beyond else n
    inner = rng.choice(np.arange(1, top), size=count - 2, replace=False)
    return [0, top] + [int(v) for v in inner]


def random_records(rng, floor, n_rec, multiplier, wild=0.02):
    """Raw posts as Unpack leaves them: two absolute values, then residuals of every size class -- zero, small,
    and beyond the room so that both clamp branches of UnwrapPosts run; a few out-of-range absolutes."""
    count = len(floor)
    rng_range = {1: 256, 2: 128, 3: 86, 4: 64}[multiplier]
    posts = np.zeros((n_rec, 64), dtype=np.int16)
    posts[:, 0] = rng.integers(0, rng_range, size=n_rec)
    posts[:, 1] = rng.integers(0, rng_range, size=n_rec)
    vals = rng.integers(0, 24, size=(n_rec, count - 2))
    vals[rng.random(vals.shape) < 0.4] = 0
    big = rng.random(vals.shape) < 0.1
    vals[big] = rng.integers(0, 2 * rng_range, size=int(big.sum()))
    posts[:, 2:count] = vals
    w = rng.random(n_rec) < wild
    posts[w, 0] = rng.integers(-300, 600, size=int(w.sum()))
    counts = np.full(n_rec, count, dtype=np.uint8)
    counts[rng.random(n_rec) < 0.05] = 0      # ExecuteChannel false
    return posts, counts


def oracle_rows(oracle, f, posts, counts, n):
    ys, fl, cur = [], [], []
    for r in range(len(posts)):
        if counts[r] == 0:
            ys.append(None)
            fl.append(None)
            cur.append(None)
            continue
        fy, flags, idx = oracle.floor1_indices(f, posts[r].astype(np.int32), int(counts[r]), n)
        ys.append(fy)
        fl.append(flags)
        cur.append(idx)
    return ys, fl, cur


CASES = [
    # (size0, size1, which block, x_count, multiplier, posts beyond n, records)
    (256, 2048, 1, 29, 2, False, 24000),      # the shape of 3test.ogg's long floor
    (256, 2048, 0, 19, 2, False, 24000),      # ... and its short floor
    (256, 2048, 1, 64, 1, False, 6000),       # the most posts the reference can hold
    (256, 2048, 1, 2, 4, False, 2000),        # two posts: one line
    (256, 2048, 1, 40, 3, True, 12000),       # q2: posts at and beyond n
    (256, 2048, 0, 33, 4, True, 12000),
    (64, 512, 0, 9, 2, False, 4000),          # n = 32: fewer bins than lanes
    (64, 512, 0, 12, 1, True, 4000),
    (512, 1024, 1, 48, 2, True, 8000),
    (128, 4096, 1, 64, 3, False, 3000),       # n = 2048 (the wide bitmap)
    (128, 8192, 1, 50, 2, True, 3000),        # n = 4096
    (128, 8192, 0, 7, 4, False, 3000),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "n%d_x%d_m%d%s" % ((c[1] if c[2] else c[0]) // 2, c[3], c[4],
                                                                         "_beyond" if c[5] else ""))
def test_unwrapped_posts_flags_and_every_table_index_match_the_reference_arithmetic(ctx, oracle, case):
    from vorbispizza_amd import Decoder
    size0, size1, long_blk, x_count, mult, beyond, n_rec = case
    n = (size1 if long_blk else size0) // 2
    rng = np.random.default_rng(hash(case) & 0xFFFFFFFF)
    xlist = random_xlist(rng, n, x_count, beyond)
    other = [0, max(2, (size0 if long_blk else size1) // 2), 1]
    floors = [(xlist, mult), (other, 1)]
    dec = Decoder(ctx, 1, size0, size1, floors=floors, mappings=[{"coupling": [], "channel_floor": [0]}])
    posts, counts = random_records(rng, xlist, n_rec, mult)
    curve, final_y, flags, active = dec.debug_floor1_indices(posts, counts, np.zeros(n_rec, np.uint8),
                                                             np.full(n_rec, long_blk, np.uint8))
    f = oracle.floor1_init(xlist, mult)
    ys, fls, curs = oracle_rows(oracle, f, posts, counts, n)
    if beyond:
        assert max(xlist) > n and any(n < x for x in xlist[2:]), "the case must hold posts beyond n"
    checked = clamped = 0
    for r in range(n_rec):
        if counts[r] == 0:
            assert active[r] == 0 and (curve[r] == 0xEE).all()   # nothing is written for a silent channel
            continue
        want_y = ys[r][:x_count] * mult
        in_i16 = (np.abs(want_y) < 32768).all()
        if in_i16:
            np.testing.assert_array_equal(final_y[r, :x_count], want_y, err_msg="finalY, record %d" % r)
        np.testing.assert_array_equal(flags[r, :x_count], fls[r][:x_count], err_msg="step flags, record %d" % r)
        assert active[r] == int(fls[r][:x_count].sum())   # post 0 and every flagged post
        want = curs[r]
        if not in_i16:
            continue
        inside = (want >= 0) & (want <= 255)
        clamped += int((~inside).sum())
        np.testing.assert_array_equal(curve[r, :n], np.clip(want, 0, 255).astype(np.uint8),
                                      err_msg="table indices, record %d" % r)
        assert (curve[r, n:] == 0xEE).all()                      # nothing beyond the block's bins
        checked += 1
    assert checked > 0.9 * n_rec


def test_record_total_and_q2_is_really_exercised(ctx, oracle):
    """The cases above cover > 10^5 records, and the posts-beyond-n cases differ from a render that clips only the
    loop (libvorbis / stb behaviour): without quirk q2 in the device code they would fail."""
    assert sum(c[6] for c in CASES) >= 100000
    from vorbispizza_amd import Decoder
    n = 1024
    xlist = [0, 2048, 700, 1500]        # one post beyond n: the last drawn segment runs 700 -> min(1500, n)
    dec = Decoder(ctx, 1, 256, 2048, floors=[(xlist, 1)], mappings=[{"coupling": [], "channel_floor": [0]}])
    posts = np.zeros((1, 64), dtype=np.int16)
    posts[0, :4] = [40, 10, 100, 90]   # absolute 40 at x=0, 10 at x=2048; residuals move 700 and 1500
    counts = np.array([4], dtype=np.uint8)
    curve, final_y, flags, active = dec.debug_floor1_indices(posts, counts, [0], [1])
    f = oracle.floor1_init(xlist, 1)
    fy, fl, want = oracle.floor1_indices(f, posts[0].astype(np.int32), 4, n)
    np.testing.assert_array_equal(curve[0, :n], np.clip(want, 0, 255))
    # the libvorbis-style render (slope towards the true x = 1500, loop clipped at n) gives another curve
    y700, y1500 = int(fy[2]), int(fy[3])
    k = np.arange(n - 700)
    other = y700 + np.sign(y1500 - y700) * ((abs(y1500 - y700) * k) // (1500 - 700))
    assert not np.array_equal(other, want[700:]), "the case does not distinguish q2"


def test_steep_lines_take_the_integer_walk_and_still_match(ctx, oracle):
    """The fused kernel draws the curve in float32 closed form while |dy| * adx <= 2^21 per segment and falls back to the
    integer walk of the reference (Floor1.cs:386-396) otherwise.  No valid stream breaks the bound; these records do --
    absolute posts of several thousand at both ends of the block -- so the fallback stays under test, next to records
    in the same batch that take the fast path."""
    from vorbispizza_amd import Decoder
    n, mult = 1024, 2
    rng = np.random.default_rng(77)
    xlist = random_xlist(rng, n, 24, False)
    dec = Decoder(ctx, 1, 256, 2048, floors=[(xlist, mult), ([0, 128, 1], 1)],
                  mappings=[{"coupling": [], "channel_floor": [0]}])
    n_rec = 3000
    posts, counts = random_records(rng, xlist, n_rec, mult, wild=0.0)
    steep = rng.random(n_rec) < 0.5
    posts[steep, 0] = rng.integers(-6000, 6000, size=int(steep.sum()))
    posts[steep, 1] = rng.integers(-6000, 6000, size=int(steep.sum()))
    posts[steep, 2:24] = 0    # (one line from end to end: |dy| * adx up to 24 000 * 1024)
    curve, final_y, flags, active = dec.debug_floor1_indices(posts, counts, np.zeros(n_rec, np.uint8), np.ones(n_rec, np.uint8))
    f = oracle.floor1_init(xlist, mult)
    ys, fls, curs = oracle_rows(oracle, f, posts, counts, n)
    beyond_bound = 0
    for r in range(n_rec):
        if counts[r] == 0:
            continue
        want_y = ys[r][:24] * mult
        assert (np.abs(want_y) < 32768).all()
        np.testing.assert_array_equal(final_y[r, :24], want_y)
        np.testing.assert_array_equal(curve[r, :n], np.clip(curs[r], 0, 255).astype(np.uint8), err_msg="record %d" % r)
        if steep[r] and abs(int(want_y[0]) - int(want_y[1])) * n > (1 << 21):
            beyond_bound += 1
    assert beyond_bound > 500


@pytest.mark.parametrize("sizes", [(8, 17, 29, 40, 12, 64), tuple(2 + (i * 7) % 63 for i in range(64))], ids=["six", "sixty-four"])
def test_more_floors_than_the_lds_copy_holds(ctx, oracle, sizes):
    """floor1_unwrap_kernel keeps up to four floors' tables in LDS; a decoder with more reads them from memory (the
    other instantiation).  Six floors -- and sixty-four, what a setup header can hold (6 bits + 1) --, the records spread
    over all of them."""
    from vorbispizza_amd import Decoder
    n, mult = 1024, 2
    rng = np.random.default_rng(99)
    xlists = [random_xlist(rng, n, c, False) for c in sizes]
    dec = Decoder(ctx, 1, 256, 2048, floors=[(x, mult) for x in xlists],
                  mappings=[{"coupling": [], "channel_floor": [0]}])
    per = 1800 // len(sizes)
    posts_all, counts_all, which = [], [], []
    for fi, xl in enumerate(xlists):
        p, c = random_records(rng, xl, per, mult)
        posts_all.append(p)
        counts_all.append(c)
        which += [fi] * per
    posts, counts = np.concatenate(posts_all), np.concatenate(counts_all)
    order = rng.permutation(len(which))
    posts, counts, which = posts[order], counts[order], np.asarray(which, np.uint8)[order]
    curve, final_y, flags, active = dec.debug_floor1_indices(posts, counts, which, np.ones(len(which), np.uint8))
    handles = [oracle.floor1_init(x, mult) for x in xlists]
    checked = 0
    for r in range(len(which)):
        if counts[r] == 0:
            assert active[r] == 0
            continue
        xc = len(xlists[which[r]])
        fy, fl, idx = oracle.floor1_indices(handles[which[r]], posts[r].astype(np.int32), int(counts[r]), n)
        if not (np.abs(fy[:xc] * mult) < 32768).all():
            continue
        np.testing.assert_array_equal(final_y[r, :xc], fy[:xc] * mult)
        np.testing.assert_array_equal(flags[r, :xc], fl[:xc])
        np.testing.assert_array_equal(curve[r, :n], np.clip(idx, 0, 255).astype(np.uint8))
        checked += 1
    assert checked > 1400
