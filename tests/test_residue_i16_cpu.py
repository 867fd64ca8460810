"""ABI v5, host side: the residue as 16-bit integers (vpzh_residue_is_integral / vpzh_decode_range_i16, vorbispizza_front.h).  A residue
value is a sum of codebook values (Residue0.cs:144-205); libvorbis' residue books are integer lattices, so for the reference's
fixtures every value is an integer and the int16 form is the float32 one, value for value."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("name", ["1test.ogg", "2test.ogg", "3test.ogg", "issue6test.ogg"])
def test_the_fixtures_residues_are_integers_and_the_int16_form_is_the_float_one(name):
    import __graft_entry__ as ge
    ge.build()
    from vorbispizza_amd.front import OggVorbisFile
    f = OggVorbisFile(os.path.join(GOLDEN, name))
    assert f.residue_is_integral
    pk, res, posts, counts = f.decode_packets()
    pk16, res16, posts16, counts16 = f.decode_packets(int16=True)
    assert res16.dtype == np.int16 and res.dtype == np.float32
    assert np.array_equal(res16.astype(np.float32), res)
    assert np.array_equal(pk, pk16) and np.array_equal(posts, posts16) and np.array_equal(counts, counts16)


def test_a_stream_with_fractional_codebook_values_is_not_integral_and_is_refused():
    import __graft_entry__ as ge
    ge.build()
    import synthetic_streams as ss
    from vorbispizza_amd.front import FrontError, OggVorbisFile
    seen = {True: 0, False: 0}
    for name in ("stereo_coupled_res2", "mono_floor1_res1", "six_channels_51"):
        stream, rng = ss.ALL[name]()
        ogg, _ = stream.build(rng, 8)
        f = OggVorbisFile(bytes(ogg))
        pk, res, _, _ = f.decode_packets()
        integral_values = bool(np.array_equal(res, np.round(res)) and np.abs(res).max(initial=0) < 32768)
        seen[f.residue_is_integral] += 1
        if f.residue_is_integral:
            assert integral_values  # (the setup header's word holds for what was decoded)
            assert np.array_equal(f.decode_packets(int16=True)[1].astype(np.float32), res)
        else:
            with pytest.raises(FrontError):
                f.decode_packets(int16=True)
    assert seen[False] > 0, "the spec-based writer's books are random floats: at least one stream must come out non-integral"
