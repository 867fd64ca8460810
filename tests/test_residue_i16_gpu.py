"""ABI v5: vpz_decoder_set_residue_format(VPZ_RESIDUE_I16).  The same values as 16-bit integers must give the bits of the float32
call on every route -- the floored stereo fast path reads them in place and widens them in registers (round 5), every other kernel
gets them widened into the decoder's staging buffer first -- from host memory and from device memory (aligned for the in-place read
or not), and through the dispatcher (which ships integral residues as int16 by default)."""
import os

import numpy as np
import pytest

import helpers
from test_dual_gpu import ROUTES, same_bits
from test_host_paths_gpu import env, run, stream_major_batch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def ctx():
    import __graft_entry__ as ge
    ge.build()
    from vorbispizza_amd import Context
    c = Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("name", ["2test.ogg", "3test.ogg", "issue6test.ogg"])
@pytest.mark.parametrize("s16_out", [False, True])
def test_real_files_decode_the_same_from_int16_residue(ctx, name, s16_out):
    from vorbispizza_amd import Decoder, capi
    from vorbispizza_amd.front import OggVorbisFile
    f = OggVorbisFile(os.path.join(GOLDEN, name))
    assert f.residue_is_integral
    layout = capi.OUT_INTERLEAVED_S16 if s16_out else capi.OUT_INTERLEAVED
    outs = []
    for int16 in (False, True, False):  # (and back to float32 on the same decoder)
        pk, res, posts, counts = f.decode_packets(int16=int16)
        if not outs:
            dec = Decoder(ctx, f.channels, f.block_size0, f.block_size1, floors=f.floors, mappings=f.mappings)
        dec.reset(-1)
        outs.append(dec.synth(pk, res, posts, counts, out_layout=layout, on_mismatch="ignore")[0])
    dec.close()
    assert outs[0].shape[0] > 0
    assert np.array_equal(outs[0].view(np.uint8), outs[1].view(np.uint8)) and np.array_equal(outs[0].view(np.uint8), outs[2].view(np.uint8))


@pytest.mark.parametrize("channels,interleaved", [(2, True), (2, False), (6, True), (3, True), (10, True)])
def test_every_route_takes_int16_residue(ctx, channels, interleaved):
    """synthetic batches with integer residues, float32 against int16, on the stereo fast path / group mode / the separate coupling
    pass (route switches of tests/test_dual_gpu.py)"""
    from vorbispizza_amd import capi
    n_streams, frames = 5, 40
    pk, res, posts, counts = stream_major_batch(n_streams, frames, channels, seed=7100 + channels, floor=True, interleaved=interleaved,
                                                p_ls=0.15, p_sl=0.3, silent_prob=0.1)
    res = np.round(res * 8.0).astype(np.float32)
    assert np.abs(res).max() < 32768
    res16 = res.astype(np.int16)
    pk["mapping"] = pk["flags"] & 1
    floors = [(helpers.SHORT_XLIST, 2), (helpers.LONG_XLIST, 2)]
    steps = [(0, 1)] if channels < 4 else [(0, 1), (2, 3)]
    maps = [{"coupling": steps, "channel_floor": [0] * channels}, {"coupling": steps, "channel_floor": [1] * channels}]
    routes = ROUTES if channels == 2 else ROUTES[1:]
    for name, kv in routes:
        with env(**dict(kv, VPZ_PAR_MIN_PACKETS=1, VPZ_HOST_THREADS=3)):
            ref = run(ctx, pk, res, posts, counts, n_streams, channels, floors, maps, layout=capi.OUT_PLANAR, splits=2)
            got = run(ctx, pk, res16, posts, counts, n_streams, channels, floors, maps, layout=capi.OUT_PLANAR, splits=2)
        same_bits(got, ref, "%s, int16 residue" % name)
        assert np.abs(ref[0].astype(np.float64)).max() > 0


def test_int16_residue_in_device_memory_and_bad_formats(ctx):
    import torch
    from vorbispizza_amd import Decoder, capi
    from vorbispizza_amd.front import OggVorbisFile
    f = OggVorbisFile(os.path.join(GOLDEN, "3test.ogg"))
    pk, res, posts, counts = f.decode_packets()
    dev = torch.device("cuda", 0)
    dec = Decoder(ctx, f.channels, f.block_size0, f.block_size1, floors=f.floors, mappings=f.mappings)
    cap = int(f.total_samples) + 2048
    outs = []
    for int16 in (False, True, "misaligned"):
        r = torch.from_numpy(res.astype(np.int16) if int16 else res).to(dev)
        if int16 == "misaligned":  # (a pointer that is only 2-byte aligned: no 8-byte loads, the values are widened first)
            r = torch.cat([torch.zeros(1, dtype=torch.int16, device=dev), r])[1:]
        out = torch.zeros(cap * f.channels, device=dev)
        dec.reset(-1)
        w = dec.synth_raw(pk, r, torch.from_numpy(posts).to(dev), torch.from_numpy(counts).to(dev), out, None, cap, capi.OUT_INTERLEAVED, 0,
                          capi.MEM_DEVICE, residue_floats=res.size)
        ctx.synchronize()
        outs.append(out.cpu().numpy()[: int(w[0]) * f.channels])
    assert outs[0].size > 0 and np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))
    assert np.array_equal(outs[0].view(np.uint32), outs[2].view(np.uint32))
    with pytest.raises(capi.SynthError) as e:
        dec.set_residue_format(7)
    assert e.value.status == capi.E_INVALID_ARG
    dec.close()


def test_the_dispatcher_ships_integral_residues_as_int16_and_floats_on_request(ctx):
    from test_multi_gpu import library, run_dispatcher
    raws = library(("3test.ogg", "issue6test.ogg", "2test.ogg"), 9)
    a = run_dispatcher([0, 0], raws, host_threads=4, streams_per_call=2)
    b = run_dispatcher([0, 0], raws, host_threads=4, streams_per_call=2, float_residue=True)
    assert (a[2]["status"] == 0).all() and (b[2]["status"] == 0).all()
    assert np.array_equal(a[2]["samples"], b[2]["samples"]) and a[2]["samples"].min() > 0
    assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))
