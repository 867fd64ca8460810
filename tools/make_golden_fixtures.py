#!/usr/bin/env python3
"""Writes the small golden vectors of SURVEY.md section 8c into tests/golden/ (committed; data only).

Source of the numbers: the CPU oracle (oracle/vorbis_synth_oracle.c -- the restatement of the reference's
arithmetic) and the C++ front end (bit-exact entropy stage) run HERE on seeded inputs and on the reference's own
.ogg fixtures.  The reference itself cannot run in this pipeline (C#, no .NET) and holds no vectors of its own, so
these files freeze the oracle as it was when tests/test_spec_crosscheck_cpu.py tied it to the specification-derived
float64 synthesis; tests/test_golden_vectors_cpu.py fails if a later edit of the oracle moves a single bit, and
tests/test_golden_vectors_gpu.py holds the HIP path to them without needing any oracle code on the GPU box.

    python tools/make_golden_fixtures.py        # rewrites tests/golden/*.npz
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
OUT = os.path.join(ROOT, "tests", "golden")


def main():
    import helpers
    import oracle
    from vorbispizza_amd.front import OggVorbisFile

    # ---- IMDCT, N = 256 and 2048: seeded N(0, 1) spectra -> Mdct.Reverse outputs (Mdct.cs:15-19)
    vec = {}
    for n, rows in ((256, 8), (2048, 4)):
        x = np.random.default_rng(n).standard_normal((rows, n // 2)).astype(np.float32)
        vec["spectra_%d" % n] = x
        vec["pcm_%d" % n] = oracle.mdct_reverse(x, n)
    np.savez_compressed(os.path.join(OUT, "imdct_vectors.npz"), **vec)

    # ---- window switching + overlap-add: every PacketInfo geometry of Mode.cs:30-66 at least once
    L, P, N = helpers.PKT_BLOCK_FLAG, helpers.PKT_PREV_FLAG, helpers.PKT_NEXT_FLAG
    #                 long pl/nl  long pl/ns  short short long ps/ns short long ps/nl  long pl/nl
    flags = np.array([L | P | N, L | P, 0, 0, L, 0, L | N, L | P | N], dtype=np.uint8)
    spectra = helpers.gaussian_spectra((len(flags), 2, 1024), seed=11)
    pcm = oracle.synth_stream_planar(2, 256, 2048, flags, spectra)
    infos = np.array([[getattr(oracle.packet_info(256, 2048, f & 1, bool(f & 2), bool(f & 4)), k)
                       for k in ("Length", "LeftUseSize1", "LeftStart", "LeftEnd", "RightStart", "RightEnd")]
                      for f in flags], dtype=np.int32)
    np.savez_compressed(os.path.join(OUT, "window_ola_sequence.npz"), flags=flags, spectra=spectra, pcm=pcm,
                        packet_info=infos)

    # ---- inverse coupling: the four sign quadrants, zeros of either sign, equal magnitudes (Mapping.cs:198-269)
    m = np.array([3.0, 3.0, -3.0, -3.0, 0.0, -0.0, 0.0, 2.0, -2.0, 5.0, -5.0, 1.5, -1.5, 0.0, -0.0, 7.0], dtype=np.float32)
    a = np.array([1.0, -1.0, 1.0, -1.0, 2.0, 2.0, -2.0, 0.0, -0.0, 5.0, -5.0, -1.5, 1.5, 0.0, -0.0, -0.0], dtype=np.float32)
    mv, av = oracle.apply_coupling(m, a, vector_form=True)
    ms, as_ = oracle.apply_coupling(m, a, vector_form=False)
    np.savez_compressed(os.path.join(OUT, "coupling_quadrants.npz"), magnitude=m, angle=a, out_magnitude_vector=mv,
                        out_angle_vector=av, out_magnitude_scalar=ms, out_angle_scalar=as_)

    # ---- Floor1: the 29-post long floor of 3test.ogg, posts of its first long packets -> finalY, flags, curve
    f = OggVorbisFile(os.path.join(OUT, "3test.ogg"))
    pk, res, posts, counts = f.decode_packets()
    long_floor = next(i for i, fl in enumerate(f.floors) if len(fl[0]) == 29)
    xlist, mult = f.floors[long_floor]
    of = oracle.floor1_init(xlist, mult)
    rows = [i * 2 + c for i in range(len(pk)) if pk["flags"][i] & 1 for c in range(2) if counts[i * 2 + c]][:16]
    raw = posts[rows].astype(np.int16)
    fy, fl, cur = zip(*[oracle.floor1_indices(of, raw[r].astype(np.int32), 29, 1024) for r in range(len(rows))])
    np.savez_compressed(os.path.join(OUT, "floor1_3test_long.npz"), x_list=np.array(xlist, np.int32),
                        multiplier=np.int32(mult), raw_posts=raw, final_y=np.array(fy, np.int32)[:, :29],
                        step_flags=np.array(fl, np.uint8)[:, :29], table_index=np.array(cur, np.int32),
                        inverse_db_table=oracle.inverse_db_table())

    # ---- the first 4096 PCM samples per channel of every reference fixture, and the decoded lengths
    pcm_head = {}
    for name in ("1test.ogg", "2test.ogg", "3test.ogg", "issue6test.ogg"):
        f = OggVorbisFile(os.path.join(OUT, name))
        pk, res, posts, counts = f.decode_packets()
        ref, pos, clipped = helpers.oracle_decode(oracle, f.channels, f.block_size0, f.block_size1,
                                                  helpers.packets_for_oracle(f, pk, res, posts, counts),
                                                  floors=f.floors, mappings=f.mappings, clip=True)
        key = name.replace(".ogg", "")
        pcm_head[key + "_pcm"] = ref[:, :4096]
        # a window around the loudest sample too (these files start, and some of them continue, with silence)
        mid = int(np.clip(int(np.abs(ref).max(axis=0).argmax()) - 1024, 0, ref.shape[1] - 2048))
        pcm_head[key + "_meta"] = np.array([f.channels, f.sample_rate, f.audio_packets, ref.shape[1], pos, int(clipped), mid],
                                           dtype=np.int64)
        pcm_head[key + "_mid"] = ref[:, mid: mid + 2048]
    np.savez_compressed(os.path.join(OUT, "fixture_pcm_heads.npz"), **pcm_head)
    for n in sorted(os.listdir(OUT)):
        if n.endswith(".npz"):
            print("%-28s %7d bytes" % (n, os.path.getsize(os.path.join(OUT, n))))


if __name__ == "__main__":
    main()
