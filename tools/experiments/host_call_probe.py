#!/usr/bin/env python3
"""One host-memory vpz_decoder_synth call of the end-to-end leg in isolation (16 real stereo streams, pinned buffers, idle host):
what the call costs next to its copies (VPZ_HOST_PROFILE=1 prints the library's own split)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import bench
    from vorbispizza_amd import Context, Decoder, capi, front
    from vorbispizza_amd.front import OggVorbisFile
    ctx = Context(0)
    name, samples = bench.REAL_FIXTURES[0]
    data = np.frombuffer(open(os.path.join(ROOT, "tests", "golden", name), "rb").read(), dtype=np.uint8)
    probe = OggVorbisFile(data.tobytes())
    n, C_, rf = probe.audio_packets, probe.channels, probe.info.residue_floats
    sub = int(sys.argv[1]) if len(sys.argv) > 1 else 16

    def pinned(k, dtype):
        return torch.empty(k, dtype=dtype, pin_memory=True).numpy()
    pk = capi.make_packets(n * sub)
    res, posts, counts = pinned(rf * sub, torch.float32), pinned(n * sub * C_ * 64, torch.int16).reshape(n * sub * C_, 64), pinned(n * sub * C_, torch.uint8)
    front.decode_many([data] * sub, [j * n for j in range(sub)], [j * rf for j in range(sub)], pk, res, posts, counts, threads=8)
    cap = samples + 2048
    for s16 in (False, True):
        out = pinned(sub * cap * C_, torch.int16 if s16 else torch.float32)
        dec = Decoder(ctx, C_, probe.block_size0, probe.block_size1, floors=probe.floors, mappings=probe.mappings, n_streams=sub)
        offs = np.arange(sub, dtype=np.int64) * cap * C_
        ts = []
        for i in range(12):
            dec.reset(-1)
            t = time.perf_counter()
            dec.synth_raw(pk, res, posts, counts, out, offs, cap, capi.OUT_INTERLEAVED_S16 if s16 else capi.OUT_INTERLEAVED, 0, capi.MEM_HOST,
                          on_mismatch="ignore")
            ts.append(time.perf_counter() - t)
        mb_in = (res.nbytes + posts.nbytes + counts.nbytes) / 1e6
        mb_out = sub * samples * C_ * (2 if s16 else 4) / 1e6
        print("%d streams, %s: call %.2f ms (best of 12; median %.2f); H2D %.1f MB, D2H %.1f MB = %.2f ms at 57 GB/s each"
              % (sub, "s16" if s16 else "f32", min(ts) * 1e3, sorted(ts)[6] * 1e3, mb_in, mb_out, (mb_in + mb_out) / 57e3 * 1e3), flush=True)
        dec.close()
    ctx.close()


if __name__ == "__main__":
    main()
