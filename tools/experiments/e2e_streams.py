#!/usr/bin/env python3
"""Tuning helper: configs[4] end to end (host entropy decode on T threads + host-memory synth calls): thread counts, and
the pipeline's granularity (streams per synth call, contexts taking the calls in turn).
usage: e2e_streams.py [copies] [threads | sweep]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import bench
    from vorbispizza_amd import Context
    ctx = Context(0)
    copies = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    mode = sys.argv[2] if len(sys.argv) > 2 else "threads"
    if mode == "threads":
        for thr in (1, 2, 4, 8, 16):
            tot, (t_all, t_dec, t_syn) = bench.end_to_end_real_streams(ctx, torch, copies, thr)
            print("%2d threads: decode %.1f ms, synth(host mem) %.1f ms, end to end %.0f Msamples/s"
                  % (thr, t_dec * 1e3, t_syn * 1e3, tot / t_all / 1e6), flush=True)
    else:
        thr = bench.host_threads()
        for rep in range(2):
            for sub, lanes in [tuple(int(v) for v in c.split("x")) for c in os.environ.get("E2E_CASES", "16x2,8x2,8x3,4x3,16x3").split(",")]:
                for s16 in (False, True):
                    for streamed in (False, True):
                        tot, (t_all, t_dec, t_syn) = bench.end_to_end_real_streams(ctx, torch, copies, thr, sub=sub, synth_lanes=lanes,
                                                                                   s16=s16, streamed=streamed)
                        print("%s, %2d threads, %2d streams per call, %d contexts, %s: wall %.1f ms (decode %.1f, synth calls summed %.1f), %.0f Msamples/s"
                              % ("streamed " if streamed else "fork-join", thr, sub, lanes, "s16" if s16 else "f32", t_all * 1e3, t_dec * 1e3,
                                 t_syn * 1e3, tot / t_all / 1e6), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
