#!/usr/bin/env python3
"""Finds the seeds of tests/hostile_setups.mutate_setup whose damaged setup header the C++ front end still OPENS and writes them
to tests/golden/hostile_setup_seeds.json (CPU only).  The campaign of tests/test_hostile_setup_gpu.py replays exactly these."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import hostile_setups as hs
    from vorbispizza_amd.front import FrontError, OggVorbisFile
    per_source = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    tried = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    out = {}
    for name, raw in hs.sources().items():
        kept = []
        t0 = time.time()
        for seed in range(tried):
            data = hs.mutate_setup(raw, seed)
            if data == raw:
                continue
            t1 = time.time()
            try:
                f = OggVorbisFile(data)
                f.close()
            except FrontError:
                continue
            if time.time() - t1 > 0.5:  # (a setup that takes the front end seconds to unpack: not for a test suite's time budget)
                continue
            kept.append(seed)
            if len(kept) >= per_source:
                break
        out[name] = kept
        print("%-28s %3d kept of %3d tried (%.1f s)" % (name, len(kept), seed + 1, time.time() - t0))
    json.dump({"what": "seeds of tests/hostile_setups.mutate_setup whose containers the front end opens", "seeds": out},
              open(hs.SEEDS_FILE, "w"), indent=0)
    print("total", sum(len(v) for v in out.values()))


if __name__ == "__main__":
    main()
