#!/usr/bin/env python3
"""Writes mutated blobs (1–3 corrupted words each: header, tables, template records, payloads)
of several small models: `python tools/fuzz_blob_gen.py SEED N_PER_MODEL OUTDIR`."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)

NAMES = ["quadrotor_5", "pandemic_20x3", "farmer_5", "opf_7", "operator_zoo", "irregular", "quadrotor_oc3_40",
         "foreign:pandemic_20x3", "foreign:quadrotor_oc3_40"]      # flat explicit iterators: exercises the lattice recovery


def blob_of(name):
    import cases
    if name.startswith("foreign:"):
        from test_foreign_producer import foreign_blob
        return foreign_blob(cases.build_core(name.split(":", 1)[1]))[1]
    return cases.build_core(name).to_blob()


def mutations(words, rng, n):
    w = words
    payload = min([int(w[14 + 6 * i + 2]) for i in range(int(w[6])) if int(w[14 + 6 * i + 2]) > 0] or [len(w)])
    for _ in range(n):
        v = w.copy()
        for _ in range(rng.integers(1, 4)):
            pos = int(rng.integers(2, payload if rng.random() < 0.7 else len(w)))
            mode = rng.integers(0, 5)
            with np.errstate(over="ignore"):
                if mode == 0:
                    v[pos] = rng.integers(-5, 5)
                elif mode == 1:
                    v[pos] += rng.integers(-3, 4)
                elif mode == 2:
                    v[pos] = rng.integers(0, 1 << 20)
                elif mode == 3:
                    v[pos] = np.int64(rng.integers(-(1 << 62), 1 << 62))
                else:
                    v[pos] = -v[pos]
        yield v


def main():
    seed, n, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    os.makedirs(out, exist_ok=True)
    rng = np.random.default_rng(seed)
    for name in NAMES:
        w = np.frombuffer(blob_of(name), dtype=np.int64).copy()
        for i, v in enumerate(mutations(w, rng, n)):
            v.tofile(os.path.join(out, f"{name.replace(':', '_')}_{i:04d}.bin"))


if __name__ == "__main__":
    main()
