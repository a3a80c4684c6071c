"""Malformed blobs must be rejected with an error code (or accepted, when the mutation happens to
be harmless) — never crash the host process: the C-ABI takes blobs from foreign producers (the
Julia writer).  Runs in a child process so that a crash is a test failure, not a dead pytest;
`tools/fuzz_blob.sh` runs the same mutations under ASan/UBSan."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys
sys.path.insert(0, {root!r}); sys.path.insert(0, {root!r} + '/tests'); sys.path.insert(0, {root!r} + '/oracle'); sys.path.insert(0, {root!r} + '/tools')
import numpy as np
import cases, fuzz_blob_gen as fg
from infiniteexamodels.jl_amd import lib as L
rng = np.random.default_rng(11)
ok = rej = sh_ok = sh_rej = 0
for name in fg.NAMES:
    w = np.frombuffer(fg.blob_of(name), dtype=np.int64).copy()
    for v in fg.mutations(w, rng, 40):
        b = v.tobytes()
        try:
            L.emit_source(b); L.emit_launch_plan(b); L.blob_hess_structure(b)
            ok += 1
        except L.IemError:
            rej += 1
        for rank, world in ((0, 2), (1, 2), (2, 3)):      # the window cut (slab table, re-based indices, serialiser)
            try:
                lb = L.shard_blob(b, 1, rank, world)[0]
                L.emit_source(lb)
                sh_ok += 1
            except L.IemError:
                sh_rej += 1
    for cut in (8, 14 * 8, len(w) * 4, len(w) * 8 - 8):      # truncations
        try:
            L.emit_source(w.tobytes()[:cut]); ok += 1
        except L.IemError:
            rej += 1
print("FUZZ", sh_ok, sh_rej, len(fg.NAMES), ok, rej)
"""


def test_mutated_blobs_never_crash_the_host():
    r = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, f"child died ({r.returncode}):\n{r.stderr[-2000:]}"
    sh_ok, sh_rej, len_names, ok, rej = (int(v) for v in r.stdout.strip().split()[-5:])
    assert rej > 50 and ok + rej == len_names * 44
    assert sh_ok > 50 and sh_rej > 50 and sh_ok + sh_rej == len_names * 40 * 3
