#!/usr/bin/env python3
"""Does a model's run-time source hit the in-tree code-object cache (built offline by build())?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from infiniteexamodels.jl_amd import lib as iemlib, transcribe, workloads
from infiniteexamodels.jl_amd.model import ExaModel

def probe(name, blob, make):
    src, key = iemlib.emit_source(blob)
    path = os.path.join(iemlib.KERNEL_DIR, f"iem_{key:016x}.hsaco")
    before = os.path.exists(path)
    size = os.path.getsize(path) if before else 0
    gm = make()
    print(f"{name:34s} key {key:016x} offline object {'present' if before else 'ABSENT'} ({size} B) -> jit={gm.kernels()[0]['jit']}", flush=True)
    gm.close()

g = transcribe.exa_core(workloads.quadrotor(100)).to_blob()
probe("quadrotor 100", g, lambda: ExaModel.from_blob(g))
g2 = transcribe.exa_core(workloads.quadrotor(100_000)).to_blob()
probe("quadrotor 1e5", g2, lambda: ExaModel.from_blob(g2))
g3 = transcribe.exa_core(workloads.quadrotor(1_000_000)).to_blob()
for r, w in ((3, 8), (0, 8)):
    lb = iemlib.shard_blob(g3, 1, r, w)[0]
    probe(f"shard {r}/{w} of 1e6 (from its blob)", lb, lambda: ExaModel.from_blob(lb))
    probe(f"shard {r}/{w} of 1e6 (create_sharded)", lb, lambda: ExaModel.sharded(g3, 1, r, w))
