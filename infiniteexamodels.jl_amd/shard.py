"""Support-axis sharding for multi-GPU runs (one process per GPU) — see DESIGN.md §6."""
from __future__ import annotations


def quadrotor_shard(S_global: int, rank: int, world: int):
    raise NotImplementedError("filled in below in this round")
