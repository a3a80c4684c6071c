"""Multi-rank data path on ONE GPU (2-4 processes sharing cuda:0): iem_create_sharded handles whose
ranks own only their slice of a distributed x; the halo row arrives through iem_halo_exchange, the
objective / replicated-gradient sums through iem_allreduce_obj_grad — both over HIP-IPC mailboxes.
Reassembled cons/jac/hess equal the unsharded GPU model bit for bit and the global oracle within
1e-10 (reference stencil: /root/reference/src/transform.jl:535-557)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("name,group,size,world,mode", [
    ("quadrotor", 1, "4000", 2, "eager"), ("quadrotor", 1, "4001", 4, "graph"), ("farmer", 1, "3000", 3, "eager"),
    ("opf", 1, "500", 2, "graph"), ("pandemic", 2, "40x12", 4, "eager"), ("quadrotor_oc3", 1, "2000", 3, "eager"),
    # the asynchronous exchange (iem_halo_exchange_async): calls that read no halo entry overlap it, the others wait
    ("quadrotor", 1, "4000", 2, "async"), ("quadrotor", 1, "4001", 3, "async_graph"), ("quadrotor_oc3", 1, "2000", 2, "async"),
    ("pandemic", 2, "40x12", 2, "async")])
def test_distributed_x_halo_and_allreduce(name, group, size, world, mode, built):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "comm_worker.py"), name, str(group), size, mode],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=600)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    assert "OK" in outs[0], outs[0][-3000:]


def test_bench_line_carries_a_checked_comm_section(built):
    """bench.py at N = 2 (two ranks rehearsed on this one GPU, gloo for the host rendezvous): the JSON line reports the
    halo exchange and objective all-reduce of the C-ABI's own mailboxes — exact, timed — next to the headline value."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--same-device",
                        "--supports", "40000", "--steps", "5", "--warmup", "2", "--no-weak"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 prints ONE JSON line"
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["value"] > 0
    c = j["comm"]
    assert "error" not in c, c
    assert c["halo_exact"] is True and c["allreduce_exact"] is True
    assert c["halo_exchange_us"] > 0 and c["allreduce_obj_us"] > 0 and c["pair_behind_blocking_halo"]["ms_per_step"] > 0
    # the headline step of an N > 1 line CONTAINS the halo exchange: wired in-process after the children came back clean
    assert j["halo_in_timed_loop"] is True and "behind iem_halo_exchange_async" in j["config"]["step"]
    assert j["halo"]["status_after_timed_loop"] == 0 and j["halo"]["mailbox_kind"] in (1, 2, 3)
    assert j["halo"]["reads_halo_rank1"] == {"cons": True, "jac": False, "hess": False, "pair": False}   # difference rows are linear
    assert j["comm_path"]["path"] == "own" and j["launch"]["world_size"] == 2 and len(j["launch"]["rank_ms_per_step"]["all"]) == 2
    # the timed step is the metric's own call pair; the one-launch form is reported beside it
    assert j["pair_no_halo"]["value"] > 0 and j["fused_pair"]["value"] > 0 and "iem_jac_coord + iem_hess_coord" in j["config"]["step"]


def test_bench_line_with_the_torch_distributed_fallback_in_the_timed_step(built):
    """The same N = 2 rehearsal with the mailboxes deliberately NOT used (--force-rccl): the timed step then contains the
    torch.distributed send / recv of the halo doubles (shard.ShardComm's fallback; gloo here, RCCL on a multi-GPU node) — an
    N > 1 line never times a communication-free step silently."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--same-device", "--force-rccl",
                        "--supports", "40000", "--steps", "5", "--warmup", "2", "--no-comm-check"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert j["n_gpus"] == 2 and j["value"] > 0 and j["halo_in_timed_loop"] is True
    assert j["comm_path"]["path"] == "rccl" and j["comm_path"]["why"] == "--force-rccl"
    assert "torch.distributed send/recv" in j["config"]["step"] and j["pair_no_halo"]["value"] > j["value"] * 0.5


def test_bench_under_a_process_group_of_one(built):
    """`--gpus 1` under torchrun-style environment (RANK / WORLD_SIZE set): the process group is created, the line is the
    single-GPU line (no exchange, no comm section)."""
    import json
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--supports", "100000", "--steps", "10", "--warmup", "3",
                        "--no-cpu-baseline", "--no-variants", "--no-cold"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert j["n_gpus"] == 1 and j["value"] > 0 and j["launch"]["process_group"] is True and j["launch"]["world_size"] == 1
    assert "comm_path" not in j and j["config"]["step"] == "iem_jac_coord + iem_hess_coord"


def test_a_skipped_exchange_surfaces_as_an_error(built):
    """A rank that does not take part: the peer's kernel times out (bounded), poisons what it should have delivered and
    the next host synchronisation point returns IEM_E_COMM — never a silent stale halo (iem_device.h)."""
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "comm_timeout_worker.py")],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=300)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    assert "OK timeout surfaced" in outs[0], outs[0][-3000:]
