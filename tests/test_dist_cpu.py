"""CPU, world_size 2, gloo: the batch-parallel sharding + disparity all-gather used by bench.py --gpus N
(nndepth_amd/parallel.py).  The GPU path uses the same code with backend "nccl" (= RCCL)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from nndepth_amd import parallel
    r, w, _ = parallel.init_distributed("gloo")
    assert (r, w) == (rank, world)
    pairs = list(parallel.shard_range(5, r, w))  # 5 pairs over 2 ranks -> 3 + 2
    # each rank "computes" a disparity map per pair (value = pair id) at a small shape
    local = torch.stack([torch.full((1, 4, 6), float(p)) for p in pairs[:2]])  # equal count per rank for gather
    out = parallel.gather_disparity(local)
    t = parallel.max_over_ranks(1.0 + rank, "cpu")
    parallel.barrier()
    q.put((rank, pairs, out[:, 0, 0, 0].tolist(), t))
    dist.destroy_process_group()


def test_shard_and_gather_world2():
    from nndepth_amd import parallel
    assert [list(parallel.shard_range(10, r, 4)) for r in range(4)] == [[0, 1, 2], [3, 4, 5], [6, 7], [8, 9]]
    assert list(parallel.shard_range(3, 0, 1)) == [0, 1, 2]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 1, 2] and res[1][1] == [3, 4]
    # all-gather returns rank-ordered concatenation on every rank
    assert res[0][2] == res[1][2] == [0.0, 1.0, 3.0, 4.0]
    assert res[0][3] == res[1][3] == 2.0  # max over ranks


def _sharded_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from nndepth_amd import parallel
    parallel.init_distributed("gloo")
    seen = []

    def load(ids):  # stands in for "decode + upload + Padder": frames that carry their pair id
        seen.append(list(ids))
        f = torch.tensor(ids, dtype=torch.float32).view(-1, 1, 1, 1).expand(-1, 3, 4, 6).contiguous()
        return f, f + 0.5

    def forward(f1, f2):  # stub model: "disparity" = pair id + 0.5 at 1 channel
        return (f1[:, :1] + f2[:, :1]) * 0.5 + 0.25

    out = parallel.sharded_inference(5, load, forward, micro_batch=2)  # 5 pairs over 2 ranks: 3 (2 + 1) and 2
    local = parallel.sharded_inference(5, load, forward, micro_batch=2, gather=False)
    parallel.barrier()
    q.put((rank, seen[:len(seen) // 2], out[:, 0, 0, 0].tolist(), tuple(out.shape), local[:, 0, 0, 0].tolist()))
    dist.destroy_process_group()


def test_sharded_inference_world2():
    """Control flow of bench.py --config kitti64 / cre8 (nndepth_amd.parallel.sharded_inference) with a stub model on gloo:
    contiguous shards, micro-batches, ragged all-gather back into pair order on every rank."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == [[0, 1], [2]] and res[1][1] == [[3, 4]]          # micro-batches of the two shards
    want = [0.5, 1.5, 2.5, 3.5, 4.5]
    assert res[0][2] == res[1][2] == want and res[0][3] == (5, 1, 4, 6)   # pair order, on every rank
    assert res[0][4] == want[:3] and res[1][4] == want[3:]


def test_single_process_is_a_noop():
    from nndepth_amd import parallel
    x = torch.arange(6.0).view(1, 1, 2, 3)
    assert parallel.gather_disparity(x) is x
    assert parallel.max_over_ranks(3.5, "cpu") == 3.5
    y = torch.arange(3.0).view(3, 1, 1, 1)
    assert parallel.gather_ragged(y, 3) is y
    got = parallel.sharded_inference(3, lambda ids: (y[ids], y[ids]), lambda a, b: a + b, micro_batch=2, rank=0, world=1)
    assert got[:, 0, 0, 0].tolist() == [0.0, 2.0, 4.0]


def test_sharded_inference_rejects_more_ranks_than_pairs_on_every_rank():
    """the world <= n_pairs check runs before any work or collective and raises the same error on every rank
    (a rank without pairs used to assert alone while the others waited in the all-gather)."""
    from nndepth_amd import parallel
    y = torch.arange(3.0).view(3, 1, 1, 1)
    for rank in range(4):
        with pytest.raises(ValueError, match="cannot shard"):
            parallel.sharded_inference(3, lambda ids: (y[ids], y[ids]), lambda a, b: a + b, micro_batch=1, rank=rank, world=4)
    # rank given, world resolved from the environment (was: world stayed None)
    got = parallel.sharded_inference(3, lambda ids: (y[ids], y[ids]), lambda a, b: a + b, micro_batch=2, rank=0)
    assert got.shape[0] == 3


@pytest.mark.parametrize("config", ["raft544", "kitti64"])
def test_bench_self_launch_world2(config):
    """`python bench.py --gpus 2` with no launcher around it (the shape of the driver's command): bench.py starts the two
    ranks itself as a child torch.distributed.run before touching the GPU; --rendezvous-only stops after the rendezvous,
    barrier and max-over-ranks on gloo (no GPU here) and rank 0's JSON line comes back through the parent."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--config", config, "--rendezvous-only"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line == {"rendezvous_only": True, "n_gpus": 2, "config": config, "max_over_ranks": 2.0}
