"""world_size-2 gloo test of the N>1 path: sequences are sharded one per rank, tracked independently
(host logic over the oracle backend, CPU), metrics gathered with the path's single collective."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _track(seed):
    from oracle.backend_cpu import Ramp_vo, cpu_oracle_ops
    from rampvo_amd.config import make_cfg
    from rampvo_amd.synthetic import SyntheticStream, make_network
    torch.manual_seed(100 + seed)
    with cpu_oracle_ops(), torch.no_grad():
        cfg = make_cfg("default", PATCHES_PER_FRAME=8, MIXED_PRECISION=False)
        slam = Ramp_vo(cfg, make_network("SingleScale", device="cpu"), {"event_bias": True}, ht=64, wd=96, device="cpu")
        for t, (im, ev, K, mask) in enumerate(SyntheticStream(64, 96, 10, seed=seed)):
            slam(t, input_tensor=(ev, im, mask), intrinsics=K)
        traj, _ = slam.terminate()
    return [float(slam.n), float(len(slam._ii)), float(np.abs(traj).sum())]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    torch.set_num_threads(2)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rampvo_amd.shard import gather_metrics, max_over_ranks, my_sequences
    seqs = my_sequences(2, rank, world)
    assert seqs == [rank]
    vals = _track(seed=seqs[0])
    g = gather_metrics(vals, "cpu")
    tmax = max_over_ranks(1.0 + rank, "cpu")
    if rank == 0:
        q.put((g.numpy(), tmax))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_shard_sequences_and_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    g, tmax = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert g.shape == (2, 3) and tmax == 2.0
    sys.path.insert(0, ROOT)
    torch.set_num_threads(2)
    for r in range(2):           # every rank's result equals a single-process run of that sequence
        assert np.allclose(g[r], _track(seed=r), rtol=0, atol=0), (r, g[r])
    assert not np.array_equal(g[0], g[1])


def test_sequence_assignment():
    from rampvo_amd.shard import my_sequences
    assert my_sequences(8, 3, 8) == [3]
    assert my_sequences(10, 1, 4) == [1, 5, 9]
    assert sorted(sum((my_sequences(11, r, 4) for r in range(4)), [])) == list(range(11))
