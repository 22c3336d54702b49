"""Sequence sharding across the GPUs of one node.

A tracker is strictly sequential in time and batch-1, so the only parallel axis is *independent
sequences*: sequence s runs on rank s % world, one process per GPU, no data-path collective.  The one
collective of the whole path is the gather of per-sequence metrics at the end (RCCL on GPUs, gloo in
the CPU tests); payload is a few floats per rank, i.e. latency-bound -- xGMI bandwidth is irrelevant.
"""
import torch
import torch.distributed as dist


def my_sequences(n_sequences, rank, world):
    return [s for s in range(n_sequences) if s % world == rank]


def gather_metrics(values, device):
    """values: list of python floats for this rank -> [world, len(values)] tensor on every rank"""
    if dist.is_available() and dist.is_initialized() and dist.get_backend() == "gloo":
        device = "cpu"            # (gloo gathers host tensors only: the CPU tests, and bench.py's two-ranks-on-one-GPU rehearsal)
    mine = torch.tensor(values, dtype=torch.float64, device=device)
    if not (dist.is_available() and dist.is_initialized()):
        return mine[None]
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return torch.stack(out, 0)


def max_over_ranks(seconds, device):
    if dist.is_available() and dist.is_initialized() and dist.get_backend() == "gloo":
        device = "cpu"
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized():      # (a world of one still runs the collective: RCCL smoke)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
