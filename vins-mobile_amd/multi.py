"""Replica-parallel helpers for the throughput mode (a trajectory does not shard; independent sequences
are partitioned over one process per GPU). No data-path collective: only a barrier and a max-over-ranks of the step
time, over RCCL on GPUs (backend "nccl") or gloo in CPU tests."""


def sequences_of_rank(n_total, rank, world):
    """Global sequence ids owned by `rank`: sequence_id % world == rank."""
    return list(range(rank, n_total, world))


def seed_of_sequence(sequence_id):
    """seed = 42 + sequence_id."""
    return 42 + sequence_id


def max_over_ranks(dist, value, device="cpu"):
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def aggregate_rate(dist, local_units, seconds, device="cpu"):
    """Whole-job throughput: units of all ranks / max time over ranks."""
    import torch
    u = torch.tensor([float(local_units)], dtype=torch.float64, device=device)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(u.item()) / max_over_ranks(dist, seconds, device)
