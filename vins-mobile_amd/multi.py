"""Replica-parallel helpers for the throughput mode (a trajectory does not shard; independent sequences
are partitioned over one process per GPU: global sequence id % world == rank). No data-path collective: a barrier, a
max-over-ranks of the step time and -- as evidence that the replicas computed what one GPU computes -- an all_gather of
each rank's result poses of one sequence, over RCCL on GPUs (backend "nccl") or gloo in CPU tests."""


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


def all_gather_array(dist, arr, device="cpu"):
    """Every rank's 1-D float64 array (same length on all ranks) on every rank: the result poses of a sequence are a few
    hundred bytes (gather / all_gather of P x 7 poses + stats per sequence)."""
    import numpy as np
    import torch
    t = torch.as_tensor(np.asarray(arr, np.float64), device=device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [o.cpu().numpy() for o in out]


def per_rank_values(dist, value, device="cpu"):
    """One float per rank, gathered on every rank (rank order): e.g. every rank's own ms_per_step next to the max that
    defines the job's step time."""
    return [float(a[0]) for a in all_gather_array(dist, [float(value)], device=device)]


def replica_report(dist, sequences_per_gpu, steps, seconds, device="cpu"):
    """The throughput line of a replica-parallel run at one batch size: whole-job frames/s = all ranks' frames / slowest
    rank's time, plus every rank's own step time (bench.py emits it for the headline batch AND for BASELINE.json's
    configs[3] read literally, 8 sequences per GPU, so that one launch of the job yields both)."""
    world = dist.get_world_size()
    per_rank = per_rank_values(dist, seconds / steps * 1e3, device=device)
    worst = max_over_ranks(dist, seconds, device=device)
    return {"sequences_per_gpu": int(sequences_per_gpu), "n_gpus": world, "value": sequences_per_gpu * world * steps / worst,
            "unit": "frames/s", "ms_per_step": worst / steps * 1e3, "per_rank_ms_per_step": per_rank}
