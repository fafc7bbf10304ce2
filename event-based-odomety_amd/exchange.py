"""The two exchanges of the multi-GPU layout over torch.distributed — backend
"nccl" is RCCL over xGMI on MI355X, "gloo" rehearses the same layout on CPUs.  Host plumbing
for bench.py and tests; C++ callers without a framework use ebo_allgather_device /
ebo_allgather_tracks of include/ebo.h, which do the same thing with the library's own RCCL
communicator.

  allgather_rows    config 4: one window's patch rows are sharded over ranks; ONE all-gather of
                    the solved flows ([rows][2] doubles per rank, max-padded when the rows do not
                    divide evenly) gives every rank the whole window's flows in patch order.
  allgather_tracks  config 5: independent sequences, one per GPU; per-rank variable-length lists
                    of (id, t_us, x, y) track records (tools/evaluator/src/evaluator.cpp:125-150)
                    -> every rank gets all of them, rank 0's first: an all-gather of the counts,
                    then ONE all-gather of max-padded 32-byte records.
"""
import numpy as np

TRACK_DTYPE = np.dtype([("id", "<i8"), ("t_us", "<i8"), ("x", "<f8"), ("y", "<f8")])


def shard_counts(n_units, world):
    """Units per rank under ebo_shard_range (contiguous, the first n % world ranks get one more)."""
    base, rem = divmod(int(n_units), int(world))
    return [base + (1 if r < rem else 0) for r in range(world)]


def allgather_rows(t_local, counts, out=None, group=None):
    """t_local: torch tensor [counts[rank], ...] on the collective's device; returns the
    concatenation over ranks [sum(counts), ...].  Equal counts: one all_gather_into_tensor straight
    into the result; unequal: the same call on max-padded blocks, padding dropped afterwards."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    assert len(counts) == world and t_local.shape[0] == counts[rank]
    tail = tuple(t_local.shape[1:])
    mx = max(counts)
    if min(counts) == mx:
        if out is None:
            out = torch.empty((world * mx,) + tail, dtype=t_local.dtype, device=t_local.device)
        dist.all_gather_into_tensor(out, t_local.contiguous(), group=group)
        return out
    pad = torch.zeros((mx,) + tail, dtype=t_local.dtype, device=t_local.device)
    pad[: counts[rank]] = t_local
    full = torch.empty((world * mx,) + tail, dtype=t_local.dtype, device=t_local.device)
    dist.all_gather_into_tensor(full, pad, group=group)
    res = torch.cat([full[q * mx: q * mx + counts[q]] for q in range(world)])
    if out is not None:
        out.copy_(res)
        return out
    return res


def allgather_tracks(local, device="cpu", group=None):
    """local: numpy structured array (TRACK_DTYPE) of this rank's track points, any length.
    Returns (all records in rank order, counts per rank)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    local = np.ascontiguousarray(local, dtype=TRACK_DTYPE)
    cnt = torch.tensor([len(local)], dtype=torch.int64, device=device)
    cnts = torch.zeros(world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(cnts, cnt, group=group)
    counts = [int(v) for v in cnts.cpu().tolist()]
    mx = max(counts)
    if mx == 0:
        return np.zeros(0, dtype=TRACK_DTYPE), counts
    send = np.zeros(mx, dtype=TRACK_DTYPE)  # max-padded
    send[: len(local)] = local
    t_send = torch.from_numpy(send.view(np.uint8).reshape(mx, 32)).to(device)
    t_recv = torch.empty((world * mx, 32), dtype=torch.uint8, device=device)
    dist.all_gather_into_tensor(t_recv, t_send, group=group)
    got = t_recv.cpu().numpy().reshape(world, mx * 32)
    parts = [got[q, : counts[q] * 32].copy().view(TRACK_DTYPE) for q in range(world)]
    return np.concatenate(parts), counts
