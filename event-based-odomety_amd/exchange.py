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


def halo_exchange(top, bottom, from_above, from_below, flag, group=None):
    """The band-limited final image's exchange over torch.distributed (what ebo_band_exchange_device does on the
    library's own communicator): `top` goes to rank - 1, `bottom` to rank + 1, the neighbours' halos arrive in
    `from_above` / `from_below` (any of them None or empty: nothing travels that way), then `flag` (one int32) becomes
    the maximum over the ranks.  Tensors on the collective's device (GPU for "nccl", CPU for "gloo")."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    ops = []

    def some(t):
        return t is not None and t.numel() > 0
    if rank > 0 and some(top):
        ops.append(dist.P2POp(dist.isend, top, rank - 1, group))
    if rank < world - 1 and some(bottom):
        ops.append(dist.P2POp(dist.isend, bottom, rank + 1, group))
    if rank > 0 and some(from_above):
        ops.append(dist.P2POp(dist.irecv, from_above, rank - 1, group))
    if rank < world - 1 and some(from_below):
        ops.append(dist.P2POp(dist.irecv, from_below, rank + 1, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
    return flag


FAILED_ID = b"EBO-NO-ID"  # what rank 0 hands over when it could not make an id: the others fail at once instead of waiting


def handover_bytes(prefix, rank, world, payload=None, timeout=120.0, not_before=None):
    """Rank 0's `payload` (bytes: the 128-byte id of ebo_comm_unique_id) on every rank of one node, through the
    file `prefix`.id written atomically by rank 0 -- no framework, no sockets.  Then every rank leaves a mark
    (`prefix`.got.<rank>); rank 0 removes the files once all marks are there.  not_before (seconds since the epoch,
    e.g. the time this process started): a file older than that -- the id OR a mark -- is what a crashed earlier run
    left behind under the same name and is ignored (rank 0 replaces the id and removes stale marks before it writes).
    Rank 0 ALWAYS publishes something: payload None / FAILED_ID says "no id" and every rank raises RuntimeError at
    once (the caller votes `no`), instead of the other ranks waiting out their timeout."""
    import os
    import time
    path = prefix + ".id"

    def fresh(p):
        try:
            return os.path.exists(p) and (not_before is None or os.path.getmtime(p) >= not_before)
        except OSError:
            return False
    if rank == 0:
        for q in range(world):  # marks of a crashed run must not let this run delete the id before a slow rank read it
            try:
                os.remove("%s.got.%d" % (prefix, q))
            except OSError:
                pass
        tmp = path + ".tmp.%d" % os.getpid()
        with open(tmp, "wb") as fp:
            fp.write(payload if payload is not None else FAILED_ID)
        os.replace(tmp, path)
    t0 = time.time()
    while not fresh(path):
        if time.time() - t0 > timeout:
            raise TimeoutError("no %s after %.0f s (rank 0 never handed its id over)" % (path, timeout))
        time.sleep(0.01)
    with open(path, "rb") as fp:
        got = fp.read()
    open("%s.got.%d" % (prefix, rank), "wb").close()
    if rank == 0:
        while not all(fresh("%s.got.%d" % (prefix, q)) for q in range(world)):
            if time.time() - t0 > timeout:
                missing = [q for q in range(world) if not fresh("%s.got.%d" % (prefix, q))]
                raise TimeoutError("rank(s) %s never read %s within %.0f s" % (missing, path, timeout))
            time.sleep(0.01)
        for q in range(world):
            os.remove("%s.got.%d" % (prefix, q))
        os.remove(path)
    if got == FAILED_ID:
        raise RuntimeError("rank 0 could not make a communicator id")
    return got


_AGREE_ROUND = {}


def agree(prefix, rank, world, ok, timeout=120.0, not_before=None):
    """Every rank's yes / no on every rank (files `prefix`.ok.<round>.<rank>): True when all said yes.  For decisions
    every rank must take the same way BEFORE any collective exists (does the library's communicator work here?).
    The round number (how often this process has voted under this prefix: the same on every rank) keeps a second
    vote from reading the first one's files; a rank removes its own file of the round before.  TimeoutError names
    the rank that never voted."""
    import os
    import time
    rnd = _AGREE_ROUND.get(prefix, 0)
    _AGREE_ROUND[prefix] = rnd + 1
    mine = "%s.ok.%d.%d" % (prefix, rnd, rank)
    if rnd > 0:
        try:
            os.remove("%s.ok.%d.%d" % (prefix, rnd - 1, rank))
        except OSError:
            pass
    tmp = mine + ".tmp"
    with open(tmp, "w") as fp:
        fp.write("1" if ok else "0")
    os.replace(tmp, mine)
    t0 = time.time()
    votes = []
    for q in range(world):
        p = "%s.ok.%d.%d" % (prefix, rnd, q)
        while not (os.path.exists(p) and (not_before is None or os.path.getmtime(p) >= not_before)):
            if time.time() - t0 > timeout:
                raise TimeoutError("rank %d never voted within %.0f s (%s)" % (q, timeout, p))
            time.sleep(0.01)
        with open(p) as fp:
            votes.append(fp.read().strip() == "1")
    return all(votes)
