"""Worker of tests/test_tracks.py: one rank of config 5's exchange on CPU (gloo).  Every rank
builds the track list of ITS sequence (a different number of records per rank, rank 1 possibly
none), the ranks gather them (exchange.allgather_tracks: counts + ONE max-padded all-gather),
and every rank writes what it received."""
import importlib
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

exchange = importlib.import_module("event-based-odomety_amd.exchange")


def tracks_of_rank(rank, sizes):
    """Deterministic per-sequence tracks: sizes[rank] patches' worth of points."""
    rng = np.random.default_rng(1000 + rank)
    n_pts = sizes[rank]
    pts = np.zeros(n_pts, dtype=exchange.TRACK_DTYPE)
    pts["id"] = rng.integers(0, 50, n_pts)
    pts["t_us"] = np.sort(rng.integers(0, 10**9, n_pts))
    pts["x"] = rng.uniform(0, 346, n_pts)
    pts["y"] = rng.uniform(0, 260, n_pts)
    return pts


def main():
    out_dir = sys.argv[1]
    sizes = [int(v) for v in sys.argv[2].split(",")]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    assert world == len(sizes)
    mine = tracks_of_rank(rank, sizes)
    got, counts = exchange.allgather_tracks(mine)
    assert counts == sizes, (counts, sizes)
    # padded flows gather of a window whose patch rows do not divide evenly (config 4 layout)
    import torch
    rows = exchange.shard_counts(7, world)
    b = sum(rows[:rank])
    local = torch.arange(b * 4, (b + rows[rank]) * 4, dtype=torch.float64).reshape(rows[rank], 2, 2)
    full = exchange.allgather_rows(local, rows)
    assert torch.equal(full, torch.arange(0, 7 * 4, dtype=torch.float64).reshape(7, 2, 2))
    np.save(os.path.join(out_dir, "tracks_rank%d.npy" % rank), got)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
