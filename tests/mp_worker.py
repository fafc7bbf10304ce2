"""Worker of tests/test_multiprocess.py: one rank of the N > 1 path on CPU (gloo).

Mirrors bench.py's multi-GPU step: units (windows) are sharded with ebo_shard_range,
every rank evaluates only its shard, one all-gather publishes (r, J0, J1) to all ranks.
There is no GPU here, so the per-shard evaluation is done by the CPU oracle — what is
under test is the sharding + collective layout, which is identical on the GPU."""
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc  # noqa: E402

ebo = importlib.import_module("event-based-odomety_amd")
synth = importlib.import_module("event-based-odomety_amd.synth")


def main():
    out_path = sys.argv[1]
    n_windows = int(sys.argv[2])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    prm = orc.default_params(loss=1, tv_weight=0.0)
    P = 108
    b, e = ebo.shard_range(n_windows, rank, world)
    per = (n_windows + world - 1) // world  # all_gather needs equal shards: pad
    local = np.zeros((per, P, 3))
    for k, w in enumerate(range(b, e)):
        ev, gt = synth.make_window(0, window=w, n_events=4000)
        r, J, _, _ = orc.window_eval(ev, prm, gt * 0.5)
        local[k, :, 0] = r
        local[k, :, 1:] = J
    t_local = torch.from_numpy(local)
    gathered = torch.zeros((world * per, P, 3), dtype=torch.float64)
    dist.all_gather_into_tensor(gathered, t_local)
    # drop the padding rows: rank q contributed shard_range(q) real rows
    rows = []
    for q in range(world):
        qb, qe = ebo.shard_range(n_windows, q, world)
        rows.append(gathered[q * per:q * per + (qe - qb)])
    full = torch.cat(rows).numpy()
    # per-window solved flows gathered the same way (EBO_SOLVE_INDEPENDENT's exchange)
    flows_local = np.zeros((per, P, 2))
    for k, w in enumerate(range(b, e)):
        flows_local[k] = w + 0.5  # stand-in payload: identifies the window
    g2 = torch.zeros((world * per, P, 2), dtype=torch.float64)
    dist.all_gather_into_tensor(g2, torch.from_numpy(flows_local))
    dist.barrier()
    if rank == 0:
        np.save(out_path, full)
        np.save(out_path + ".flows.npy", g2.numpy())
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
