"""Worker of tests/test_gpu_multiprocess.py: one rank of the config-4 layout THROUGH THE HIP PATH.
All ranks share the box's one GPU (RCCL refuses two ranks on one device, so the collective runs
over gloo on host copies; layout, sharding and every kernel are the ones an 8-GPU node runs).

Rank r loads the patches of its grid rows of every window with the events inside them
(ebo_shard_range + ebo_set_patches), solves them on the device (ebo_solve_device), and ONE
all-gather gives every rank all flows.  Every rank saves what it received.  Then the last third of
compensateEventsContrast (feature_detector.cpp:433-463): every rank counts ITS events warped by the
gathered flows at the window's reference time (ebo_count_image_shard) and ONE reduce sums the
integer-valued partial images; rank 0 saves the result."""
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ebo = importlib.import_module("event-based-odomety_amd")
synth = importlib.import_module("event-based-odomety_amd.synth")
exchange = importlib.import_module("event-based-odomety_amd.exchange")
import bench  # noqa: E402  (bucket_rows: the same sharding code the benchmark uses)


def main():
    out_dir, config, n_windows, iters = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    cfg = synth.CONFIGS[config]
    iw, ih = cfg["image"]
    pw, ph = cfg["patch"]
    npx, npy = iw // pw, ih // ph
    rows = exchange.shard_counts(npy, world)
    b, e = ebo.shard_range(npy, rank, world)
    _, _, rects = synth.grid_rects(cfg["image"], cfg["patch"])
    my_rects = rects[b * npx:e * npx]
    evs, cnts, t_ref = [], [], []
    for w in range(n_windows):
        ev, _ = synth.make_window(config, window=w, n_events=min(cfg["events"], 40000))
        # the WINDOW's reference time: from its first / last event, known to whoever cut the window
        t_ref.append(ebo.window_ref_time(ev["t_us"][0], ev["t_us"][-1]))
        ev_w, c_w = bench.bucket_rows(ev, cfg, b, e)
        evs.append(ev_w)
        cnts.append(c_w)
    ev = np.concatenate(evs)
    offs = np.zeros(n_windows * len(my_rects) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(np.concatenate(cnts))
    n_units = n_windows * len(my_rects)
    with ebo.Context(device=0, image_w=iw, image_h=ih, patch_w=pw, patch_h=ph, loss=ebo.LOSS_VARIANCE,
                     tv_weight=0.0, max_events=len(ev), max_windows=n_windows) as c:
        c.set_stream(torch.cuda.current_stream().cuda_stream)
        c.set_patches(ev, offs, np.tile(my_rects, (n_windows, 1)))
        d_sol = torch.zeros((n_units, 2), dtype=torch.float64, device="cuda")
        c.solve_device(ebo.default_solver(mode=ebo.SOLVE_INDEPENDENT, max_num_iterations=iters), d_sol.data_ptr())
        torch.cuda.synchronize()
        counts = [n_windows * q * npx for q in rows]
        full = exchange.allgather_rows(d_sol.cpu(), counts)
        # rank q's block is [window][its rows][px]; reorder to [window][row][px] = patch order
        parts = []
        at = 0
        for q in range(world):
            parts.append(full[at:at + counts[q]].reshape(n_windows, rows[q] * npx, 2))
            at += counts[q]
        flows = torch.cat(parts, dim=1).contiguous()
        np.save(os.path.join(out_dir, "flows_rank%d.npy" % rank), flows.numpy())
        # the final image: partial image of this rank's events on the device, one reduce
        d_flows = flows.to("cuda")
        d_img = torch.zeros((n_windows, ih, iw), dtype=torch.float64, device="cuda")
        c.count_image_shard_device(n_windows, t_ref, d_flows.data_ptr(), d_img.data_ptr())
        torch.cuda.synchronize()
        img = d_img.cpu()
        dist.reduce(img, dst=0, op=dist.ReduceOp.SUM)
        if rank == 0:
            np.save(os.path.join(out_dir, "image_rank0.npy"), img.numpy())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
