"""Worker of tests/test_multiprocess.py::test_band_limited_image_over_gloo: one rank of the band-limited final
image on CPU.  The rank's band counts come from the CPU oracle (there is no GPU here: what is under
test is the plan + the halo exchange + the assembly, which are the same on the GPU, where k_count_band fills the
same three buffers); the exchange is exchange.halo_exchange over gloo."""
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
exchange = importlib.import_module("event-based-odomety_amd.exchange")


def main():
    out_path, halo, amp = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    prm = orc.default_params(loss=1, tv_weight=0.0)  # 240x180, 20x20 patches: 12 x 9 grid
    npx, npy = orc.grid(prm)
    ev, _ = synth.make_window(0, n_events=12000)
    rng = np.random.default_rng(5)
    flows = rng.uniform(-amp, amp, (npx * npy, 2))
    ih, iw, ph = 180, 240, 20
    bounds = [ebo.shard_range(npy, q, world)[0] * ph for q in range(world)] + [ih]
    band = ebo.band_plan(ih, bounds, rank, halo)
    # this rank's events: those of its grid rows; its band image = the oracle's final loop over them, cut to the band
    gy = np.clip(ev["y"] // ph, 0, npy - 1)
    b, e = ebo.shard_range(npy, rank, world)
    mine = ev[(gy >= b) & (gy < e)]
    t_ref = ebo.window_ref_time(ev["t_us"][0], ev["t_us"][-1])
    # the final loop (feature_detector.cpp:433-463) over this rank's events at the WINDOW's reference time
    gxm = np.clip(mine["x"] // 20, 0, npx - 1)
    gym = np.clip(mine["y"] // ph, 0, npy - 1)
    f = flows[gym * npx + gxm]
    dtw = (t_ref - mine["t_us"]).astype(np.float64)
    cround = lambda v: np.sign(v) * np.floor(np.abs(v) + 0.5)  # C round(): halves away from zero
    nx = cround(mine["x"] + dtw * prm.scale * f[:, 0]).astype(np.int64)
    ny = cround(mine["y"] + dtw * prm.scale * f[:, 1]).astype(np.int64)
    ok = (nx >= 0) & (nx < iw) & (ny >= 0) & (ny < ih)
    part = np.zeros((ih, iw))
    np.add.at(part, (ny[ok], nx[ok]), 1.0)
    escaped = int(part[:band.band_row0].sum() + part[band.band_row1:].sum() > 0)
    cut = lambda r0, r1: torch.from_numpy(part[r0:r1].astype(np.int32)[None]) if r1 > r0 else None
    top, own, bottom = cut(band.band_row0, band.own_row0), cut(band.own_row0, band.own_row1), cut(band.own_row1, band.band_row1)
    mk = lambda rows: torch.zeros((1, rows, iw), dtype=torch.int32) if rows else None
    from_above, from_below = mk(band.recv_above), mk(band.recv_below)
    flag = torch.tensor([escaped], dtype=torch.int32)
    exchange.halo_exchange(top, bottom, from_above, from_below, flag)
    img = own[0].numpy().astype(np.float64)
    if from_above is not None:
        img[:band.recv_above] += from_above[0].numpy()
    if from_below is not None:
        img[band.own_rows - band.recv_below:] += from_below[0].numpy()
    # assembly on rank 0: the owned rows only
    rows = [torch.zeros((bounds[q + 1] - bounds[q], iw), dtype=torch.float64) for q in range(world)]
    dist.all_gather(rows, torch.from_numpy(img)) if len({r.shape for r in rows}) == 1 else None
    if len({r.shape for r in rows}) != 1:
        mx = max(r.shape[0] for r in rows)
        pad = torch.zeros((mx, iw), dtype=torch.float64)
        pad[:img.shape[0]] = torch.from_numpy(img)
        got = [torch.zeros((mx, iw), dtype=torch.float64) for _ in range(world)]
        dist.all_gather(got, pad)
        rows = [got[q][:bounds[q + 1] - bounds[q]] for q in range(world)]
    if rank == 0:
        np.save(out_path, torch.cat(rows).numpy())
        np.save(out_path + ".whole.npy", orc.final_count_image(ev, prm, flows))
        np.save(out_path + ".flag.npy", flag.numpy())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
