"""Worker of tests/test_gpu_multiprocess.py: the REFERENCE-FAITHFUL TV mode across ranks (SURVEY 8(e)
"Collective": per LM evaluation one all-gather of (r, J0, J1) = 24 B per patch, the same host solver
replicated on every rank).  Rank r holds the patches of its grid rows (ebo_set_patches), evaluates their
data terms on the device at the point the solver asks for (ebo_eval: the reference's edge loss), ONE
all-gather per evaluation (gloo here: the ranks share the box's one GPU), and every rank's ebo_lm takes the
same step.  Every rank saves its flows."""
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
import bench  # noqa: E402  (bucket_rows)


def main():
    out_dir, config, loss = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    cfg = synth.CONFIGS[config]
    iw, ih = cfg["image"]
    pw, ph = cfg["patch"]
    npx, npy = iw // pw, ih // ph
    P = npx * npy
    rows = exchange.shard_counts(npy, world)
    b, e = ebo.shard_range(npy, rank, world)
    _, _, rects = synth.grid_rects(cfg["image"], cfg["patch"])
    my_rects = rects[b * npx:e * npx]
    ev, _ = synth.make_window(config, n_events=min(cfg["events"], 30000))
    ev_mine, cnts = bench.bucket_rows(ev, cfg, b, e)
    offs = np.zeros(len(my_rects) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(cnts)
    n_mine = len(my_rects)
    counts3 = [q * npx * 3 for q in rows]
    evaluations = 0
    with ebo.Context(device=0, image_w=iw, image_h=ih, patch_w=pw, patch_h=ph, loss=loss, max_events=max(len(ev_mine), 1),
                     max_windows=1) as c:
        c.set_patches(ev_mine, offs, my_rects)
        # which patches have a data term (n_events > compensateMinNumEvents): every rank learns all of them
        mine_active = torch.tensor([1.0 if c.patch_info(p)[1] else 0.0 for p in range(n_mine)], dtype=torch.float64)
        active = exchange.allgather_rows(mine_active.reshape(-1, 1), [q * npx for q in rows]).reshape(-1).numpy() > 0
        with ebo.HostSolver(npx, npy, active, tv_weight=c.params.tv_weight, tv_huber=c.params.tv_huber) as lm:
            while True:
                what, flows = lm.request()
                if what == ebo.HostSolver.DONE:
                    break
                r, J = c.eval(flows[b * npx:e * npx], want_jac=(what == ebo.HostSolver.NEED_JACOBIAN))
                triple = np.zeros((n_mine, 3))
                triple[:, 0] = r[0]
                if J is not None:
                    triple[:, 1:] = J[0]
                # THE collective of this mode: 24 B per patch per evaluation
                allt = exchange.allgather_rows(torch.from_numpy(triple.reshape(-1, 1)), counts3).reshape(P, 3).numpy()
                lm.supply(allt[:, 0], allt[:, 1:] if what == ebo.HostSolver.NEED_JACOBIAN else None)
                evaluations += 1
            flows, summ = lm.result()
    np.save(os.path.join(out_dir, "tvflows_rank%d.npy" % rank), flows)
    np.save(os.path.join(out_dir, "tvstats_rank%d.npy" % rank),
            np.array([summ.iterations, summ.termination, evaluations], dtype=np.int64))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
