"""A/B on the GPU box: evaluation implementations (EBO_EVAL_IMPL 0/1/2), LDS size and
workgroup size, value+Jacobian and value-only.  No oracle involved."""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
# the switches this tool flips (EBO_*) exist only in the A/B build (make ab; csrc/ab_env.h)
os.environ.setdefault("EBO_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "libebo_hip_ab.so"))
ebo = importlib.import_module("event-based-odomety_amd")
synth = importlib.import_module("event-based-odomety_amd.synth")


def setup(config, windows):
    cfg = synth.CONFIGS[config]
    evs, gts = [], []
    for w in range(windows):
        e, g = synth.make_window(config, window=w)
        evs.append(e)
        gts.append(g)
    offsets = np.zeros(windows + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum([len(e) for e in evs])
    ev = np.concatenate(evs)
    ctx = ebo.Context(image_w=cfg["image"][0], image_h=cfg["image"][1], patch_w=cfg["patch"][0],
                      patch_h=cfg["patch"][1], loss=ebo.LOSS_VARIANCE, tv_weight=0.0,
                      max_events=len(ev), max_windows=windows)
    stream = torch.cuda.current_stream()
    ctx.set_stream(stream.cuda_stream)
    ctx.set_windows(ev, offsets)
    d_flows = torch.from_numpy(np.stack(gts) * 0.5).to("cuda")
    d_out = torch.zeros((windows * ctx.P, 3), dtype=torch.float64, device="cuda")
    return ctx, stream, d_flows, d_out, len(ev)


def timeit(ctx, stream, d_flows, d_out, jac, steps=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(2):
        ctx.eval_device(d_flows.data_ptr(), jac, d_out.data_ptr())
    torch.cuda.synchronize()
    e0.record(stream)
    for _ in range(steps):
        ctx.eval_device(d_flows.data_ptr(), jac, d_out.data_ptr())
    e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


def main():
    config = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    for windows in (128, 1):
        ctx, stream, d_flows, d_out, n = setup(config, windows)
        ref = None
        for impl, kbs, tiles_list in ((0, (0,), (1, 3)), (1, (16, 32, 64), (0,)), (2, (16, 32, 64), (0, 1, 2, 4, 8) if windows == 1 else (0,))):
            for kb in kbs:
                for tiles in tiles_list:
                    for block in (128, 256, 512):
                        os.environ["EBO_EVAL_IMPL"] = str(impl)
                        os.environ["EBO_LDS_KB"] = str(kb or 32)
                        os.environ["EBO_EVAL_TILES"] = str(tiles)
                        os.environ["EBO_EVAL_BLOCK"] = str(block)
                        try:
                            mj = timeit(ctx, stream, d_flows, d_out, 1)
                            out = d_out.cpu().numpy().copy()
                            mv = timeit(ctx, stream, d_flows, d_out, 0)
                            if ref is None:
                                ref = out
                            err = np.abs(out - ref).max() / np.abs(ref).max()
                            print("cfg %d win %3d impl %d lds %2dKB tiles %d block %3d : jac %7.3f ms %8.1f Mev/s | value %7.3f ms %8.1f Mev/s | max rel diff vs first %.1e"
                                  % (config, windows, impl, kb, tiles, block, mj, n / mj / 1e3, mv, n / mv / 1e3, err), flush=True)
                        except ebo.EboError as e:
                            print("impl %d kb %d tiles %d block %d: %s" % (impl, kb, tiles, block, e))
        ctx.close()


if __name__ == "__main__":
    main()
