"""dump the edge-loss (r, J) of a fixed batch for bit comparison of two builds: dump_edge.py <config> <windows> <out.npy>"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
ebo = importlib.import_module("event-based-odomety_amd")
synth = importlib.import_module("event-based-odomety_amd.synth")
config, windows, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
cfg = synth.CONFIGS[config]
ev, offsets, gt = synth.make_stream(config, windows)
with ebo.Context(image_w=cfg["image"][0], image_h=cfg["image"][1], patch_w=cfg["patch"][0], patch_h=cfg["patch"][1],
                 loss=ebo.LOSS_EDGE, tv_weight=0.0, max_events=len(ev), max_windows=windows) as c:
    c.set_windows(ev, offsets)
    res = []
    for scale in (0.0, 0.5, 1.0):
        r, J = c.eval(gt * scale)
        res.append(np.concatenate([r[..., None], J], axis=-1))
np.save(out, np.stack(res))
