"""A/B on the GPU box: edge-loss evaluation time for env settings. usage: ab_edge.py <config> <windows> "K=V,.." ..."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import os
# the switches this tool flips (EBO_*) exist only in the A/B build (make ab; csrc/ab_env.h)
os.environ.setdefault("EBO_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "libebo_hip_ab.so"))
ebo = importlib.import_module("event-based-odomety_amd")
synth = importlib.import_module("event-based-odomety_amd.synth")
config, windows = int(sys.argv[1]), int(sys.argv[2])
cfg = synth.CONFIGS[config]
ev, offsets, gt = synth.make_stream(config, windows)
ctx = ebo.Context(image_w=cfg["image"][0], image_h=cfg["image"][1], patch_w=cfg["patch"][0], patch_h=cfg["patch"][1],
                  loss=ebo.LOSS_EDGE, tv_weight=0.0, max_events=len(ev), max_windows=windows)
stream = torch.cuda.current_stream(); ctx.set_stream(stream.cuda_stream); ctx.set_windows(ev, offsets)
d_flows = torch.from_numpy(gt * float(os.environ.get("EBO_AB_FLOWSCALE", "0.5"))).to("cuda")
d_out = torch.zeros((windows * ctx.P, 3), dtype=torch.float64, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
# a single setting is applied BEFORE the warm-up, so that every dispatch of the process runs under it (counter passes
# average over all dispatches: tools/ab/edge_phase_mix.sh)
if len(sys.argv) == 4:
    for kv in filter(None, sys.argv[3].split(",")):
        k, v = kv.split("="); os.environ[k] = v
# clocks ramp up during the first few hundred milliseconds: without this the FIRST setting reads ~5 % slow
for _ in range(200):
    ctx.eval_device(d_flows.data_ptr(), 1, d_out.data_ptr())
torch.cuda.synchronize()
ref = None
for st in sys.argv[3:] or [""]:
    for k in [k for k in os.environ if k.startswith("EBO_EDGE_")]:  # every switch of the setting before
        os.environ.pop(k, None)
    for kv in filter(None, st.split(",")):
        k, v = kv.split("="); os.environ[k] = v
    res = []
    for jac in (1, 0):
        for _ in range(2):
            ctx.eval_device(d_flows.data_ptr(), jac, d_out.data_ptr())
        best = None
        for _ in range(3):  # best of three batches of 5 launches
            torch.cuda.synchronize(); e0.record(stream)
            for _ in range(5):
                ctx.eval_device(d_flows.data_ptr(), jac, d_out.data_ptr())
            e1.record(stream); torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / 5
            best = t if best is None else min(best, t)
        res.append(best)
        if jac:
            out = d_out.cpu().numpy().copy()
    ref = out if ref is None else ref
    print("cfg %d win %d [%-34s] jac %8.3f ms %7.0f Mev/s | val %8.3f ms %7.0f Mev/s | d=%.0e"
          % (config, windows, st, res[0], len(ev) / res[0] / 1e3, res[1], len(ev) / res[1] / 1e3,
             np.abs(out - ref).max() / np.abs(ref).max()), flush=True)
