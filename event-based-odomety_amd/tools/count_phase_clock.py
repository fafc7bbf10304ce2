"""Phase clocks of k_count_tiles (warped count image) on the GPU box, from the instrumented build
(libebo_hip_prof.so, -DEBO_EDGE_TIMING).  usage: count_phase_clock.py <config> <windows>"""
import ctypes as C, importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import os
# the switches this tool flips (EBO_*) exist only in the A/B build (make ab; csrc/ab_env.h)
os.environ.setdefault("EBO_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "libebo_hip_ab.so"))
ebo = importlib.import_module("event-based-odomety_amd")
ebo.LIB_PATH = os.path.join(os.path.dirname(ebo.LIB_PATH), "libebo_hip_prof.so")
synth = importlib.import_module("event-based-odomety_amd.synth")
config, windows = int(sys.argv[1]), int(sys.argv[2])
cfg = synth.CONFIGS[config]
ev, offsets, gt = synth.make_stream(config, windows)
ctx = ebo.Context(image_w=cfg["image"][0], image_h=cfg["image"][1], patch_w=cfg["patch"][0], patch_h=cfg["patch"][1],
                  loss=ebo.LOSS_VARIANCE, max_events=len(ev), max_windows=windows)
lib = C.CDLL(ebo.LIB_PATH)
lib.ebo_debug_edge_clocks.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
stream = torch.cuda.current_stream(); ctx.set_stream(stream.cuda_stream); ctx.set_windows(ev, offsets)
d_flows = torch.from_numpy(gt * 0.8).to("cuda")
d_img = torch.zeros((windows, cfg["image"][1], cfg["image"][0]), dtype=torch.float64, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(40):
    ctx.count_image_device(ebo.COUNT_WARPED, d_flows.data_ptr(), d_img.data_ptr())
torch.cuda.synchronize()
buf = (C.c_ulonglong * 32)()
lib.ebo_debug_edge_clocks(buf, 1)
reps = 10
e0.record(stream)
for _ in range(reps):
    ctx.count_image_device(ebo.COUNT_WARPED, d_flows.data_ptr(), d_img.data_ptr())
e1.record(stream); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
lib.ebo_debug_edge_clocks(buf, 0)
v = np.array(list(buf), dtype=np.float64)
wgs = v[23] / reps
clk = v[16:22] / v[23]
names = ["select units", "zero counters", "stream events (thread 0's wave)", "  wait for the other waves", "store", "  wait for the other waves"]
algo = 8 * len(ev) + d_img.numel() * 8
print("cfg %d win %d: %.3f ms per launch (%.1f %% of 8 TB/s algorithmic), %d workgroups, %.1f units per workgroup, %.0f clocks per workgroup"
      % (config, windows, ms, algo / ms / 1e6 / 80, wgs, v[22] / v[23], clk.sum()))
for n, c in zip(names, clk):
    print("   %-34s %9.0f clk  %5.1f %%" % (n, c, 100 * c / clk.sum()))
