"""Phase clocks of k_eval_edge on the GPU box, from the instrumented build
(make -C event-based-odomety_amd/csrc OUT=../libebo_hip_prof.so FLAGS="... -DEBO_EDGE_TIMING").
usage: edge_phase_clock.py <config> <windows>
Prints, per phase, the shader-clock cycles thread 0 of a workgroup spends between two barrier-separated
points, averaged over the active units of one launch."""
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
                  loss=ebo.LOSS_EDGE, tv_weight=0.0, max_events=len(ev), max_windows=windows)
lib = C.CDLL(ebo.LIB_PATH)
lib.ebo_debug_edge_clocks.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
stream = torch.cuda.current_stream(); ctx.set_stream(stream.cuda_stream); ctx.set_windows(ev, offsets)
d_flows = torch.from_numpy(gt * 0.5).to("cuda")
d_out = torch.zeros((windows * ctx.P, 3), dtype=torch.float64, device="cuda")
n_active = sum(1 for w in range(windows) for p in range(ctx.P) if ctx.patch_info(p, w)[1])
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(100):
    ctx.eval_device(d_flows.data_ptr(), 1, d_out.data_ptr())
torch.cuda.synchronize()
names = ["bbox", "zero", "scatter", "convert+mean", "eigen", "nms", "zeroA", "reverse", "convertA", "gather"]
buf = (C.c_ulonglong * 32)()
for jac in (1, 0):
    lib.ebo_debug_edge_clocks(buf, 1)
    reps = 5
    torch.cuda.synchronize(); e0.record(stream)
    for _ in range(reps):
        ctx.eval_device(d_flows.data_ptr(), jac, d_out.data_ptr())
    e1.record(stream); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    lib.ebo_debug_edge_clocks(buf, 0)
    clk = np.array(list(buf), dtype=np.float64)[:10] / reps / n_active
    tot = clk.sum()
    print("cfg %d win %d jac %d: %.3f ms per launch, %d active units, %.0f clocks per unit" % (config, windows, jac, ms, n_active, tot))
    cnt = np.array(list(buf), dtype=np.float64)[10:14] / reps / n_active
    print("   per unit: box %.0f px, eigenvalue region %.0f px, NMS windows %.0f, argmax entries %.0f" % tuple(cnt))
    for n, c in zip(names, clk):
        print("   %-13s %9.0f clk  %5.1f %%  ~%.3f ms of the launch" % (n, c, 100 * c / tot, ms * c / tot))
