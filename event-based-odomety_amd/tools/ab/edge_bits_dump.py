"""(r, J0, J1) of the edge loss over configurations, batch sizes and flow scales -> one .npz per library build
(EBO_LIB_PATH); tools/ab/edge_bits_cmp.py compares two of them bit for bit.  usage: edge_bits_dump.py out.npz"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
ebo = importlib.import_module("event-based-odomety_amd")
synth = importlib.import_module("event-based-odomety_amd.synth")
out = {}
for config, windows in ((0, 64), (0, 3), (3, 24), (2, 8), (4, 2), (1, 2)):
    cfg = synth.CONFIGS[config]
    ev, offsets, gt = synth.make_stream(config, windows)
    ctx = ebo.Context(image_w=cfg["image"][0], image_h=cfg["image"][1], patch_w=cfg["patch"][0], patch_h=cfg["patch"][1],
                      loss=ebo.LOSS_EDGE, tv_weight=0.0, max_events=len(ev), max_windows=windows)
    ctx.set_windows(ev, offsets)
    rng = np.random.default_rng(100 + config)
    for scale in (0.0, 0.5, 1.0, 2.5):
        f = gt * scale + (rng.normal(0, 0.2, gt.shape) if scale else 0.0)
        d_flows = torch.from_numpy(np.ascontiguousarray(f)).to("cuda")
        d_out = torch.zeros((windows * ctx.P, 3), dtype=torch.float64, device="cuda")
        for jac in (1, 0):
            d_out.zero_()
            ctx.eval_device(d_flows.data_ptr(), jac, d_out.data_ptr())
            torch.cuda.synchronize()
            out["c%d_w%d_s%.1f_j%d" % (config, windows, scale, jac)] = d_out.cpu().numpy().copy()
    ctx.close()
np.savez(sys.argv[1], **out)
print("wrote", sys.argv[1], len(out), "cases")
