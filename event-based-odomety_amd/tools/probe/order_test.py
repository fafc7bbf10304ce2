import importlib, sys, ctypes
sys.path.insert(0, "/root/repo")
ebo = importlib.import_module("event-based-odomety_amd")
which = sys.argv[1]
if which == "lib_first":
    c = ebo.Context()
    import torch
    print("torch after lib:", torch.zeros(1, device="cuda"))
elif which == "torch_first":
    import torch
    print(torch.zeros(1, device="cuda"))
    c = ebo.Context()
    print("lib after torch ok")
elif which == "lib_then_hip":
    c = ebo.Context()
    hip = ctypes.CDLL("libamdhip64.so")
    p = ctypes.c_void_p()
    print("hipMalloc rc", hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(1024)))
elif which == "shard_then_torch":
    import numpy as np
    synth = importlib.import_module("event-based-odomety_amd.synth")
    ev, gt = synth.make_window(0, n_events=3000)
    c = ebo.Context(max_events=len(ev), max_windows=2)
    c.set_patches(ev, [0, 1000, 3000], [(0, 0, 20, 20), (20, 0, 20, 20)])
    try:
        c.count_image_shard(1, [int(ev["t_us"][0]) + (1 << 32)], np.zeros((1, c.P, 2)))
    except Exception as e:
        print("expected", e)
    import torch
    print("torch after shard error:", torch.zeros(1, device="cuda"))
