import importlib, sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch, numpy as np
ebo = importlib.import_module("event-based-odomety_amd")
import test_gpu_shard as T
import pytest
# run the test body but catch the assembled / whole arrays

seed = 15
orig = np.array_equal
def spy(a, b):
    r = orig(a, b)
    if not r and getattr(a, "shape", None) == getattr(b, "shape", None) and a.ndim == 3:
        d = np.argwhere(a != b)
        print("differences:", len(d), "first", d[:10].tolist())
        print("assembled", [a[tuple(i)] for i in d[:10]], "whole", [b[tuple(i)] for i in d[:10]])
        print("rows with differences", sorted(set(d[:, 1].tolist()))[:40], "cols", sorted(set(d[:, 2].tolist()))[:40])
    return r
np.array_equal = spy
for seed in range(16):
    try:
        T.test_band_limited_images_on_random_geometries(ebo, seed)
        print("seed", seed, "ok")
    except AssertionError as e:
        print("seed", seed, "assert failed")
