"""CPU rehearsal of the N > 1 path: world size 2 (and 3, uneven shards) over gloo."""
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("world,n_windows", [(2, 4), (3, 5)])
def test_sharded_eval_allgather_matches_single_process(tmp_path, orc, ebo, synth, world, n_windows):
    out = str(tmp_path / "gathered.npy")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(29500 + world),
           os.path.join(HERE, "mp_worker.py"), out, str(n_windows)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    full = np.load(out)
    assert full.shape == (n_windows, 108, 3)
    prm = orc.default_params(loss=1, tv_weight=0.0)
    for w in range(n_windows):
        ev, gt = synth.make_window(0, window=w, n_events=4000)
        r, J, _, _ = orc.window_eval(ev, prm, gt * 0.5)
        assert np.array_equal(full[w, :, 0], r)
        assert np.array_equal(full[w, :, 1:], J)
    flows = np.load(out + ".flows.npy")
    per = (n_windows + world - 1) // world
    for q in range(world):
        b, e = ebo.shard_range(n_windows, q, world)
        for k, w in enumerate(range(b, e)):
            assert np.all(flows[q * per + k] == w + 0.5)
