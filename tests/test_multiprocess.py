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


@pytest.mark.parametrize("world,halo,amp,escapes", [(2, 20, 0.7, False), (3, 24, 0.9, False), (3, 10, 3.0, True)])
def test_band_limited_image_over_gloo(tmp_path, orc, world, halo, amp, escapes):
    """The N > 1 layout of the band-limited final image at world size 2 and 3 over gloo: band plan per
    rank, halos to the two neighbours (exchange.halo_exchange), own rows + received halos, owned rows assembled on
    rank 0 = the oracle's image of the whole window, bit for bit; flows beyond the halo raise the flag on every rank."""
    out = str(tmp_path / "band.npy")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(29520 + world),
           os.path.join(HERE, "mp_band_worker.py"), out, str(halo), str(amp)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    full, whole, flag = np.load(out), np.load(out + ".whole.npy"), np.load(out + ".flag.npy")
    assert int(flag[0]) == (1 if escapes else 0)
    if not escapes:
        assert np.array_equal(full, whole) and whole.sum() > 6000
    else:
        assert full.sum() < whole.sum()  # events were lost beyond the halo: exactly what the flag reports


def _handover_rank(prefix, rank, world, q, not_before):
    import time
    ex = importlib.import_module("event-based-odomety_amd.exchange")
    payload = bytes(range(128)) if rank == 0 else None
    if rank == 0:
        time.sleep(0.5)  # the others meet the stale file first
    got = ex.handover_bytes(prefix, rank, world, payload, timeout=30, not_before=not_before)
    all_ok = ex.agree(prefix, rank, world, rank != 1, not_before=not_before)  # rank 1 says no
    q.put((rank, got, all_ok))


def test_comm_id_handover_without_a_framework(tmp_path):
    """bench.py --comm ebo: rank 0's 128-byte communicator id reaches every rank through a file (no sockets, no
    process group), and a yes / no vote is seen the same way by every rank -- three processes, CPU only."""
    import multiprocessing as mp
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    import time
    prefix = str(tmp_path / "comm")
    # what a crashed earlier run left behind under the same name: older than this run, so nobody may take it
    for stale in (prefix + ".id", prefix + ".ok.0", prefix + ".ok.2"):
        with open(stale, "wb") as fp:
            fp.write(b"1" if ".ok." in stale else b"stale" * 20)
        os.utime(stale, (time.time() - 3600, time.time() - 3600))
    procs = [mpc.Process(target=_handover_rank, args=(prefix, r, 3, q, time.time() - 1.0)) for r in range(3)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=60) for _ in range(3))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert [r[0] for r in res] == [0, 1, 2]
    assert all(r[1] == bytes(range(128)) for r in res)
    assert all(r[2] is False for r in res)
    assert not os.path.exists(prefix + ".id")  # rank 0 cleaned up after every rank had read it
