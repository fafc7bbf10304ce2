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
    for stale in (prefix + ".id", prefix + ".ok.0.0", prefix + ".ok.0.2", prefix + ".got.1"):
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


def test_band_plan_send_and_receive_counts_pair_up_on_every_rank(ebo):
    """Host only (advisor, round 4): ebo_band_exchange_device posts one grouped ncclSend / ncclRecv per neighbour with
    counts taken from EACH rank's own ebo_band_plan -- a rank that sends a count its neighbour does not receive is a
    deadlock no fallback can rescue.  Over random row bounds and halos, including empty leading, trailing and middle
    ranks: the plan is either refused on EVERY rank (the dense path runs) or, on every rank, what rank r sends up is what
    rank r - 1 receives from below and what it sends down is what rank r + 1 receives from above, first / last rank
    send nothing outwards, and a rank without rows sends and receives nothing."""
    rng = np.random.default_rng(20)
    plans = refused = 0
    for trial in range(4000):
        world = int(rng.integers(1, 9))
        image_h = int(rng.integers(8, 720))
        cuts = np.sort(rng.integers(0, image_h + 1, world - 1)) if world > 1 else np.array([], dtype=int)
        if trial % 5 == 0 and world > 2:
            cuts[: int(rng.integers(1, world - 1))] = 0          # empty leading ranks
        if trial % 7 == 0 and world > 2:
            cuts[-int(rng.integers(1, world - 1)):] = image_h     # empty trailing ranks
        bounds = np.concatenate([[0], np.sort(cuts), [image_h]]).astype(np.int32)
        halo = int(rng.choice([0, 1, 7, 32, 64, 500]))
        got = []
        for r in range(world):
            try:
                got.append(ebo.band_plan(image_h, bounds, r, halo))
            except ebo.EboError as exc:
                got.append(exc.code)
        if any(isinstance(g, int) for g in got):
            assert all(isinstance(g, int) and g == got[0] for g in got), (bounds, halo, got)  # the same verdict everywhere
            refused += 1
            continue
        plans += 1
        for r, b in enumerate(got):
            up, down = b.own_row0 - b.band_row0, b.band_row1 - b.own_row1
            rows = b.own_row1 - b.own_row0
            assert (b.own_row0, b.own_row1) == (bounds[r], bounds[r + 1])
            if rows == 0:
                assert up == down == b.recv_above == b.recv_below == 0
            sends_up = up if r > 0 else 0
            sends_down = down if r < world - 1 else 0
            if r == 0:
                assert up == 0 and b.recv_above == 0
            if r == world - 1:
                assert down == 0 and b.recv_below == 0
            if r > 0:
                assert sends_up == got[r - 1].recv_below, (bounds, halo, r)
            if r < world - 1:
                assert sends_down == got[r + 1].recv_above, (bounds, halo, r)
    assert plans > 500 and refused > 100


def _silent_rank(prefix, rank, world, q, not_before, mode):
    """rank 1 never votes (mode 'vote') / rank 0 has no id to hand over (mode 'id')"""
    import time
    ex = importlib.import_module("event-based-odomety_amd.exchange")
    t0 = time.time()
    try:
        if mode == "id":
            ex.handover_bytes(prefix, rank, world, None, timeout=60, not_before=not_before)  # rank 0 publishes "no id"
            q.put((rank, "handed over", time.time() - t0))
            return
        if rank == 1:
            time.sleep(8)  # well beyond the others' timeout; never votes
            q.put((rank, "silent", time.time() - t0))
            return
        ex.agree(prefix, rank, world, True, timeout=3.0, not_before=not_before)
        q.put((rank, "agreed", time.time() - t0))
    except Exception as exc:
        q.put((rank, "%s: %s" % (type(exc).__name__, exc), time.time() - t0))


@pytest.mark.parametrize("mode", ["vote", "id"])
def test_first_contact_failures_end_fast_with_a_reason(tmp_path, mode):
    """Round 5 (fail fast on first multi-GPU contact).  (i) A rank that never writes its vote: every other rank gets a
    TimeoutError that names it after the vote's timeout (3 s here, 120 s in bench.py -- never the 900 s of round 4),
    which bench.py counts as a `no`.  (ii) Rank 0 cannot make an id: it hands a failure mark over and EVERY rank
    raises at once instead of waiting for a file that never comes.  (iii) bench.py's launcher ends children that
    outlive its deadline -- their whole process group -- and exits non-zero with a one-line diagnosis."""
    import multiprocessing as mp
    import time
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    prefix = str(tmp_path / ("first_contact_" + mode))
    ps = [mpc.Process(target=_silent_rank, args=(prefix, r, 3, q, time.time() - 1.0, mode)) for r in range(3)]
    t0 = time.time()
    for p in ps:
        p.start()
    res = dict()
    for _ in range(3):
        r, what, dt = q.get(timeout=60)
        res[r] = (what, dt)
    for p in ps:
        p.join(30)
    assert time.time() - t0 < 60
    if mode == "vote":
        for r in (0, 2):
            assert res[r][0].startswith("TimeoutError: rank 1 never voted") and res[r][1] < 6.0, res
        assert res[1][0] == "silent"
    else:
        for r in range(3):
            assert res[r][0].startswith("RuntimeError: rank 0 could not make a communicator id") and res[r][1] < 10.0, res


def test_launcher_ends_children_that_outlive_its_deadline(tmp_path):
    import subprocess
    import time
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    child = tmp_path / "child.py"
    child.write_text("import subprocess, sys, time\n"
                     "subprocess.Popen([sys.executable, '-c', 'import time; time.sleep(600)'])  # a grandchild in the same group\n"
                     "time.sleep(600)\n")
    driver = tmp_path / "driver.py"
    driver.write_text("import sys\n"
                      "sys.path.insert(0, %r)\n"
                      "import bench\n"
                      "sys.exit(bench.run_children([sys.executable, %r], None, 2.0))\n" % (ROOT, str(child)))
    t0 = time.time()
    out = subprocess.run([sys.executable, str(driver)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 124 and time.time() - t0 < 60, (out.returncode, out.stderr)
    assert "did not finish within 2 s" in out.stderr and out.stderr.count("\n") == 1, out.stderr
    ok = tmp_path / "ok.py"
    ok.write_text("import sys\n"
                  "sys.path.insert(0, %r)\n"
                  "import bench\n"
                  "sys.exit(bench.run_children([sys.executable, '-c', 'import sys; sys.exit(7)'], None, 30.0))\n" % ROOT)
    assert subprocess.run([sys.executable, str(ok)], capture_output=True, timeout=120).returncode == 7


def test_launcher_takes_its_children_along_when_it_is_ended(tmp_path):
    """The ranks run in their own session (so that their whole group can be ended at the deadline): a SIGTERM to the
    launcher must therefore be passed on, or they would outlive it and keep their GPUs."""
    import signal
    import subprocess
    import time
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    mark = tmp_path / "child.pid"
    child = tmp_path / "child.py"
    child.write_text("import os, time\nopen(%r, 'w').write(str(os.getpid()))\ntime.sleep(600)\n" % str(mark))
    driver = tmp_path / "driver.py"
    driver.write_text("import sys\nsys.path.insert(0, %r)\nimport bench\n"
                      "sys.exit(bench.run_children([sys.executable, %r], None, 300.0))\n" % (ROOT, str(child)))
    p = subprocess.Popen([sys.executable, str(driver)])
    t0 = time.time()
    while not (mark.exists() and mark.read_text().strip()) and time.time() - t0 < 30:
        time.sleep(0.05)
    pid = int(mark.read_text())
    p.send_signal(signal.SIGTERM)
    assert p.wait(timeout=60) == 128 + signal.SIGTERM
    for _ in range(100):
        try:
            os.kill(pid, 0)
            time.sleep(0.05)
        except ProcessLookupError:
            break
    else:
        os.kill(pid, signal.SIGKILL)
        raise AssertionError("the child outlived the launcher")
