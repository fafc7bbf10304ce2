"""Config 5's exchange (BASELINE.json configs[4]: independent sequences, one per GPU, gather of
per-sequence tracks) and the trajectory.txt format it carries
(tools/evaluator/src/evaluator.cpp:125-150; reference test tools/evaluator/test/evaluator_test.cpp:19-83)."""
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def test_trajectory_txt_round_trip_as_the_reference_test(tmp_path, ebo):
    """evaluator_test.cpp:19-83 saveTrajectoryTest: 2 patches, the initial corner (0,0)@0 plus
    30 setCorner samples (j, j)@j us each; the file parses back to the same (id, ts, x, y) list
    within EXPECT_FLOAT_EQ (4 float ulps)."""
    recs = []
    for i in range(2):
        recs.append((i, 0, 0.0, 0.0))
        for j in range(30):
            recs.append((i, j, float(j), float(j)))
    pts = np.array(recs, dtype=ebo.TRACK_DTYPE)
    path = tmp_path / "trajectory.txt"
    ebo.write_tracks_txt(path, pts)
    lines = open(path).read().splitlines()
    assert len(lines) == len(pts) == 62
    # std::fixed << std::setprecision(8): id as an integer, three doubles with 8 decimals
    assert lines[0] == "0 0.00000000 0.00000000 0.00000000"
    assert lines[2] == "0 0.00000100 1.00000000 1.00000000"
    assert lines[61] == "1 0.00002900 29.00000000 29.00000000"
    parsed = [tuple(float(v) for v in ln.split()) for ln in lines]
    for (pid, ts, x, y), want in zip(parsed, pts):
        assert pid == want["id"]
        for got, exp in ((ts, want["t_us"] / 1e6), (x, want["x"]), (y, want["y"])):
            assert np.float32(got) == pytest.approx(np.float32(exp), rel=4 * 2.0 ** -23, abs=0)
    back = ebo.read_tracks_txt(path)
    assert np.array_equal(back["id"], pts["id"]) and np.array_equal(back["t_us"], pts["t_us"])
    assert np.array_equal(back["x"], pts["x"]) and np.array_equal(back["y"], pts["y"])


def test_trajectory_txt_real_timestamps_and_errors(tmp_path, ebo):
    pts = np.array([(7, 1468939993086614, 120.123456789, 64.5), (-1, 1, -0.5, 1e-9)], dtype=ebo.TRACK_DTYPE)
    path = tmp_path / "t.txt"
    ebo.write_tracks_txt(path, pts)
    lines = open(path).read().splitlines()
    assert lines[0] == "7 1468939993.08661389 120.12345679 64.50000000"  # duration<double>: one division
    assert lines[1] == "-1 0.00000100 -0.50000000 0.00000000"
    back = ebo.read_tracks_txt(path)
    assert back["t_us"].tolist() == [1468939993086614, 1]
    bad = tmp_path / "bad.txt"
    bad.write_text("3 0.5 1.0 2.0\n4 0.5 nonsense\n")
    with pytest.raises(ebo.EboError) as ei:
        ebo.read_tracks_txt(bad)
    assert ei.value.code == ebo.ERR_RANGE
    with pytest.raises(ebo.EboError):
        ebo.write_tracks_txt(tmp_path / "no" / "such" / "dir.txt", pts)
    # a file with more records than the buffer: never a silently truncated list -- the call says how many
    # it holds (and keeps the first cap), the wrapper grows its buffer and reads them all
    import ctypes as C
    many = np.zeros(10, dtype=ebo.TRACK_DTYPE)
    many["id"] = np.arange(10)
    many["t_us"] = 1000 * np.arange(10)
    longer = tmp_path / "long.txt"
    ebo.write_tracks_txt(longer, many)
    out = np.zeros(4, dtype=ebo.TRACK_DTYPE)
    n = C.c_size_t()
    rc = ebo.lib().ebo_read_tracks_txt(str(longer).encode(), out.ctypes.data_as(C.c_void_p), C.c_size_t(4), C.byref(n))
    assert rc == ebo.ERR_ARG and n.value == 10 and out["id"].tolist() == [0, 1, 2, 3]
    assert np.array_equal(ebo.read_tracks_txt(longer, cap=4), many)


@pytest.mark.parametrize("sizes", [[5, 12], [9, 0, 4]])
def test_track_gather_over_gloo_equals_the_concatenated_lists(tmp_path, sizes):
    """World 2 and 3, unequal (and empty) per-rank lists: every rank ends up with the single-process
    concatenation rank 0 .. rank N-1."""
    world = len(sizes)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(29520 + world),
           os.path.join(HERE, "mp_tracks_worker.py"), str(tmp_path), ",".join(str(s) for s in sizes)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    sys.path.insert(0, HERE)
    worker = importlib.import_module("mp_tracks_worker")
    want = np.concatenate([worker.tracks_of_rank(r, sizes) for r in range(world)])
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "tracks_rank%d.npy" % r))
        assert got.dtype == want.dtype and np.array_equal(got, want)


@pytest.mark.gpu
def test_allgather_tracks_single_rank_rccl(ebo):
    """The library's own RCCL path (ebo_allgather_tracks) at the one rank a single-GPU box allows:
    counts + padded all-gather on the context's stream, padding dropped, order kept; the empty
    list; the too-small output buffer."""
    import ctypes as C
    rng = np.random.default_rng(3)
    pts = np.zeros(37, dtype=ebo.TRACK_DTYPE)
    pts["id"] = rng.integers(0, 9, 37)
    pts["t_us"] = np.sort(rng.integers(0, 10**12, 37))
    pts["x"] = rng.uniform(0, 346, 37)
    pts["y"] = rng.uniform(0, 260, 37)
    with ebo.Context(image_w=346, image_h=260) as c:
        with pytest.raises(ebo.EboError) as ei:
            c.allgather_tracks(pts)
        assert ei.value.code == ebo.ERR_STATE  # no communicator yet
        assert c.comm_size() == (0, 1)
        c.comm_init(ebo.comm_unique_id(), 0, 1)
        assert c.comm_size() == (0, 1)
        got, counts = c.allgather_tracks(pts)
        assert counts.tolist() == [37] and np.array_equal(got, pts)
        got, counts = c.allgather_tracks(pts[:0])
        assert counts.tolist() == [0] and len(got) == 0
        out = np.zeros(5, dtype=ebo.TRACK_DTYPE)
        n = C.c_size_t()
        rc = ebo.lib().ebo_allgather_tracks(c._h, pts.ctypes.data_as(C.c_void_p), C.c_size_t(37),
                                            out.ctypes.data_as(C.c_void_p), C.c_size_t(5), C.byref(n), None)
        assert rc == ebo.ERR_ARG and n.value == 37 and not out["id"].any()
        # the counts collective alone (what sizes the buffer), and the reduce of SURVEY 8(e)'s final image
        cnt = np.zeros(1, dtype=np.uint64)
        assert ebo.lib().ebo_allgather_track_counts(c._h, C.c_size_t(37), C.byref(n), cnt.ctypes.data_as(C.c_void_p)) == 0
        assert n.value == 37 and cnt.tolist() == [37]
        import torch
        a = torch.arange(1000, dtype=torch.float64, device="cuda")
        b = torch.zeros_like(a)
        c.set_stream(torch.cuda.current_stream().cuda_stream)
        c.reduce_sum_device(a.data_ptr(), b.data_ptr(), 1000, root=0)
        c.synchronize()
        assert torch.equal(a, b)
        b.zero_()
        c.reduce_sum_device(a.data_ptr(), b.data_ptr(), 1000, root=-1)  # all-reduce
        c.synchronize()
        assert torch.equal(a, b)
        c.comm_destroy()
