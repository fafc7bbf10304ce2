"""The N > 1 path through the HIP kernels with more than one process: ranks share the box's one
GPU (at most 3 processes touch it), collectives over gloo (RCCL needs one device per rank; its
1-rank path is tests/test_gpu_comm.py).  What an 8-GPU node runs differs in the transport only."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _torchrun(world, port, script, *argv, env=None, timeout=900):
    e = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    e.update(env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port), script] + [str(a) for a in argv]
    return subprocess.run(cmd, env=e, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize("config,world,n_windows", [(2, 2, 3), (0, 3, 2)])
def test_sharded_device_solve_gather_and_final_image_equal_one_process(tmp_path, ebo, synth, config, world, n_windows):
    iters = 8
    res = _torchrun(world, 29540 + world, os.path.join(HERE, "mp_gpu_worker.py"), tmp_path, config, n_windows, iters)
    assert res.returncode == 0, res.stderr[-3000:]
    cfg = synth.CONFIGS[config]
    evs, offs = [], [0]
    for w in range(n_windows):
        ev, _ = synth.make_window(config, window=w, n_events=min(cfg["events"], 40000))
        evs.append(ev)
        offs.append(offs[-1] + len(ev))
    ev = np.concatenate(evs)
    with ebo.Context(image_w=cfg["image"][0], image_h=cfg["image"][1], patch_w=cfg["patch"][0],
                     patch_h=cfg["patch"][1], loss=ebo.LOSS_VARIANCE, tv_weight=0.0, max_events=len(ev),
                     max_windows=n_windows) as c:
        c.set_windows(ev, offs)
        whole, _ = c.solve(mode=ebo.SOLVE_INDEPENDENT, max_num_iterations=iters)
        act = np.array([[c.patch_info(p, w)[1] for p in range(c.P)] for w in range(n_windows)])
        first = np.load(os.path.join(str(tmp_path), "flows_rank0.npy"))
        # config 4 END TO END: the reduced partial images of the ranks = the one-process final image
        # (feature_detector.cpp:433-463) at the same flows, bit for bit
        image = np.load(os.path.join(str(tmp_path), "image_rank0.npy"))
        assert np.array_equal(image, c.count_image(ebo.COUNT_WARPED, first))
        assert image.sum() > 0.9 * len(ev)
    for r in range(1, world):
        assert np.array_equal(np.load(os.path.join(str(tmp_path), "flows_rank%d.npy" % r)), first)  # every rank: all flows
    assert first.shape == whole.shape
    np.testing.assert_allclose(first[act], whole[act], rtol=0, atol=1e-9)
    assert np.all(first[~act] == 0)


@pytest.mark.parametrize("world,loss", [(2, 0), (3, 0), (3, 1)])
def test_reference_faithful_tv_mode_across_ranks_equals_ebo_solve(tmp_path, ebo, synth, world, loss):
    """SURVEY 8(e) "Collective", reference-faithful mode: the reference's ONE problem per window (data terms +
    TV terms between neighbouring patches, feature_detector.cpp:357-414) with the data terms evaluated where the
    patches live -- every rank its grid rows, on the device -- ONE all-gather of (r, J0, J1) per LM evaluation,
    and the resumable host solver of the ABI (ebo_lm_*) replicated on every rank.  The flows of every rank equal
    the one-process ebo_solve(EBO_SOLVE_GLOBAL) of the same window BIT FOR BIT (reference defaults: 12 x 9 grid of
    equal 20 x 20 patches; edge loss = the reference's own, variance loss = the north-star objective)."""
    res = _torchrun(world, 29560 + world + 10 * loss, os.path.join(HERE, "mp_gpu_tv_worker.py"), tmp_path, 0, loss)
    assert res.returncode == 0, res.stderr[-3000:]
    cfg = synth.CONFIGS[0]
    ev, _ = synth.make_window(0, n_events=min(cfg["events"], 30000))
    with ebo.Context(image_w=cfg["image"][0], image_h=cfg["image"][1], patch_w=cfg["patch"][0], patch_h=cfg["patch"][1],
                     loss=loss, max_events=len(ev)) as c:
        c.set_window(ev)
        whole, summ = c.solve(ebo.default_solver())
    for r in range(world):
        flows = np.load(os.path.join(str(tmp_path), "tvflows_rank%d.npy" % r))
        its, term, evals = np.load(os.path.join(str(tmp_path), "tvstats_rank%d.npy" % r)).tolist()
        assert np.array_equal(flows, whole[0]), np.abs(flows - whole[0]).max()
        assert its == summ[0].iterations and term == summ[0].termination and evals >= its
    # a real solve: several LM iterations (the TV terms hold the reference-default flows near zero -- ~1e-4 px/ms
    # with the edge loss, ~1e-17 with the variance loss, whose lowest-cost visited point is the start: that IS
    # the reference's problem; the bit-equality above is over every evaluation the solvers asked for)
    assert summ[0].iterations >= 3


@pytest.mark.parametrize("extra", [[], ["--replicas", "--windows", "4"]])
def test_bench_runs_at_two_ranks_without_a_launcher(extra):
    """`python bench.py --gpus 2` as the driver calls it (no torchrun around it): it spawns its own
    ranks before touching the GPU and prints ONE JSON line naming the N > 1 workload."""
    env = dict(os.environ, EBO_BENCH_REHEARSE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--c4-windows", "1"] + extra
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 0 and "rehearsal" in line
    if extra:
        assert "configs[4]" in line["config"]["workload"]
        assert line["extras"]["track_gather"]["per_rank"] == [100, 97]
    else:
        assert "patch rows sharded over 2" in line["config"]["workload"]
        assert line["config"]["grid_rows_per_gpu"] == [16, 16]
        assert line["extras"]["allgather_check"] is True


@pytest.mark.parametrize("extra", [["--workload", "c4", "--c4-windows", "2"], ["--replicas", "--windows", "4"]])
def test_bench_on_the_librarys_own_communicator(extra):
    """`--comm ebo` at the one rank this box allows (EBO_BENCH_FORCE_DIST=1 takes the N > 1 code path): the id hand-over,
    ebo_comm_init, barrier / reductions / all-gather of flows through ebo_allgather_device, the band image's
    ebo_band_exchange_device + ebo_band_gather_device, the tracks through ebo_allgather_tracks -- no process group."""
    env = dict(os.environ, EBO_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--comm", "ebo",
           "--no-extras", "--cpu-seconds", "0.5"] + extra
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][0])
    assert "ebo_comm_init" in line["config"]["transport"] and "fallback" not in line["config"]
    if "--replicas" in extra:
        assert line["extras"]["track_gather"]["per_rank"] == [100]
    else:
        fi = line["extras"]["final_image"]
        assert line["config"]["final_image"]["mode"] == "band" and fi["mode"] == "band"
        assert fi["assembled_on_rank0"] and fi["integer_valued"] and fi["sum_equals_own_rows_sum"]
        assert fi["escaped_flag_max_over_ranks"] == 0.0 and 0.9 * fi["events_in_batch"] < fi["events_counted"] <= fi["events_in_batch"]
        assert line["extras"]["allgather_check"] is True


def test_band_and_dense_final_images_count_the_same_events_at_two_ranks():
    """Two ranks on this box's one GPU (rehearsal over gloo: not a measurement): the c4 step with the band-limited
    image (halo rows exchanged between the ranks) counts exactly the events the dense image + reduce counts."""
    env = dict(os.environ, EBO_BENCH_REHEARSE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    counted = {}
    for mode in ("band", "dense"):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--preheat", "0",
               "--c4-windows", "1", "--c4-image", mode, "--cpu-seconds", "0.2"]
        res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        assert res.returncode == 0, res.stderr[-3000:]
        line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][0])
        assert line["config"]["final_image"]["mode"] == mode
        counted[mode] = line["extras"]["final_image"]["events_counted"]
        if mode == "band":
            assert line["config"]["final_image"]["bytes_sent_per_rank_per_window"] == 32 * 1280 * 4
            assert line["extras"]["final_image"]["escaped_flag_max_over_ranks"] == 0.0
    assert counted["band"] == counted["dense"] and counted["band"] > 1.5e6
