"""The C++ facade (tracker::FeatureDetector / contrastFunctor / ContrastBatch with the
reference's names) built with the host compiler and run on the GPU box."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CPP = os.path.join(HERE, "cpp")


def test_facade_compiles_against_the_abi(ebo, orc):
    """CPU: the facade headers + test program compile and link against libebo_hip.so."""
    ebo.lib()
    subprocess.check_call(["make", "-s", "-C", CPP, "facade_test"])
    assert os.path.exists(os.path.join(CPP, "facade_test"))


@pytest.mark.gpu
def test_facade_against_oracle_on_gpu(ebo, orc):
    ebo.lib()
    subprocess.check_call(["make", "-s", "-C", CPP, "facade_test"])
    out = subprocess.run([os.path.join(CPP, "facade_test")], capture_output=True, text=True, timeout=600)
    print(out.stdout[-3000:], out.stderr[-2000:])
    assert out.returncode == 0, out.stdout[-3000:]
    assert "all passed" in out.stdout


def test_ceres_surface_compiles(ebo):
    """CPU: the EBO_HAVE_CERES branch of feature_tracker/contrast_functor.h (HipContrastCost, a
    ceres::SizedCostFunction<1, 2>; ContrastBatch, a ceres::EvaluationCallback) compiles under
    -Wall -Wextra against the test-only declarations of tests/cpp/stubs/ceres/ceres.h."""
    ebo.lib()
    out = subprocess.run(["make", "-B", "-C", CPP, "ceres_adaptor_test"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "warning" not in out.stderr, out.stderr[-3000:]


@pytest.mark.gpu
def test_ceres_surface_solves_like_ebo_solve(ebo):
    """The reference's problem construction (feature_detector.cpp:301-367) with HipContrastCost blocks
    and ContrastBatch as the evaluation callback, minimised by the product's host LM through
    CostFunction::Evaluate: the same flows as ebo_solve(EBO_SOLVE_GLOBAL), bit for bit, both losses."""
    ebo.lib()
    subprocess.check_call(["make", "-s", "-C", CPP, "ceres_adaptor_test"])
    out = subprocess.run([os.path.join(CPP, "ceres_adaptor_test")], capture_output=True, text=True, timeout=600)
    print(out.stdout[-3000:], out.stderr[-2000:])
    assert out.returncode == 0, out.stdout[-3000:]
    assert "all passed" in out.stdout


def test_reference_construction_lines_compile(ebo):
    """CPU: the reference's own statements -- `new ceres::AutoDiffCostFunction<tracker::contrastFunctor, 1,
    2>(new tracker::contrastFunctor(patchEvents, patchRect, params_.compensateScale))`, the
    AutoDiffCostFunction<tracker::totalVarianceFunctor, 2, 2, 2> + HuberLoss blocks and Solve
    (feature_detector.cpp:357-414) -- compile under -Wall -Wextra against the facade (reference-style
    <feature_tracker/...> includes): the functor's operator() is instantiated with ceres::Jet."""
    ebo.lib()
    out = subprocess.run(["make", "-B", "-C", CPP, "ceres_reference_lines_test"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "warning" not in out.stderr, out.stderr[-3000:]


@pytest.mark.gpu
def test_reference_construction_lines_solve_like_ebo_solve(ebo):
    """The problem built by those unchanged statements, every data term evaluated through
    AutoDiffCostFunction<contrastFunctor,1,2>::Evaluate -> operator()<Jet<double,2>> -> one device launch,
    minimised by the product's host LM behind ceres::Solve: the flows of ebo_solve(EBO_SOLVE_GLOBAL) bit
    for bit, both losses; the Jet path equals Evaluate() and obeys the chain rule for a general seed."""
    ebo.lib()
    subprocess.check_call(["make", "-s", "-C", CPP, "ceres_reference_lines_test"])
    out = subprocess.run([os.path.join(CPP, "ceres_reference_lines_test")], capture_output=True, text=True, timeout=900)
    print(out.stdout[-3000:], out.stderr[-2000:])
    assert out.returncode == 0, out.stdout[-3000:]
    assert "all passed" in out.stdout


def test_reference_replayer_and_trajectory_scenarios(ebo):
    """CPU: tools/replayer/test/replayer_test.cpp:67-125 (nextTest, nextImageTest, resetTest on the
    reference's own two data files) through tools::StreamPump, and
    tools/evaluator/test/evaluator_test.cpp:19-83 (saveTrajectoryTest) through
    tools::saveFeaturesTrajectory -- host entry points only, no GPU."""
    ebo.lib()
    subprocess.check_call(["make", "-s", "-C", CPP, "tools_test"])
    out = subprocess.run([os.path.join(CPP, "tools_test"), os.path.join(HERE, "golden", "replayer")],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "all passed" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_front_end_lines_compile_and_host_subset(ebo):
    """CPU: every member the front end calls on its tracker (evaluator.cpp:15-45,106-109,120-123; keyframe.cpp:5-14) is
    pinned by name, argument and result type (a member-pointer table over feature_detector.h:33-88) and called, in the
    evaluator's order, by the test's own driver under -Wall -Wextra against the facade's ONE tracker::FeatureDetector; the
    scenarios of the reference's updatePatchTest / associatedPatchesTest; the host-only subset (patch bookkeeping, every DetectorParams
    field with the reference's default, EBO_ERR_UNSUPPORTED for newImage without hooks) runs without a device."""
    ebo.lib()
    out = subprocess.run(["make", "-B", "-C", CPP, "front_end_lines_test"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "warning" not in out.stderr, out.stderr[-3000:]
    run = subprocess.run([os.path.join(CPP, "front_end_lines_test"), "--cpu"], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0 and "all passed" in run.stdout, run.stdout[-2000:] + run.stderr[-2000:]


@pytest.mark.gpu
def test_front_end_lines_run_like_the_pair(ebo):
    """The evaluator's event loop through the one FeatureDetector (tracker + compensation on one device
    context): patches, flows and both images bit-equal to the stand-alone TrackedPatches + FeatureDetector
    pair; the newImage life cycle through FrontEndHooks; Keyframe over getPatches()."""
    ebo.lib()
    subprocess.check_call(["make", "-s", "-C", CPP, "front_end_lines_test"])
    out = subprocess.run([os.path.join(CPP, "front_end_lines_test")], capture_output=True, text=True, timeout=900)
    print(out.stdout[-4000:], out.stderr[-2000:])
    assert out.returncode == 0, out.stdout[-4000:]
    assert "all passed" in out.stdout


def test_front_end_lines_with_opencv_types_compile_and_host_subset(ebo):
    """CPU: the facade's EBO_HAVE_OPENCV branch (common::Point2i = cv::Point2i, tracker::Size / Rect2i / Rect2d = the
    cv:: types, Mat64 a cv::Mat of CV_64F, ImageSample = Sample<cv::Mat>) against test-only OpenCV declarations
    (tests/cpp/stubs_opencv): the evaluator shell keeps `cv::Size2i imageSize` and `cv::Mat const&
    getCompensatedEventImage()`, i.e. the reference's statements compile with no type renamed."""
    ebo.lib()
    out = subprocess.run(["make", "-B", "-C", CPP, "front_end_lines_opencv_test"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "warning" not in out.stderr, out.stderr[-3000:]
    run = subprocess.run([os.path.join(CPP, "front_end_lines_opencv_test"), "--cpu"], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0 and "all passed" in run.stdout, run.stdout[-2000:] + run.stderr[-2000:]


@pytest.mark.gpu
def test_front_end_lines_with_opencv_types_run_like_the_pair(ebo):
    ebo.lib()
    subprocess.check_call(["make", "-s", "-C", CPP, "front_end_lines_opencv_test"])
    out = subprocess.run([os.path.join(CPP, "front_end_lines_opencv_test")], capture_output=True, text=True, timeout=900)
    print(out.stdout[-4000:], out.stderr[-2000:])
    assert out.returncode == 0, out.stdout[-4000:]
    assert "all passed" in out.stdout


def test_patch_lines_compile_and_host_subset(ebo):
    """CPU: the scenarios of the reference's patch tests (patch_test.cpp:7-33,35-60,62-91) in the test's own statements,
    the integrateEvents known answer as a data table, and a member-pointer table over every public member of patch.h:15-160,
    under -Wall -Wextra against the facade's tracker::Patch; the host-only subset runs: the event window behaves as the
    reference's patch.cpp implies (three expectations of the reference's own addEventsTest are stale there), the
    bookkeeping getters, and every per-patch device member throws "no device context" when none is bound."""
    ebo.lib()
    out = subprocess.run(["make", "-B", "-C", CPP, "patch_lines_test"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "warning" not in out.stderr, out.stderr[-3000:]
    run = subprocess.run([os.path.join(CPP, "patch_lines_test"), "--cpu"], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0 and ": OK" in run.stdout, run.stdout[-2000:] + run.stderr[-2000:]


@pytest.mark.gpu
def test_patch_lines_run_on_the_device(ebo):
    """integrateEventsTest and warpImageTest of the reference through the per-patch members (one launch each),
    integrateMotionCompensatedEvents against patch.cpp:87-130 written out in the test, warpImage() away from the
    border = the batched ABI call, OptimizerParams::drawCostMap through Optimizer::optimize = ebo_optimizer_cost_map."""
    ebo.lib()
    subprocess.check_call(["make", "-s", "-C", CPP, "patch_lines_test"])
    out = subprocess.run([os.path.join(CPP, "patch_lines_test")], capture_output=True, text=True, timeout=600)
    print(out.stdout[-4000:], out.stderr[-2000:])
    assert out.returncode == 0 and ": OK" in out.stdout, out.stdout[-4000:]


def test_optimizer_cost_lines_compile_and_host_subset(ebo):
    """CPU: the tracker's Ceres cost function built the way optimizer.cpp:9,15-31,72-79,86-97 builds it (the interleaved
    grid, `Grid`, `Interpolator`, `tracker::OptimizerCostFunctor`, `ceres::AutoDiffCostFunction<tracker::OptimizerCostFunctor,
    ceres::DYNAMIC, Sophus::SE2d::num_parameters, 1>`; own statements, constructor signatures pinned by static_asserts)
    compiles under -Wall -Wextra against <feature_tracker/optimizer_cost.h> (test-only Ceres / OpenCV declarations); a
    functor without an interpolator reports failure."""
    ebo.lib()
    out = subprocess.run(["make", "-B", "-C", CPP, "optimizer_cost_lines_test"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "warning" not in out.stderr, out.stderr[-3000:]
    run = subprocess.run([os.path.join(CPP, "optimizer_cost_lines_test"), "--cpu"], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0 and ": OK" in run.stdout, run.stdout[-2000:] + run.stderr[-2000:]


@pytest.mark.gpu
def test_optimizer_cost_lines_evaluate_like_the_abi(ebo):
    """AutoDiffCostFunction<OptimizerCostFunctor, DYNAMIC, 4, 1>::Evaluate through the facade functor (Jet path, one
    launch) = ebo_optimizer_eval on another context, bit for bit: residuals, both Jacobian blocks, one block, none."""
    ebo.lib()
    subprocess.check_call(["make", "-s", "-C", CPP, "optimizer_cost_lines_test"])
    out = subprocess.run([os.path.join(CPP, "optimizer_cost_lines_test")], capture_output=True, text=True, timeout=600)
    print(out.stdout[-4000:], out.stderr[-2000:])
    assert out.returncode == 0 and ": OK" in out.stdout, out.stdout[-4000:]


def test_reference_reader_lines(ebo, tmp_path):
    """CPU: the known answers of tools/dataset_reader/test/davis240c_reader_test.cpp:19-48 (a data table) against the facade's
    tools::Davis240cReader on the reference's own events.txt fixture; a 2.3 M-event recording read in the reference's
    pieces of EVENT_LENGTH lines (ebo_read_events_txt_at) and again from the packed sidecar; the reference's exception text
    for a bad sign; trajectory.txt read back one Patch per line."""
    ebo.lib()
    out = subprocess.run(["make", "-B", "-C", CPP, "reader_lines_test"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "warning" not in out.stderr, out.stderr[-3000:]
    run = subprocess.run([os.path.join(CPP, "reader_lines_test"), os.path.join(HERE, "golden", "davis_events_fixture.txt"), str(tmp_path)],
                         capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and ": OK" in run.stdout, run.stdout[-2000:] + run.stderr[-2000:]
