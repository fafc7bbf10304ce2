"""CPU: the product's host concurrency under the sanitizers (SURVEY section 5 puts sanitizers on the CPU build).

csrc/host_pool.h (the spin-then-park thread pool that lives for one solve call) and csrc/lockstep.h (the
lock-step drivers of EBO_SOLVE_GLOBAL / the lock-step EBO_SOLVE_INDEPENDENT, incl. the pipelined form with
2-4 groups of windows in flight) are the code libebo_hip.so ships; they are free of HIP, and
tests/cpp/hostlm_stress.cpp instantiates them with a CPU objective evaluated on another thread (the device is
asynchronous to the host in the same way).  Built with -fsanitize=thread and with
-fsanitize=address,undefined and run with 1..16 host threads, with and without the spin phase: no report,
and the pipelined solves equal the plain one bit for bit."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CPP = os.path.join(HERE, "cpp")


def _run(binary, threads, spin_us=None, scale="1"):
    env = dict(os.environ, EBO_HOST_THREADS=str(threads))
    if spin_us is not None:
        env["EBO_HOST_SPIN_US"] = str(spin_us)
    env.setdefault("TSAN_OPTIONS", "halt_on_error=1")
    out = subprocess.run([os.path.join(CPP, binary), scale], env=env, capture_output=True, text=True, timeout=600)
    text = out.stdout + out.stderr
    assert out.returncode == 0, text[-3000:]
    assert "all passed" in out.stdout
    for bad in ("ThreadSanitizer", "AddressSanitizer", "runtime error", "LeakSanitizer"):
        assert bad not in text, text[-3000:]


def test_host_stress_plain_build():
    subprocess.check_call(["make", "-s", "-C", CPP, "hostlm_stress"])
    _run("hostlm_stress", 8, scale="2")


@pytest.mark.parametrize("threads,spin_us", [(1, None), (2, None), (5, None), (16, None), (6, 0), (16, 2000)])
def test_host_pool_and_lock_step_drivers_under_thread_sanitizer(threads, spin_us):
    subprocess.check_call(["make", "-s", "-C", CPP, "hostlm_stress_tsan"])
    _run("hostlm_stress_tsan", threads, spin_us)


@pytest.mark.parametrize("threads", [1, 4, 16])
def test_host_pool_and_lock_step_drivers_under_address_and_ub_sanitizers(threads):
    subprocess.check_call(["make", "-s", "-C", CPP, "hostlm_stress_asan"])
    _run("hostlm_stress_asan", threads)


@pytest.mark.parametrize("no_avx2", ["", "1"])
def test_host_lm_trajectories_keep_their_bits(no_avx2):
    """The vectorised banded Cholesky of round 3 (four chains per vector register; AVX2 where the CPU has it,
    SSE2 otherwise) performs the scalar loop's operations in the scalar loop's order: 200 synthetic solves end in
    the flows the row-by-row factorisation of rounds 1-2 produced, bit for bit, with either instruction set."""
    subprocess.check_call(["make", "-s", "-C", CPP, "hostlm_golden"])
    env = dict(os.environ)
    env.pop("EBO_LM_NO_AVX2", None)
    if no_avx2:
        env["EBO_LM_NO_AVX2"] = no_avx2
    out = subprocess.run([os.path.join(CPP, "hostlm_golden")], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "as pinned" in out.stdout


def _run_txt(binary, lines, *more):
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        env = dict(os.environ)
        env.setdefault("TSAN_OPTIONS", "halt_on_error=1")
        out = subprocess.run([os.path.join(CPP, binary), d, str(lines)] + list(more), env=env, capture_output=True, text=True, timeout=900)
    text = out.stdout + out.stderr
    assert out.returncode == 0 and "txt_events_stress: OK" in out.stdout, text[-3000:]
    for bad in ("ThreadSanitizer", "AddressSanitizer", "runtime error"):
        assert bad not in text, text[-3000:]
    return out.stdout


def test_parallel_events_txt_reader_equals_the_single_thread_reader():
    """Round 5: csrc/txt_events.h parses an events.txt on the host's threads, as the reference's
    reader does (dataset_reader.h:33-97).  Against the single-thread reader of rounds 1-4 (kept in the test): the same
    events, count, byte offset and status for 24 mixed files (blank lines, comments, CRLF, exponents, signs, tabs,
    mantissas above 2^53, a malformed line, no trailing newline) x 1..16 threads x caps around the merge's edge cases
    and chains of offsets, and a 4 M-line canonical file bit for bit (10 M lines by hand: profiles/r05_text_ingest.txt) -- also read in the reference's pieces of
    1 000 000 lines; the rates are printed."""
    subprocess.check_call(["make", "-s", "-C", CPP, "txt_events_stress"])
    text = _run_txt("txt_events_stress", 4_000_000)
    print(text[-900:])


def test_parallel_events_txt_reader_under_thread_sanitizer():
    subprocess.check_call(["make", "-s", "-C", CPP, "txt_events_stress_tsan"])
    _run_txt("txt_events_stress_tsan", 100_000, "quick")
