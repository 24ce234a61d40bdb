"""csrc/ocd_cma.c's host threads (ABI 8: ocd_cma_many_args.host_threads) under ThreadSanitizer, on the CPU: a plain-C
client (tests/c_client/cma_pool_client.c) and ocd_cma.c compiled together with -fsanitize=thread drive ocd_cma_run_many
with a host function as the episode launch.  A data race is a ThreadSanitizer report and a non-zero exit; the checksum over
every run's history and final state must not depend on the number of threads or launch groups.
(GPU AddressSanitizer / TSan runs do not exist on this pool: sanitizers run on the CPU build only.)"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def client(tmp_path_factory):
    exe = tmp_path_factory.mktemp("cma_pool") / "cma_pool_client"
    cmd = ["gcc", "-std=c11", "-O1", "-g", "-fsanitize=thread", "-ffp-contract=off", "-fno-fast-math", "-mfma", "-pthread", "-Wall",
           "-Wextra", "-Werror", os.path.join(ROOT, "tests", "c_client", "cma_pool_client.c"),
           os.path.join(ROOT, "l4dc-mpc-ocd_amd", "csrc", "ocd_cma.c"), "-o", str(exe), "-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "tsan" in r.stderr.lower():
        pytest.skip("no ThreadSanitizer runtime for this gcc")
    assert r.returncode == 0, r.stderr

    def run(*args):
        env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66")
        out = subprocess.run([str(exe), *map(str, args)], capture_output=True, text=True, timeout=300, env=env)
        assert out.returncode == 0 and "ThreadSanitizer" not in out.stderr, out.stderr[-3000:]
        rows = dict(ln.split() for ln in out.stdout.splitlines())
        return int(rows["done"]), rows["checksum"]
    return run


def test_threads_and_groups_give_one_checksum_and_no_race(client):
    base = client(28, 40, 1, 0)
    assert base[0] == 30                                           # maxiter inside the call: every run dropped out
    for threads, groups in ((2, 0), (8, 0), (4, 2), (16, 3), (64, 8)):
        assert client(28, 40, threads, groups) == base, (threads, groups)


def test_workers_that_went_to_sleep_are_woken(client):
    # a 6 ms stand-in stream wait: past the workers' spin time, so every generation's tells start with a wake-up
    assert client(12, 12, 4, 2, 6000) == client(12, 12, 1, 0)


def test_more_threads_than_runs(client):
    assert client(2, 20, 8, 0) == client(2, 20, 1, 0)
    assert client(1, 20, 8, 0)[0] == 15
