"""kai_parallel.hpp — the parallel primitive of kai_session_open's host preparation — under stress (tests/host_sim/pool_stress.cpp): results of chunked sums on the worker pool, with
concurrent callers, nested loops, an exception carried out of a chunk, and fork()ed children, with the pool and with KAI_HOST_POOL=0."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def stress_binary(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("pool") / "pool_stress")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-o", out, os.path.join(ROOT, "tests", "host_sim", "pool_stress.cpp")])
    return out


@pytest.mark.parametrize("env", [{}, {"KAI_HOST_POOL": "0"}, {"KAI_HOST_THREADS": "3"}, {"KAI_HOST_THREADS": "16"}], ids=["pool", "no pool", "3 threads", "16 threads"])
def test_parallel_chunks_under_stress(stress_binary, env):
    e = dict(os.environ); e.pop("KAI_HOST_THREADS", None); e.pop("KAI_HOST_POOL", None); e.update(env)
    r = subprocess.run([stress_binary], env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "pool stress ok" in r.stdout, (r.stdout, r.stderr)
