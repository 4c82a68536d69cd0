"""bench.py's own launcher (CPU): `python bench.py --gpus N` without a launcher's environment must start N ranks itself
and pass on their failure -- round 1 silently ran one rank and printed n_gpus: 1."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def test_gpus_n_without_a_launcher_starts_n_ranks_and_propagates_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the ranks would run the benchmark")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--patches", "64"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0                                       # no GPU here: both ranks refuse, the parent says so
    # two ranks were started (torchrun may stop the second one before it gets to say so, once the first has failed)
    assert 1 <= r.stderr.count("bench.py needs an MI355X") <= 2 and "ChildFailedError" in r.stderr
    assert '"n_gpus"' not in r.stdout                               # and no one-rank line was printed in their place


def test_world_size_must_match_gpus():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "--gpus 4 but WORLD_SIZE=2" in r.stderr


def test_the_launcher_kills_ranks_that_do_not_finish_in_time():
    """A rank stuck in a collective must not hang the parent for ever: past LF_BENCH_LAUNCH_LIMIT_S the children's process
    group is killed and the parent exits non-zero (124), without launching anything again."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["LF_BENCH_LAUNCH_LIMIT_S"] = "0.5"      # the ranks need longer than that just to import torch
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--patches", "64"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 124 and "did not finish within" in r.stderr
    assert time.time() - t0 < 60 and '"n_gpus"' not in r.stdout
