"""-m gpu: bench.py's multi-rank control flow.  `python bench.py --gpus 2` must launch two ranks itself (one process per
GPU under torch.distributed.run; on the one-GPU test box both ranks share cuda:0 and the control-plane collectives go over
gloo -- on a multi-GPU node the same code runs over RCCL) and print ONE JSON line with n_gpus = 2, under both scalings."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _bench(*flags):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu", "--prewarm-ms", "20", *flags],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_single_rank_line():
    r = _bench("--pairs", "20000")
    assert r["n_gpus"] == 1 and r["scaling"] == "weak" and r["strong_scaling"] is None
    assert r["config"]["units_total"] == 20000 and r["value"] > 0
    assert r["roofline"]["bound"] == "hbm" and 0 < r["roofline"]["frac"] < 1
    assert r["kernel"]["kernel"] == 3


def test_two_ranks_weak_and_strong():
    w = _bench("--gpus", "2", "--pairs", "20000")
    assert w["n_gpus"] == 2 and w["scaling"] == "weak"
    assert w["config"]["units_per_gpu"] == 20000 and w["config"]["units_total"] == 40000
    assert w["strong_scaling"]["units_total"] == 20000 and w["strong_scaling"]["value"] > 0      # the same batch partitioned
    s = _bench("--gpus", "2", "--pairs", "20000", "--scaling", "strong")
    assert s["n_gpus"] == 2 and s["scaling"] == "strong"
    assert s["config"]["units_per_gpu"] == 10000 and s["config"]["units_total"] == 20000


def test_two_ranks_search_is_one_haystack():
    r = _bench("--gpus", "2", "--workload", "cfg5", "--pairs", "8")
    assert r["n_gpus"] == 2 and r["config"]["units_total"] == 2 * (8 << 20)
    assert "all-gathered" in r["config"]["parallelism"]
