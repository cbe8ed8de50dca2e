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
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu", "--prewarm-ms", "20", *(() if "--pmc-live" in flags else ("--no-pmc",)),
                        *[f for f in flags if f != "--pmc-live"]],
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


def test_eight_ranks_on_one_gpu():
    """The configuration's world size through bench.py's own launcher: 8 ranks sharing the one GPU of the test box (gloo for the
    control plane), pair batches under both scalings and the sharded search."""
    w = _bench("--gpus", "8", "--pairs", "8000")
    assert w["n_gpus"] == 8 and w["config"]["units_total"] == 64000 and w["strong_scaling"]["units_total"] == 8000
    r = _bench("--gpus", "8", "--workload", "cfg5", "--pairs", "4")
    assert r["n_gpus"] == 8 and r["config"]["units_total"] == 8 * (4 << 20)
    assert "all-gathered" in r["config"]["parallelism"]


def test_one_rank_under_the_launcher_equals_plain_run():
    """The N = 1 path of the multi-rank code (bench.py under torch.distributed.run with one rank) reports the figure of the
    plain run (+- a few percent: the same kernels on the same data)."""
    env = dict(os.environ)
    for v in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(v, None)
    flags = ["--steps", "20", "--warmup", "3", "--no-cpu", "--no-pmc", "--pairs", "500000", "--prewarm-ms", "300"]
    plain = _bench(*flags)                            # (argparse keeps the last --steps / --warmup)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", "29631", os.path.join(ROOT, "bench.py"), "--gpus", "1", *flags],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert line["n_gpus"] == 1 and line["config"]["units_total"] == plain["config"]["units_total"]
    assert abs(line["roofline"]["device_ms_per_pass"] / plain["roofline"]["device_ms_per_pass"] - 1.0) < 0.05


def test_line_carries_the_round3_fields():
    r = _bench("--pairs", "50000", "--dist", "ragged")
    assert r["end_to_end_ms"] > r["ms_per_step"] and r["prewarm_passes"] >= 1
    assert r["roofline"]["kernel_name"].startswith("lev_bits") and "traffic_source" in r["roofline"]
    assert 2000 < r["config"]["credited_cells_per_unit"] < 15584


def test_traffic_measured_in_the_run():
    """Without --no-pmc (the driver's command) and with rocprofv3 on the box, roofline.traffic comes from counter passes of THIS run
    (scripts/pmc_collect.py around the same workload), not from a committed file; the timed region is one hipGraph."""
    import shutil
    if not shutil.which("rocprofv3"):
        pytest.skip("no rocprofv3 on this box")
    r = _bench("--pairs", "200000", "--pmc-live")
    assert r["timed_region"].startswith("one hipGraph")
    assert r["roofline"]["traffic_source"].startswith("measured in this run"), r["roofline"]["traffic_source"]
    assert 0.8 < r["roofline"]["traffic"] / r["roofline"]["algorithmic_bytes_per_pass"] < 1.5


def test_single_process_device_set_lines():
    """`bench.py --gpus N --single-process`: ONE process, the library's device set (device 0 listed N times on this box) shards the batch /
    the haystack -- the same JSON shape, `value` from the resident sharded handle, `end_to_end_ms` through the host-pointer entry."""
    w = _bench("--single-process", "--gpus", "3", "--pairs", "20000")
    assert w["n_gpus"] == 3 and w["scaling"] == "weak" and w["single_process"] and w["config"]["units_total"] == 60000
    assert w["config"]["shards"] == 3 and w["value"] > 0 and w["end_to_end_ms"] > 0 and w["parity_checked_some"] > 1000
    s = _bench("--single-process", "--gpus", "8", "--pairs", "40000", "--scaling", "strong")
    assert s["n_gpus"] == 8 and s["config"]["units_total"] == 40000 and s["config"]["devices"] == [0] * 8
    r = _bench("--single-process", "--gpus", "2", "--workload", "cfg5", "--pairs", "8")
    assert r["n_gpus"] == 2 and r["config"]["units_total"] == 2 * (8 << 20) and r["parity_checked_some"] > 0 and r["end_to_end_ms"] > 0


def test_default_line_carries_the_overlapped_figure_and_value_is():
    r = _bench("--pairs", "300000")
    assert r["value_is"] == "one GPU" and r["overlapped_passes"]["device_ms_per_pass"] > 0
    assert r["overlapped_passes"]["value"] > 0.9 * r["value"]
