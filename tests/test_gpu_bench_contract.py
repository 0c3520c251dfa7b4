"""bench.py's output contract (one JSON line with the driver's keys, roofline and cpu_baseline objects) and the N > 1
code path (two ranks sharing the one GPU of the test box through --rehearse-one-gpu: partition, bind, exchange step)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline"}


def _last_json_line(text):
    lines = [ln for ln in text.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, text[-2000:]
    return json.loads(lines[0])


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_single_gpu_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--spp", "32"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _last_json_line(out.stdout)
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["unit"] == "Msamples/s" and d["higher_is_better"] is True and d["vs_baseline"] is None and d["dtype"] == "f32"
    assert d["value"] > 1000 and abs(d["value"] - 1920 * 1080 * 32 / (d["ms_per_step"] * 1e-3) / 1e6) < 1e-6 * d["value"]
    r = d["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(r) and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert 200 < r["algorithmic_flops_per_sample"] < 300          # F = 251.7 on C2
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert "workload" in d["config"] and "model" not in d["config"]


@pytest.mark.parametrize("workload,partition", [("c2", "samples"), ("c4", "tiles")])
def test_two_ranks_rehearsal(workload, partition):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--spp", "8", "--workload", workload, "--rehearse-one-gpu"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _last_json_line(out.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == ("weak" if partition == "samples" else "strong")
    assert ("sample-range x2" if partition == "samples" else "tiles round-robin x2") in d["config"]["partition"]
    w, h = (1920, 1080) if workload == "c2" else (3840, 2160)
    work = (2 if partition == "samples" else 1) * w * h * 8
    assert abs(d["value"] - work / (d["ms_per_step"] * 1e-3) / 1e6) < 1e-6 * d["value"]


def test_plain_invocation_with_gpus_2_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (the form the driver uses for N = 1): the parent spawns the
    ranks itself, relays the one JSON line and the exit code."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                          "--spp", "8", "--rehearse-one-gpu"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _last_json_line(out.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "gloo" in d["exchange"]["how"]
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--spp", "1", "--workload", "nonsense"], capture_output=True, text=True, timeout=120, cwd=ROOT, env=env)
    assert bad.returncode != 0
