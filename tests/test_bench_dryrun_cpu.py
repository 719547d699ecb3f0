"""`python bench.py --gpus N` without GPUs: launcher, sharded step, timing, rc and the one-JSON-line contract (gloo, toy
model, kernel shims: tests/bench_dryrun.py).  The driver runs the same code with RCCL on 8 MI355X at round end."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def run(gpus, env_extra=None, timeout=600):
    env = dict(os.environ, **(env_extra or {}))
    env.pop("WORLD_SIZE", None)
    return subprocess.run([sys.executable, str(ROOT / "tests" / "bench_dryrun.py"), "--gpus", str(gpus), "--steps", "1",
                           "--warmup", "0", "--frames", "12"], capture_output=True, text=True, timeout=timeout, env=env,
                          cwd=str(ROOT))


@pytest.mark.parametrize("gpus", [1, 2])
def test_bench_job_prints_exactly_one_json_line(gpus):
    r = run(gpus)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    # the collective library may print its own banner through C stdio (gloo here, RCCL on the GPU box: bench.py drains
    # it on every rank before the last barrier): exactly ONE line is a JSON object, and it is the LAST line of stdout
    js = [l for l in lines if l.lstrip().startswith("{")]
    assert len(js) == 1 and lines[-1] == js[0], lines
    d = json.loads(js[0])
    assert d["n_gpus"] == gpus and d["steps"] == 1 and d["unit"] == "frames/s" and d["value"] > 0
    assert d["metric"].startswith("video frames/sec fwd")
    assert d["config"]["parallelism"] == ("single GPU" if gpus == 1 else "sequence-sharded x2 (RCCL)")
    for key in ("ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "roofline", "rooflines"):
        assert key in d
    # every HBM companion of the scan is event-timed inside the timed step
    kinds = " | ".join(e["kernel"] for e in d["rooflines"])
    for name in ("ssd_scan", "conv1d_xbc", "rmsnorm_gated", "Mamba mixer trio", "rmsnorm_kernel", "layernorm"):
        assert name in kinds, (name, kinds)


def test_bench_job_propagates_a_failing_rank():
    r = run(2, {"TV_BENCH_DRYRUN_FAIL_RANK": "1"})
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.strip().startswith("{")], "a failed job must not print a result line"
