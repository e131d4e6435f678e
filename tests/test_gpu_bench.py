"""bench.py's contract, run as the driver runs it: one JSON line on stdout with the agreed keys; `--gpus 2` started as a
plain script launches its own two ranks (here both on the one GPU of the box, gloo between them) and reports the whole
job; strong scaling over walkers and over sources."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

KEYS = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline", "cpu_baseline"]


def run_bench(*flags, timeout=600):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(flags), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, env=env, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-3000:]
    lines = r.stdout.decode().strip().splitlines()
    assert len(lines) == 1, lines                       # ONE line on stdout, whatever RCCL / torch print
    return json.loads(lines[0])


def test_one_gpu_line_has_the_contract_keys():
    d = run_bench("--steps", "3", "--warmup", "1", "--nsrc", "40000", "--walkers", "64", "--cpu-budget", "1", "--no-extras")
    for k in KEYS:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["dtype"] == "f64" and "synthetic" in d["data"]
    assert d["value"] > 0 and abs(d["value"] * d["ms_per_step"] * 1e-3 - 64) < 1e-6 * 64        # walkers per step / time per step
    rf, cb = d["roofline"], d["cpu_baseline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert 0 < rf["frac"] < 1 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0
    assert cb["max_rel_diff_gpu_vs_port"] < 1e-12                                              # the baseline doubles as a parity check


@pytest.mark.parametrize("extra,scaling,shard", [([], "weak", "walkers"),
                                                 (["--scaling", "strong", "--shard", "walkers"], "strong", "walkers"),
                                                 (["--scaling", "strong", "--shard", "sources"], "strong", "sources")])
def test_two_ranks_self_launched(extra, scaling, shard):
    d = run_bench("--gpus", "2", "--share-gpu", "--backend", "gloo", "--steps", "2", "--warmup", "1", "--nsrc", "40000",
                  "--walkers", "64", "--no-cpu-baseline", *extra)
    assert d["n_gpus"] == 2 and d["scaling"] == scaling
    assert d["config"]["shard"] == shard and ("all-reduce" if shard == "sources" else "all-gather") in d["config"]["parallelism"], d["config"]
    total = 128 if scaling == "weak" else 64                                                    # whole-job walkers per step
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 - total) < 1e-6 * total
    if scaling == "weak":
        ss = d["strong_scaling"]                                                                # the other axis rides along, both ways
        for sh in ("walkers", "sources"):
            assert ss[sh]["value"] > 0 and ss[sh]["walkers_total"] == 64, ss
            assert abs(ss[sh]["value"] * ss[sh]["ms_per_step"] * 1e-3 - 64) < 1e-6 * 64
