"""GPU: `bench.py` prints exactly ONE JSON line on stdout that carries the driver contract, the roofline object and the CPU baseline
(tiny workload so the whole check takes seconds; the headline run is the same code with the default workload)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_json_contract(dev, lib):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                        "--workload", "cfg1_tiny_mlm_bs2"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                   ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict)):
        assert isinstance(d[k], typ), (k, d[k])
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic" and d["config"]["workload"] == "cfg1_tiny_mlm_bs2"
    assert abs(d["value"] - d["config"]["global_pairs"] / (d["ms_per_step"] / 1000.0)) < 1e-2 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s") and rf["peak"] > 0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and "traffic" in rf
    cb = d["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] in ("reference", "port") and isinstance(cb["sample"], str) and cb["unit"] == d["unit"]
