"""The bench line contract (task statement, section 4): the latest committed
bench line under profiles/ carries every field the driver and the judge read,
with the types they expect.  CPU only - it checks the recorded output of
`python bench.py` on the GPU box, not a new run."""
import glob
import json
import os

from tests.conftest import ROOT


def _latest_line():
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_r16_bench.json")))
    assert paths, "no committed bench line"
    with open(paths[-1]) as f:
        return json.loads(f.read().strip().splitlines()[-1]), paths[-1]


def test_latest_bench_line_has_the_contract_fields():
    r, path = _latest_line()
    for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int),
                 ("warmup", int), ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str),
                 ("dtype", str), ("data", str), ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(r[k], t), (path, k, r.get(k))
    assert r["vs_baseline"] is None  # BASELINE.md publishes no number for this metric
    assert r["unit"] == "QPs/sec" and r["dtype"] == "f64" and r["scaling"] == "weak"
    assert "workload" in r["config"] and "model" not in r["config"]
    rf = r["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s")
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert rf["traffic"] is None or rf["traffic"] > rf["algorithmic_bytes_per_launch"]
    cb = r["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    # value = QPs of all steps / wall time
    assert abs(r["value"] - r["config"]["global_batch"] * 1e3 / r["ms_per_step"]) < 1e-6 * r["value"]


def test_algorithmic_bytes_match_the_survey_figure():
    """217,736 B per MPC QP (data 188,928 + guess 11,904 + solution 16,864 + SolverOut 40)."""
    r, _ = _latest_line()
    N, nx, nu, nc = 30, 12, 4, 20
    data = 8 * ((N + 1) * (nx * nx + nu * nu + nu * nx + nx + nu + nc * nx + nc * nu + nc) +
                N * (nx * nx + nx * nu + nx) + nx)
    nz, nl, nv = (N + 1) * (nx + nu), (N + 1) * nx, (N + 1) * nc
    per_qp = data + 8 * (nz + nl + nv) + 8 * (nz + nl + 2 * nv) + 40
    assert per_qp == 217736
    assert r["roofline"]["algorithmic_bytes_per_launch"] == per_qp * r["config"]["batch_per_gpu"]
