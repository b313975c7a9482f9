"""The bench line contract (task statement, section 4): the latest committed
bench line under profiles/ carries every field the driver and the judge read,
with the types they expect.  CPU only - it checks the recorded output of
`python bench.py` on the GPU box, not a new run."""
import glob
import json
import os

from tests.conftest import ROOT


def _latest_line():
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_line.json")))
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
    # ("issue": round 6 - the lower of the two measured ceilings is named; achieved / peak /
    # frac remain the algorithmic bytes over the HBM peak either way)
    assert rf["bound"] in ("hbm", "mfma", "issue") and rf["unit"] in ("GB/s", "TFLOP/s")
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert rf["traffic"] is None or rf["traffic"] > rf["algorithmic_bytes_per_launch"]
    cb = r["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    # value = QPs of all steps / wall time
    assert abs(r["value"] - r["config"]["global_batch"] * 1e3 / r["ms_per_step"]) < 1e-6 * r["value"]


def test_latest_bench_line_carries_the_whole_truth():
    """the roofline fraction is per step (the per-launch figure a
    named extra), replayed traffic says where it comes from, and the serial,
    time-varying / dense-row, dense (configs[1]) and receding (configs[4]) numbers
    ride in the same line, with the wall-clock window of the GPU work."""
    r, path = _latest_line()
    rf = r["roofline"]
    per_step = rf["algorithmic_bytes_per_launch"] / (r["ms_per_step"] * 1e-3) / 1e9
    assert abs(rf["achieved"] - per_step) < 1e-6 * per_step, path
    assert rf["per_launch"]["launches_in_flight"] == r["config"]["steps_in_flight"]
    assert rf["per_launch"]["achieved"] <= rf["achieved"] * 1.05
    assert (rf["traffic"] is None) == (rf["traffic_source"] is None)
    if rf["traffic"] is not None:
        assert "replayed" in rf["traffic_source"] and rf["traffic_source"].startswith("profiles/")
        # the corrected counter bytes (read side doubled, MI355X_MICROARCH.md
        # HBM section), the raw sum beside them, their ratio to the algorithmic bytes and the
        # regime the counters were taken in
        assert rf["traffic"] > rf["traffic_raw"] > rf["algorithmic_bytes_per_launch"]
        assert abs(rf["traffic_ratio"] - rf["traffic"] / rf["algorithmic_bytes_per_launch"]) < 1e-9 * rf["traffic_ratio"]
        assert "one launch at a time" in rf["traffic_regime"] or "in flight" in rf["traffic_regime"]
    cb = r["cpu_baseline"]
    assert cb["affinity_cpus"] >= 1 and cb["os_cpu_count"] >= cb["affinity_cpus"]
    assert cb["cgroup_cpu_quota"] is None or cb["cgroup_cpu_quota"] > 0
    for k in ("serial", "ltv_dense_rows", "dense", "receding"):
        assert isinstance(r[k], dict) and r[k]["value"] > 0 and r[k]["unit"] == "QPs/sec", (path, k)
    assert r["serial"]["steps_in_flight"] == 1 and r["serial"]["value"] <= r["value"] * 1.02
    assert r["ltv_dense_rows"]["all_converged"] and "workload" in r["ltv_dense_rows"]
    d = r["dense"]
    assert d["roofline"]["bound"] == "hbm" and abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-12
    assert d["kernel_ms"] > 0 and d["all_converged"]
    rc = r["receding"]
    assert rc["trajectories"] == 4096 and rc["steps"] == 200 and rc["host_syncs_per_step"] == 0
    assert rc["wall_ms_per_step"] >= rc["kernel_ms_median"] > 0 and rc["retired"] >= 0
    # no host work between steps: the wall time is the solves' own time (<= 1.5x)
    assert rc["wall_ms_per_step"] <= 1.5 * rc["kernel_ms_mean"] and rc["wall_over_kernel_sum"] <= 1.5
    g = r["gpu_leg"]
    assert g["all_gpu_work_unix"][0] <= g["timed_region_unix"][0] < g["timed_region_unix"][1] <= g["all_gpu_work_unix"][1]
    assert isinstance(r["all_converged"], bool) and r["not_converged"] >= 0


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


def test_round4_blocks_of_the_bench_line():
    """the dense block is quoted in the DEFAULT (Eigen's) elimination
    order with the opt-in orders beside it, and the line says what ONE caller sees - ms per cold
    solve of small batches and of one FBstabMpc::Solve through the C++ facade, next to the CPU
    restatement on the same QP.  (Lines older than round 4 carry neither block.)"""
    r, path = _latest_line()
    if "latency" not in r:
        return
    d = r["dense"]
    assert d["factorisation_order"].startswith("pivoted") and d["pivoted_steps_per_launch"] in (0, -1)
    for k in ("auto", "natural"):
        o = d["opt_in_orders"][k]
        assert o["value"] > 0 and o["all_converged"] and o["mean_newton_iters"] == d["mean_newton_iters"], (path, k)
    lat = r["latency"]
    for b in ("1", "16", "256", "2048"):
        e = lat["device_pointers"][b]
        assert e["ms_median"] >= e["ms_min"] > 0 and e["all_converged"], (path, b)
    assert lat["device_pointers"]["1"]["ms_median"] <= lat["device_pointers"]["2048"]["ms_median"]
    f = lat["facade_host_pointers"]
    assert "error" not in f and f["median_ms"] > 0 and f["eflag"] == 0, (path, f)
    c = lat["cpu_same_qp_single_thread"]
    assert "error" not in c and c["ms_median"] > 0 and c["newton_iters"] == f["newton_iters"]
    assert abs(lat["gpu_batch1_over_cpu_same_qp"] - lat["device_pointers"]["1"]["ms_median"] / c["ms_median"]) < 1e-9
    if r["roofline"]["traffic"] is not None:
        assert isinstance(r["roofline"]["traffic_build_matches"], bool)


def test_round_5_fields_of_the_bench_line():
    """the line says how far the kernel is from the memory wall of its own counter
    traffic (hbm_ceiling_qps = batch / (traffic / 6.29 TB/s)), how many untimed steps really ran (every lane's
    first solve is untimed: max(W, lanes)), and the launch geometry of a handle that shares the device."""
    r, path = _latest_line()
    rf = r["roofline"]
    if rf["traffic"] is not None:
        assert abs(rf["hbm_ceiling_qps"] - r["config"]["batch_per_gpu"] / (rf["traffic"] / 6.29e12)) < 1e-6 * rf["hbm_ceiling_qps"], path
        assert rf["hbm_ceiling_qps"] > r["value"] / r["n_gpus"]  # the kernel is not at that wall
    assert r["config"]["untimed_steps"] >= max(r["warmup"], min(r["config"]["steps_in_flight"], r["steps"])), path
    assert r["launch"]["workgroups"] >= 256 and r["launch"]["scratch_bytes"] > 0


def test_round_6_fields_of_the_bench_line():
    """the roofline names the ceiling the kernel is near - issue_bound_qps = batch /
    (SQ_INSTS_VALU x probed cycles per instruction / 1024 SIMDs / shader clock), replayed from the sq-counter
    summary, `bound` = the lower of that and the memory wall - and the line carries a `wide` block (the
    row-pair record instances the driver never saw).  (Lines older than round 6 carry neither.)"""
    r, path = _latest_line()
    rf = r["roofline"]
    if "issue_bound_qps" not in rf:
        return
    ib = rf["issue_bound"]
    if ib is not None:
        want = r["config"]["batch_per_gpu"] / (ib["valu_instructions_per_launch"] * ib["cycles_per_instruction"] /
                                               ib["simds"] / (ib["shader_clock_mhz"] * 1e6))
        assert abs(rf["issue_bound_qps"] - want) < 1e-6 * want, path
        assert isinstance(ib["build_matches"], bool) and ib["counter_source"].startswith("profiles/")
        if rf["traffic"] is not None:
            assert rf["bound"] == ("issue" if rf["issue_bound_qps"] < rf["hbm_ceiling_qps"] else "hbm"), path
    if "wide" in r:
        for k, w in r["wide"].items():
            assert w["value"] > 0 and w["unit"] == "QPs/sec" and w["all_converged"] in (True, False), (path, k)
            assert w["kernel"].startswith("fbstab_mpc_r32_kernel") and 0 < w["roofline"]["frac"] < 1, (path, k)
            if "in_flight" in w:  # (the same batch as a stream of launches: the headline's regime)
                f = w["in_flight"]
                assert f["value"] > 0 and f["unit"] == "QPs/sec" and f["steps_in_flight"] >= 2 and f["steps"] >= f["steps_in_flight"], (path, k)
                assert abs(f["value"] - w["batch"] * 1e3 / f["ms_per_step"]) < 1e-6 * f["value"], (path, k)
