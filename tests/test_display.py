"""The reference's per-iteration display (Display::ITER / ITER_DETAILED,
fbstab_algorithm-impl.h:411-541).

tests/golden/reference_display.json holds the text the reference's own print
functions produce for the known-answer problems (generated through
oracle/_ref, see tests/golden/make_display_golden.py).  CPU tests pin the
oracle's trace records (and the formatter the GPU tests use) against that
text exactly; the GPU tests compare the device's records and the C++ facade's
text against them."""
import json
import os

import numpy as np
import pytest

from oracle.oracle_py import default_options
from tests import helpers as H
from tests.conftest import GOLDEN


@pytest.fixture(scope="module")
def display_golden():
    with open(os.path.join(GOLDEN, "reference_display.json")) as f:
        return json.load(f)["cases"]


def _problem(kats, case):
    k = kats[case["kind"] + "_end_to_end"][case["index"]]
    return H.dense_from_kat(k) if case["kind"] == "dense" else H.mpc_from_kat(k)


def test_oracle_records_reproduce_the_reference_display(oracle, kats, display_golden):
    for case in display_golden:
        o = default_options(display_level=case["level"])
        r = oracle.solve_display(_problem(kats, case), opts=o)
        out, rec = r[4][0], r[6]
        assert (out["eflag"], out["newton_iters"], out["prox_iters"]) == (
            case["eflag"], case["newton_iters"], case["prox_iters"])
        assert H.format_display(rec, case["level"], out, o) == case["text"], case["name"]


def test_golden_display_is_what_the_reference_prints(ref_oracle, kats, display_golden):
    for case in display_golden:
        o = default_options(display_level=case["level"])
        r = ref_oracle.solve_display(_problem(kats, case), opts=o)
        assert H.normalise_time(r[5]) == case["text"], case["name"]


def test_display_off_and_final_levels_of_the_reference(ref_oracle, kats):
    """OFF prints nothing, FINAL only the summary block (impl:493-541)."""
    p = H.dense_from_kat(kats["dense_end_to_end"][0])
    assert ref_oracle.solve_display(p, opts=default_options(display_level=0))[5] == ""
    t = ref_oracle.solve_display(p, opts=default_options(display_level=1))[5]
    assert t.startswith("\nOptimization completed!  Exit code: Success\n") and "prox iter" not in t


def _device_trace(kind, p):
    from fbstab_amd import hip_api
    a = {k: np.ascontiguousarray(v[:1]) for k, v in p.arrays.items()}
    z, l, v, y = (np.zeros((1, n)) for n in (p.nz, p.nl, p.nv, p.nv))
    if kind == "dense":
        s = hip_api.FBstabDenseBatch(p.nz, p.nl, p.nv, max_batch=1)
    else:
        s = hip_api.FBstabMpcBatch(*p.sizes(), max_batch=1)
    out, rec = s.SolveTraced(a, z, l, v, y)
    return out[0], rec, (z, l, v, y)


def _records_agree(dev, ref, what):
    assert len(dev) == len(ref), (what, len(dev), len(ref))
    assert np.array_equal(dev[:, :3], ref[:, :3]), what  # kinds, iteration numbers
    # FP tolerance: the device evaluates the same formulas in a different order
    # (DESIGN.md 3): 1e-6 relative, 1e-7 of the largest residual of the solve
    # absolute (the linear blocks of the inner residual are rounding noise of
    # that size after every Newton step).  Residuals below 1e-3 of that scale
    # are what a converged Newton iteration leaves behind - quadratic convergence
    # squares the rounding difference of the step before - and only have to
    # agree within a factor of two.
    d, r = dev[:, 3:], ref[:, 3:]
    scale = _scale(ref)
    err = np.abs(d - r)
    tight = err <= 1e-7 * scale + 1e-6 * np.abs(r)
    loose = (np.abs(r) < 1e-3 * scale) & (err <= 0.5 * np.maximum(np.abs(d), np.abs(r)))
    bad = ~(tight | loose).all(axis=1)
    assert not bad.any(), (what, dev[bad], ref[bad])


def _scale(rec):
    return max(1.0, float(np.abs(rec[:, 3:]).max()))


@pytest.mark.gpu
def test_device_trace_matches_the_oracle_on_the_known_answer_problems(oracle, kats, display_golden):
    done = set()
    for case in display_golden:
        key = (case["kind"], case["index"])
        if key in done:
            continue
        done.add(key)
        p = _problem(kats, case)
        ref = oracle.solve_display(p, opts=default_options())
        out, rec, x = _device_trace(case["kind"], p)
        assert (out["eflag"], out["newton_iters"], out["prox_iters"]) == (
            case["eflag"], case["newton_iters"], case["prox_iters"]), case["name"]
        _records_agree(rec, ref[6], case["name"])
        # and the formatted text is the reference's, number for number
        for level in (2, 3):
            g = [c for c in display_golden if (c["kind"], c["index"], c["level"]) == (*key, level)][0]
            o = default_options(display_level=level)
            ok, why = H.display_texts_agree(H.format_display(rec, level, out, o), g["text"],
                                            atol=1e-7 * _scale(ref[6]))
            assert ok, (case["name"], level, why)


@pytest.mark.gpu
def test_device_trace_on_the_synthetic_workloads(oracle):
    """BASELINE shapes: one QP of the batched MPC workload (traced on the
    flat-vector kernel although batches of this shape run on the record kernel)
    and one of the dense workload; the traced solve returns the same solution
    as the batch call."""
    from fbstab_amd import hip_api
    from tools import fixtures as fx
    for kind, p in (("mpc", fx.synthetic_mpc_batch(1, first_id=5)),
                    ("dense", fx.synthetic_dense_batch(1, 50, 10, 100))):
        ref = oracle.solve_display(p, opts=default_options())
        out, rec, x = _device_trace(kind, p)
        assert out["eflag"] == ref[4]["eflag"][0] == 0
        assert out["prox_iters"] == ref[4]["prox_iters"][0]
        assert int(out["newton_iters"]) == int(ref[4]["newton_iters"][0])
        _records_agree(rec, ref[6], kind)
        assert rec[-1, 0] == 5 and rec[-1, 1] == 0  # FINAL record, SUCCESS
        assert np.hypot.reduce(rec[-1, 3:6]) == pytest.approx(out["residual"], rel=1e-9, abs=1e-15)
        for a, b in zip(x[:3], ref[:3]):
            assert np.abs(a[0] - b).max() <= 1e-5 * (1 + np.abs(b).max())
        # batch call on the same QP
        a = {k: np.ascontiguousarray(v[:1]) for k, v in p.arrays.items()}
        z, l, v, y = (np.zeros((1, n)) for n in (p.nz, p.nl, p.nv, p.nv))
        s = (hip_api.FBstabMpcBatch(*p.sizes(), max_batch=1) if kind == "mpc"
             else hip_api.FBstabDenseBatch(p.nz, p.nl, p.nv, max_batch=1))
        s.Solve(a, z, l, v, y)
        assert np.abs(z - x[0]).max() <= 1e-5 * (1 + np.abs(z).max())


@pytest.mark.gpu
def test_traced_solve_argument_errors():
    from fbstab_amd import hip_api
    from tools import fixtures as fx
    p = fx.synthetic_dense_batch(1, 20, 5, 40)
    s = hip_api.FBstabDenseBatch(p.nz, p.nl, p.nv, max_batch=1)
    a = {k: np.ascontiguousarray(v[:1]) for k, v in p.arrays.items()}
    z, l, v, y = (np.zeros((1, n)) for n in (p.nz, p.nl, p.nv, p.nv))
    with pytest.raises(hip_api.FBstabHipError):
        s.SolveTraced(a, z, l, v, y, capacity=0)
    out, rec = s.SolveTraced(a, z, l, v, y, capacity=3)  # truncated, not overrun
    assert len(rec) == 3 and out["eflag"][0] == 0


@pytest.mark.gpu
def test_final_summary_numbers_from_the_batch_kernels(oracle, kats, display_golden):
    """Display::FINAL, the reference's default level (impl:493-541), through
    fbstab_hip_*_solve_batch_final: the batch kernels solve (same iteration counts as
    solve_batch), a small kernel evaluates |rz| |rl| |rv| and the tolerance at the
    returned point.  For the known-answer problems that end in SUCCESS these are the
    numbers of the reference's summary block (the FINAL record of the oracle's trace,
    whose text tests above pin to the reference's own output); on the BASELINE
    workloads the three blocks add up to SolverOut::residual for every QP."""
    from fbstab_amd import hip_api
    from tools import fixtures as fx
    done = set()
    for case in display_golden:
        key = (case["kind"], case["index"])
        if key in done or case["eflag"] != 0:
            continue
        done.add(key)
        p = _problem(kats, case)
        ref = oracle.solve_display(p, opts=default_options())
        a = {k: np.ascontiguousarray(v[:1]) for k, v in p.arrays.items()}
        z, l, v, y = (np.zeros((1, n)) for n in (p.nz, p.nl, p.nv, p.nv))
        s = (hip_api.FBstabDenseBatch(p.nz, p.nl, p.nv, max_batch=1) if case["kind"] == "dense"
             else hip_api.FBstabMpcBatch(*p.sizes(), max_batch=1))
        out, nrm = s.SolveFinal(a, z, l, v, y)
        s.close()
        assert (out["eflag"][0], out["newton_iters"][0], out["prox_iters"][0]) == (
            case["eflag"], case["newton_iters"], case["prox_iters"]), case["name"]
        fin = ref[6][-1]
        assert fin[0] == 5  # FBSTAB_TRACE_FINAL
        _records_agree(np.concatenate([fin[:3], nrm[0], [0.0]])[None, :], fin[None, :], case["name"])
        assert nrm[0, 3] == pytest.approx(fin[6], rel=1e-12)
    # batches on the record kernel and the one-wavefront dense kernel, device memory
    import torch
    dev = torch.device("cuda:0")
    for kind, p in (("mpc", fx.synthetic_mpc_batch(96, first_id=300)), ("dense", fx.synthetic_dense_batch(96, 50, 10, 100))):
        s = (hip_api.FBstabMpcBatch(*p.sizes(), max_batch=96) if kind == "mpc"
             else hip_api.FBstabDenseBatch(p.nz, p.nl, p.nv, max_batch=96))
        data = {k: torch.from_numpy(np.ascontiguousarray(a)).to(dev) for k, a in p.arrays.items()}
        mk = lambda n: torch.zeros((96, n), dtype=torch.float64, device=dev)
        z, l, v, y = mk(p.nz), mk(p.nl), mk(p.nv), mk(p.nv)
        out, nrm = s.SolveFinal(data, z, l, v, y)
        out = hip_api.out_to_numpy(out)
        nrm = nrm.cpu().numpy()
        z2, l2, v2, y2 = mk(p.nz), mk(p.nl), mk(p.nv), mk(p.nv)
        out2 = hip_api.out_to_numpy(s.Solve(data, z2, l2, v2, y2))
        s.close()
        assert (out["eflag"] == 0).all() and np.array_equal(out["newton_iters"], out2["newton_iters"])
        assert torch.equal(z, z2) and torch.equal(v, v2)
        tot = np.sqrt((nrm[:, :3] ** 2).sum(axis=1))
        # the kernel's own residual is carried through the Newton steps of the last
        # subproblem's successor evaluation; the pass evaluates it afresh: rounding apart
        np.testing.assert_allclose(tot, out["residual"], rtol=1e-6, atol=1e-9)
        assert (nrm[:, 3] >= 1e-6).all() and (tot <= nrm[:, 3]).all()
