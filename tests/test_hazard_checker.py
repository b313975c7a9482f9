"""The build's hazard check (tools/check_dpp_hazards.py) on hand-made instruction streams: it has to
FIND what the record kernels' fused instructions must never meet, or its silence on the built objects
means nothing.  CPU only."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("check_dpp_hazards", os.path.join(ROOT, "tools", "check_dpp_hazards.py"))
chk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(chk)

FMAC = "v_fmac_f64_dpp v[4:5], v[2:3], v[6:7] row_newbcast:3 row_mask:0xf bank_mask:0xf"


def run(lines):
    """lines: 'L3: text' puts a label in front of an instruction"""
    ins = []
    for l in lines:
        lab, text = (l.split(": ", 1) if l.startswith("L") and ": " in l else (None, l))
        ins.append((lab, text))
    return chk.check_function("f", ins, "unit")


def test_producer_directly_before_and_one_instruction_before_are_found():
    for gap in ([], ["v_mov_b32_e32 v9, v8"]):
        n, f = run(["s_mov_b32 s0, 0", "s_mov_b32 s1, 0", "v_add_f64 v[2:3], v[10:11], v[12:13]"] + gap + [FMAC])
        assert n == 1 and len(f) == 1 and "DPP source written" in f[0]


def test_two_wait_states_are_enough_and_s_nop_counts():
    for gap in (["v_mov_b32_e32 v9, v8", "s_mov_b32 s0, 0"], ["s_nop 1"]):
        n, f = run(["s_mov_b32 s0, 0", "v_add_f64 v[2:3], v[10:11], v[12:13]"] + gap + [FMAC])
        assert n == 1 and f == []
    # the multiplier and the accumulator are ordinary operands: no wait states needed
    n, f = run(["s_nop 4", "v_add_f64 v[6:7], v[10:11], v[12:13]", "v_add_f64 v[4:5], v[10:11], v[12:13]", FMAC])
    assert f == []


def test_half_of_the_pair_and_both_operands_of_a_swap():
    n, f = run(["s_nop 4", "v_mov_b32_e32 v3, v8", FMAC])
    assert len(f) == 1
    n, f = run(["s_nop 4", "v_permlane16_swap_b32_e32 v20, v2", FMAC])
    assert len(f) == 1 and "v_permlane16_swap" in f[0]
    n, f = run(["s_nop 4", "v_readlane_b32 s2, v2, 3", "v_cmp_lt_f64_e32 vcc, v[2:3], v[8:9]", FMAC])
    assert f == []  # neither writes a vector register


def test_the_producer_at_the_bottom_of_a_loop_is_found_through_the_back_edge():
    body = ["s_nop 4", "L1: " + FMAC, "v_mov_b32_e32 v9, v8", "v_mov_b32_e32 v9, v8", "v_mov_b32_e32 v9, v8",
            "v_add_f64 v[2:3], v[10:11], v[12:13]", "s_cbranch_scc1 L1", "s_endpgm"]
    n, f = run(body)
    assert len(f) == 1 and "DPP source written" in f[0]  # add, branch, fmac: one wait state
    body[5:6] = ["v_add_f64 v[2:3], v[10:11], v[12:13]", "s_nop 0"]
    n, f = run(body)
    assert f == []


def test_the_producer_before_an_unconditional_branch_is_found_through_the_jump():
    """a path entered through `s_branch Lx` starts AT that branch; the instructions before
    it belong to the path (the branch itself is one wait state), as with a conditional branch."""
    for br in ("s_branch L1", "s_cbranch_scc0 L1"):
        n, f = run(["s_nop 4", "v_mul_f64 v[2:3], v[10:11], v[12:13]", br, "L9: s_nop 4", "s_nop 4", "L1: " + FMAC,
                    "s_endpgm"])
        assert n == 1 and len(f) == 1 and "DPP source written" in f[0], (br, f)
        # the branch counts as one wait state: one more instruction in front of it is enough
        n, f = run(["s_nop 4", "v_mul_f64 v[2:3], v[10:11], v[12:13]", "s_mov_b32 s0, 0", br, "L9: s_nop 4", "s_nop 4",
                    "L1: " + FMAC, "s_endpgm"])
        assert f == [], (br, f)
    # ... also when the label sits on an instruction BEFORE the fused one
    n, f = run(["s_nop 4", "v_mul_f64 v[2:3], v[10:11], v[12:13]", "s_branch L1", "L9: s_nop 4", "s_nop 4",
                "L1: s_mov_b32 s0, 0", FMAC, "s_endpgm"])
    assert f == []  # mul, branch, s_mov: two wait states
    n, f = run(["s_nop 4", "v_rcp_f64_e32 v[6:7], v[10:11]", "s_branch L1", "L9: s_nop 4", "s_nop 4", "L1: " + FMAC,
                "s_endpgm"])
    assert f == []  # a trans result needs one wait state: the branch is one
    # linear code behind an unconditional branch is still not a fall-through path
    n, f = run(["s_nop 4", "v_mul_f64 v[2:3], v[10:11], v[12:13]", "s_branch L7", "L1: " + FMAC, "L7: s_endpgm"])
    assert f == []


def test_an_object_without_a_code_object_is_a_finding(tmp_path):
    """the build gate must not pass with nothing checked."""
    import subprocess, sys
    src = tmp_path / "x.c"
    src.write_text("int f(void) { return 1; }\n")
    obj = tmp_path / "x.o"
    subprocess.check_call(["gcc", "-c", "-o", str(obj), str(src)])
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_dpp_hazards.py"), str(obj)],
                       capture_output=True, text=True)
    assert r.returncode == 1 and "no amdgcn code object" in r.stdout


def test_exec_written_by_a_compare_needs_five_and_a_trans_result_one():
    n, f = run(["v_cmpx_le_i32_e32 vcc, 3, v30", "s_mov_b64 exec, s[4:5]", "s_nop 2", FMAC])
    assert len(f) == 1 and "EXEC" in f[0]
    n, f = run(["v_cmpx_le_i32_e32 vcc, 3, v30", "s_mov_b64 exec, s[4:5]", "s_nop 4", FMAC])
    assert f == []
    n, f = run(["s_nop 4", "v_rcp_f64_e32 v[6:7], v[8:9]", FMAC])
    assert len(f) == 1 and "trans" in f[0]
    n, f = run(["s_nop 4", "v_rcp_f64_e32 v[6:7], v[8:9]", "s_nop 0", FMAC])
    assert f == []


def test_paths_that_leave_the_function_are_reported():
    n, f = run([FMAC])
    assert len(f) == 1 and "function entry" in f[0]
    n, f = run(["s_nop 4", "s_swappc_b64 s[30:31], s[4:5]", "s_mov_b32 s0, 0", FMAC])
    assert len(f) == 1 and "call returns" in f[0]
    n, f = run(["s_nop 4", "s_swappc_b64 s[30:31], s[4:5]", "s_mov_b32 s0, 0", "s_mov_b32 s0, 0", FMAC])
    assert f == []
    # code behind an unconditional jump that no label leads to: only an indirect jump gets there
    n, f = run(["s_nop 4", "s_setpc_b64 s[30:31]", "s_mov_b32 s0, 0", FMAC])
    assert len(f) == 1 and "without a label" in f[0]
    n, f = run(["s_nop 4", "s_cbranch_scc1 L2", "s_branch L9", "L2: s_mov_b32 s0, 0", "s_mov_b32 s1, 0", FMAC, "L9: s_endpgm"])
    assert f == []


def test_the_built_objects_pass_if_they_are_here():
    import glob
    objs = sorted(glob.glob(os.path.join(ROOT, "fbstab_amd", "csrc", "build", "libfbstab_hip", "rec_*.o")))
    if not objs or not os.path.exists(chk.OBJDUMP):
        import pytest
        pytest.skip("no built record objects in this tree")
    import subprocess, sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_dpp_hazards.py"), objs[0]], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "no hazard found" in r.stdout


_HAZARD_KERNEL = r"""
#include <hip/hip_runtime.h>
__global__ void k(const double* in, double* out) {
  const int t = threadIdx.x;
  double a = in[t], b = in[64 + t], acc = in[128 + t];
  double src;
  asm volatile("v_add_f64 %0, %1, %2" : "=v"(src) : "v"(a), "v"(b));
  %s
  asm volatile("v_fmac_f64_dpp %%0, %%1, %%2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(b));
  out[t] = acc;
}
"""


def test_a_compiled_object_with_the_hazard_fails_and_with_the_wait_states_passes(tmp_path):
    """End to end, the way the Makefile uses it: hipcc object in, exit code out."""
    import shutil, subprocess, sys
    if not shutil.which("hipcc") and not os.path.exists("/opt/rocm/bin/hipcc"):
        import pytest
        pytest.skip("no hipcc here")
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    rcs = {}
    for name, gap in (("bad", ""), ("good", 'asm volatile("s_nop 1");')):
        src = tmp_path / f"{name}.hip"
        src.write_text(_HAZARD_KERNEL.replace("%s", gap).replace("%%", "%"))
        obj = tmp_path / f"{name}.o"
        subprocess.check_call([hipcc, "-O3", "--offload-arch=gfx950", "-c", "-o", str(obj), str(src)], stderr=subprocess.DEVNULL)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_dpp_hazards.py"), str(obj)], capture_output=True, text=True)
        rcs[name] = (r.returncode, r.stdout)
    assert rcs["bad"][0] == 1 and "DPP source written by `v_add_f64" in rcs["bad"][1], rcs["bad"][1]
    assert rcs["good"][0] == 0 and "1 v_fmac_f64_dpp instructions checked" in rcs["good"][1], rcs["good"][1]
