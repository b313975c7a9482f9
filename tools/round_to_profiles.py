#!/usr/bin/env python3
"""Developer tool: turn the output of tools/round_profile.sh (gpurun_out/<tag>/) into the
summaries kept under profiles/ (<prefix>_bench_line.json, _bench_under_rocprof.json,
_r16_kernel_stats_pipelined_bench.csv, _r16_traffic.json, _r16_sq_counters.json,
_dense_wave_counters.json, _r16_wave_time_shares.txt, _dense_wave_time_shares.txt,
_sharded_one_device_rehearsal.json).  usage: tools/round_to_profiles.py <tag> <prefix> <build note>"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, prefix, build = sys.argv[1], sys.argv[2], sys.argv[3]
src = os.path.join(ROOT, "gpurun_out", tag)
dst = lambda name: os.path.join(ROOT, "profiles", f"{prefix}_{name}")

for a, b in (("bench_line.json", "bench_line.json"), ("bench_under_rocprof.json", "bench_under_rocprof.json"),
             ("kernel_stats_pipelined_bench.csv", "r16_kernel_stats_pipelined_bench.csv"),
             ("mpc_wave_time_shares.txt", "r16_wave_time_shares.txt"), ("dense_wave_time_shares.txt", "dense_wave_time_shares.txt"),
             ("sharded_rehearsal.json", "sharded_one_device_rehearsal.json")):
    p = os.path.join(src, a)
    if os.path.exists(p) and os.path.getsize(p) > 0:
        shutil.copy(p, dst(b))

line = json.loads(open(os.path.join(src, "bench_line.json")).read().strip().splitlines()[-1])
m = json.load(open(os.path.join(src, "pmc_mpc", "summary.json")))
B = 8192
newton = line["fp64"]["mean_newton_iters"] * B
traffic = {
    "note": "rocprofv3 --pmc passes (tools/pmc_lib.sh: FETCH_SIZE, WRITE_SIZE and the SQ / cache sets in separate processes, "
            "--kernel-trace only, timeout-wrapped) over tools/variant_bench.py 8192 1: the last dispatch of "
            "fbstab_mpc_r16_kernel<12,4,20> (8192 QPs, one launch at a time - the profiler serialises dispatches, so this is the "
            "regime of the bench line's `serial` block; the pipelined headline runs the same kernel on the same data, eight launches "
            "sharing the GPU). FETCH_SIZE / WRITE_SIZE in KiB as reported. MI355X_MICROARCH.md (HBM section): FETCH_SIZE tallies "
            "16-byte-per-lane reads at half their bytes on gfx950 - the record reads are such reads, and the per-slot ledger of "
            "DESIGN.md 4.1 confirms the factor - so the corrected figure doubles the read side; WRITE_SIZE is exact.",
    "batch": B, "build": build, "library_sha256": m.get("library_sha256"),
    "regime": "one launch at a time (rocprofv3 --pmc serialises dispatches; tools/variant_bench.py)",
    "FETCH_SIZE_KiB": m["FETCH_SIZE"], "WRITE_SIZE_KiB": m["WRITE_SIZE"],
    "hbm_bytes_per_launch_raw": m["hbm_bytes_per_launch_raw"],
    "hbm_bytes_per_launch_fetch_doubled": m["hbm_bytes_per_launch_fetch_doubled"],
    "algorithmic_bytes_per_launch": 217736 * B, "kernel_ms_under_pmc": m["kernel_ms_under_pmc"],
}
json.dump(traffic, open(dst("r16_traffic.json"), "w"), indent=1)
sq = {k: v for k, v in m.items() if k.startswith(("SQ_", "TCC_", "TCP_"))}
sq["note"] = "same passes as " + f"profiles/{prefix}_r16_traffic.json" + " (one launch at a time); SQ_WAVE_CYCLES, SQ_ACTIVE_INST_*, SQ_WAIT_* count quad-cycles"
sq["build"] = build
sq["library_sha256"] = m.get("library_sha256")
sq["derived"] = {"issuing_share_of_wave_cycles": m["SQ_ACTIVE_INST_ANY"] / m["SQ_WAVE_CYCLES"],
                 "waiting_share": m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"],
                 "valu_instructions_per_newton_step": m["SQ_INSTS_VALU"] / newton,
                 "l2_hit_rate": m["TCC_HIT"] / m["TCC_REQ"]}
sq["batch"] = B
# the shader clock the wavefronts ran at (tools/stamp_report.py: s_memtime against the 100 MHz counter), the
# figure bench.py's issue_bound_qps divides by
try:
    import re
    txt = open(os.path.join(src, "mpc_wave_time_shares.txt")).read()
    mm = re.search(r"mean shader clock over the wavefronts' lifetimes: (\d+) MHz", txt)
    if mm:
        sq["mean_shader_clock_mhz"] = float(mm.group(1))
        sq["mean_shader_clock_source"] = f"profiles/{prefix}_r16_wave_time_shares.txt (diagnostic build, same sources)"
except OSError:
    pass
json.dump(sq, open(dst("r16_sq_counters.json"), "w"), indent=1)

dpath = os.path.join(src, "pmc_dense", "summary.json")
if os.path.exists(dpath):
    d = json.load(open(dpath))
    dn = line["dense"]["mean_newton_iters"] * 4096
    out = {"note": "rocprofv3 --pmc passes (tools/pmc_lib.sh with PMC_PROG=tools/dense_bench.py) over one launch of "
                   "fbstab_dense_wave_kernel on BASELINE configs[1] (batch 4096, nz=50 nl=10 nv=100) in the DEFAULT elimination "
                   "order (Eigen's rule at every step); FETCH_SIZE / WRITE_SIZE in KiB as reported; SQ_WAVE_CYCLES, "
                   "SQ_ACTIVE_INST_*, SQ_WAIT_* count quad-cycles.",
           "order": "pivoted", "build": build, "library_sha256": d.get("library_sha256")}
    out.update({k: v for k, v in d.items() if k.startswith(("SQ_", "TCC_", "TCP_", "FETCH", "WRITE", "kernel_ms"))})
    out["derived"] = {"issuing_share_of_wave_cycles": d["SQ_ACTIVE_INST_ANY"] / d["SQ_WAVE_CYCLES"],
                      "waiting_share": d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"],
                      "valu_instructions_per_newton_iteration": d["SQ_INSTS_VALU"] / dn,
                      "vmem_reads_per_newton_iteration": d["SQ_INSTS_VMEM_RD"] / dn,
                      "hbm_bytes_per_launch_raw": d["hbm_bytes_per_launch_raw"],
                      "hbm_bytes_per_launch_fetch_doubled": d["hbm_bytes_per_launch_fetch_doubled"],
                      "hbm_bytes_per_newton_iteration_raw": d["hbm_bytes_per_launch_raw"] / dn,
                      "l2_hit_rate": d["TCC_HIT"] / d["TCC_REQ"]}
    json.dump(out, open(dst("dense_wave_counters.json"), "w"), indent=1)
print("written:", sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.startswith(prefix + "_")))
