#!/usr/bin/env python3
"""Build gate and developer tool: total registers (arch VGPR + AGPR, as allocated: rounded up to the hardware
granule of 8) of every kernel in an object / library, from the code object's metadata.

Why it matters (LABNOTES R6.3, measured on MI355X): the record kernels run one wavefront per SIMD and hold it for
the whole launch.  A SIMD has 512 registers per lane; a wavefront that allocates 504 of them leaves 8, and NO other
kernel's wavefront fits beside it - not even the 16-register fill kernel behind hipMemsetAsync / torch's zero_() -
so a stream of batches on several HIP streams degenerates into one launch after the other (headline 630 k ->
250 k QP/s).  At <= 496 allocated registers 16 stay free and the small kernels between two solves slip in.

usage: tools/check_vgpr_budget.py [--max N] [--only SUBSTR] [--warn-only] <file.o|.so> ...
With --max, exits 1 if a kernel whose (mangled) name contains SUBSTR allocates more than N registers (--warn-only:
says so and exits 0 - the <24,8,*> instances, whose noinline passes the kernel attribute does not reach)."""
import os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def code_objects(path, tmp):
    local = os.path.join(tmp, os.path.basename(path))
    with open(path, "rb") as f, open(local, "wb") as g:
        g.write(f.read())
    subprocess.check_call([OBJDUMP, "--offloading", local], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return [os.path.join(tmp, n) for n in sorted(os.listdir(tmp)) if "amdgcn" in n]


def kernels(co):
    txt = subprocess.check_output([READELF, "--notes", co], text=True)
    out, cur = [], {}
    for line in txt.splitlines():
        m = re.match(r"\s*(?:- )?\.(\w+):\s+(.*)$", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == "agpr_count":   # (first key of a kernel's record)
            if cur.get("name"):
                out.append(cur)
            cur = {"agpr_count": int(v)}
        elif k in ("name", "symbol"):
            cur.setdefault(k, v)
        elif k in ("vgpr_count", "private_segment_fixed_size", "vgpr_spill_count", "sgpr_spill_count"):
            cur[k] = int(v)
    if cur.get("name"):
        out.append(cur)
    return [k for k in out if "vgpr_count" in k]


def main():
    args = sys.argv[1:]
    limit, only = None, ""
    warn_only = "--warn-only" in args
    if warn_only:
        args.remove("--warn-only")
    if "--max" in args:
        i = args.index("--max"); limit = int(args[i + 1]); del args[i:i + 2]
    if "--only" in args:
        i = args.index("--only"); only = args[i + 1]; del args[i:i + 2]
    bad = 0
    for path in args:
        with tempfile.TemporaryDirectory() as tmp:
            for co in code_objects(path, tmp):
                for k in kernels(co):
                    alloc = (k["vgpr_count"] + 7) // 8 * 8
                    name = subprocess.run(["c++filt", k["name"]], capture_output=True, text=True).stdout.strip()
                    name = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0]
                    over = limit is not None and only in k["name"] and alloc > limit
                    bad += over
                    print(f"{os.path.basename(path):24s} {name[-64:]:64s} regs {k['vgpr_count']:3d} -> {alloc:3d} allocated"
                          f"  agpr {k['agpr_count']:3d} spill {k.get('vgpr_spill_count', 0):3d} scratch {k.get('private_segment_fixed_size', 0):5d}"
                          + ("   <-- over the budget" if over else ""))
    if bad:
        print(f"{bad} kernel(s) over the register budget of {limit}: no other kernel's wavefront fits beside them (LABNOTES R6.3)")
        sys.exit(0 if warn_only else 1)


if __name__ == "__main__":
    main()
