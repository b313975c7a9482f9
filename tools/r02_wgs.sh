# developer script: pipelined bench vs resident workgroups per CU and timed steps
set -x
O=gpurun_out/r02_$1; mkdir -p $O
L=$2
for W in 4 3 2 1; do for S in 20 60; do
  FBSTAB_HIP_LIB=$L FBSTAB_HIP_WGS_PER_CU=$W python bench.py --cpu-sample 0 --steps $S > $O/wgs${W}_s$S.json 2>> $O/err.txt
done; done
python - "$O" <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/wgs*.json")):
    d = json.loads(open(f).read())
    print(f.split("/")[-1], round(d["value"]), "QP/s", round(d["ms_per_step"], 2), d["launch"]["workgroups"])
PY
