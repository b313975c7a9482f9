#!/bin/bash
# Developer tool (GPU box): tools/ab_headline.sh with a summary (median / mean / min / max per library).
# usage: tools/ab_stats.sh <out dir under gpurun_out> <rounds> <lib.so> ...
D=gpurun_out/$1; shift
mkdir -p $D
bash tools/ab_headline.sh "$@" > $D/ab.log 2>&1
python3 - $D/ab.log <<'PY'
import re, sys, collections
d = collections.defaultdict(list)
for l in open(sys.argv[1]):
    m = re.match(r"(\S+)\s+rep \d+:\s+(\d+) QP/s.*newton ([0-9.]+)", l)
    if m: d[m.group(1)].append((int(m.group(2)), m.group(3)))
for k, v in d.items():
    q = sorted(x[0] for x in v)
    print("%-24s n=%d median %d mean %d min %d max %d  newton %s" % (k, len(q), q[len(q)//2], sum(q)/len(q), q[0], q[-1], v[0][1]))
PY
