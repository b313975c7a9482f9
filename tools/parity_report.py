#!/usr/bin/env python3
"""Developer tool (GPU box): where the device's Newton counts differ from the
oracle's, id by id - the restricted-line-search batch of
test_line_search_trial_limits_on_the_record_kernel and the eight 8192-id shards of
BASELINE configs[3] (argv: number of shards, default 8)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fbstab_amd import hip_api as hip  # noqa: E402
from oracle.oracle_py import Oracle, default_options  # noqa: E402
from tools import fixtures as fx  # noqa: E402

orc = Oracle(False)


def run(p, o):
    s = hip.FBstabMpcBatch(*p.sizes(), max_batch=p.batch)
    h = hip.Options()
    for name, _ in h._fields_:
        setattr(h, name, getattr(o, name))
    s.UpdateOptions(h)
    B = p.batch
    z = np.zeros((B, p.nz)); l = np.zeros((B, p.nl)); v = np.zeros((B, p.nv)); y = np.zeros((B, p.nv))
    out = s.Solve({k: np.ascontiguousarray(a) for k, a in p.arrays.items()}, z, l, v, y)
    s.close()
    c = orc.solve_mpc(p, opts=o, nthreads=orc.num_threads())
    return out, c[4]


def report(tag, first_id, og, oc):
    dn = og["newton_iters"].astype(int) - oc["newton_iters"].astype(int)
    ids = np.nonzero(dn)[0]
    print(f"{tag}: eflag equal {np.array_equal(og['eflag'], oc['eflag'])}, prox equal "
          f"{np.array_equal(og['prox_iters'], oc['prox_iters'])}, newton differs on {len(ids)} of {len(dn)}: "
          + ", ".join(f"id {first_id + i}: {og['newton_iters'][i]} vs {oc['newton_iters'][i]} (eflag {oc['eflag'][i]})" for i in ids))


for max_ls in (1, 2, 5, 9):
    p = fx.synthetic_mpc_batch(96, first_id=52000)
    og, oc = run(p, default_options(max_linesearch_iters=max_ls))
    report(f"max_linesearch_iters={max_ls}", 52000, og, oc)
for shard in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    p = fx.synthetic_mpc_batch(8192, first_id=shard * 8192)
    og, oc = run(p, default_options())
    report(f"shard {shard}", shard * 8192, og, oc)
