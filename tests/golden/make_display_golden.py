#!/usr/bin/env python3
"""Generates tests/golden/reference_display.json: the text the REFERENCE's own
display functions (PrintIterHeader/IterLine/DetailedHeader/DetailedLine/
DetailedFooter/PrintFinal, fbstab_algorithm-impl.h:411-541) print at
Display::FINAL, ITER and ITER_DETAILED for the known-answer problems of
reference_kats.json.  The text is produced by oracle/_ref/libfbstab_ref.so,
i.e. by the reference's FBstabAlgorithm<> template compiled from
/root/reference where it lies and writing into an OutputStream subclass that
collects the messages (oracle/oracle_capi.cc: CaptureOutput), so it only runs
where /root/reference exists.  The fixture holds outputs only: the problems
are named, their data lives in reference_kats.json.

The wall-clock figure of the "Time elapsed" line is replaced by <t>.

Usage: python tests/golden/make_display_golden.py
"""
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle.oracle_py import Oracle, default_options  # noqa: E402
from tests import helpers as H  # noqa: E402

CASES = [("dense", 0), ("dense", 1), ("dense", 3), ("dense", 4), ("mpc", 0), ("mpc", 2)]


def main():
    kats = json.load(open(os.path.join(HERE, "reference_kats.json")))
    ref = Oracle(True)
    out = {"_comment": "Display text printed by the reference's own FBstabAlgorithm<> "
                       "(see make_display_golden.py); level 1 = Display::FINAL (the default), 2 = ITER, 3 = ITER_DETAILED.",
           "cases": []}
    for kind, idx in CASES:
        k = kats[kind + "_end_to_end"][idx]
        p = H.dense_from_kat(k) if kind == "dense" else H.mpc_from_kat(k)
        for level in (1, 2, 3):
            r = ref.solve_display(p, opts=default_options(display_level=level))
            text = re.sub(r"Time elapsed: \S+ ms", "Time elapsed: <t> ms", r[5])
            out["cases"].append(dict(kind=kind, index=idx, name=k["name"], N=k.get("N"),
                                     level=level, eflag=int(r[4]["eflag"][0]),
                                     newton_iters=int(r[4]["newton_iters"][0]),
                                     prox_iters=int(r[4]["prox_iters"][0]), text=text))
    with open(os.path.join(HERE, "reference_display.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", len(out["cases"]), "cases")


if __name__ == "__main__":
    main()
