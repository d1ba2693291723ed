#!/usr/bin/env python3
"""Writes the input files of the reference-golden recipe (ral/test.cpp text format) into
tools/ref_golden/inputs/ and a manifest (inputs/cases.tsv: name, input file, CLI arguments)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from cases import CASES, ROOT, build_case  # noqa: E402

sys.path.insert(0, ROOT)
from irotavg_amd import graphio  # noqa: E402


def main():
    out = os.path.join(HERE, "inputs")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "cases.tsv"), "w") as man:
        for name, c in CASES.items():
            g = build_case(name)
            path = os.path.join(out, name + ".txt")
            graphio.write_ravg_input(path, g["I"], g["QQ"], g["Q"][:max(g["n_abs_read"], g["f"])], g["n"], g["f"])
            man.write("%s\t%s\t%s\n" % (name, os.path.basename(path), " ".join(c["args"])))
            print("wrote", path)


if __name__ == "__main__":
    main()
