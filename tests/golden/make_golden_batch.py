#!/usr/bin/env python3
"""Golden record for `acmtool -B` (tests/test_cli.py::test_cli_batch_mode).

Batch mode has no reference counterpart, but the files it leaves behind must be the ones the REFERENCE tool
(oracle/_ref/acmtool_ref, markokr/libacm v1.3 compiled by `make -C oracle ref`) writes when it decodes the same
files one by one.  This script runs that serial reference decode in the authoring container and stores the sha256 of
every output file and the stderr text in tests/golden/f8_batch.json.  Inputs are committed fixtures (tests/golden/acm).

  python tests/golden/make_golden_batch.py
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_api as O  # noqa: E402
from helpers import golden_file  # noqa: E402

# (name in the scratch directory, fixture, bytes kept: None = all)
INPUTS = [("b0.acm", "f5_plain", None), ("b1.acm", "f7_src", None), ("b2.acm", "f1_l7_r16_c1", None),
          ("b3.acm", "f1_l9_r3_c2", None), ("b4.acm", "f1_l0_r3_c1", None), ("b5.acm", "f1_l11_r3_c2", None),
          ("b6.acm", "f1_l5_r17_c1", None), ("b_trunc.acm", "f7_src", 400), ("b_junk.acm", None, None)]


def main():
    out = {"reference": "markokr/libacm v1.3 acmtool, one file at a time", "inputs": INPUTS, "runs": []}
    with tempfile.TemporaryDirectory() as td:
        names = []
        for name, src, cut in INPUTS:
            data = b"this is not an acm file at all" if src is None else golden_file(src)
            with open(os.path.join(td, name), "wb") as f:
                f.write(data if cut is None else data[:cut])
            names.append(name)
        for flags, ext in ((["-r"], ".raw"), ([], ".wav"), (["-m"], ".wav"), (["-s", "-r"], ".raw")):
            for n in names:
                p = os.path.join(td, n[:-4] + ext)
                if os.path.exists(p):
                    os.remove(p)
            r = subprocess.run([O.REF_TOOL, "-d", "-q"] + flags + names, cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            files = {}
            for n in names:
                p = os.path.join(td, n[:-4] + ext)
                files[n[:-4] + ext] = hashlib.sha256(open(p, "rb").read()).hexdigest() if os.path.exists(p) else None
            out["runs"].append({"flags": flags, "ext": ext, "rc": r.returncode, "stderr": r.stderr.decode("latin1"),
                                "stdout": r.stdout.decode("latin1"), "files": files})
    with open(os.path.join(HERE, "f8_batch.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote f8_batch.json: %d runs x %d files" % (len(out["runs"]), len(INPUTS)))


if __name__ == "__main__":
    main()
