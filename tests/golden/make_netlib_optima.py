#!/usr/bin/env python3
"""tests/golden/netlib_optima.json: the Netlib optimum table the reference carries (main.py:1317-1618, the same numbers as
benchmarks/readme.txt:83-182), as data.  Run in the build container only (imports the reference):

    python3 tests/golden/make_netlib_optima.py
"""
import contextlib, io, json, os, sys
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
os.chdir(REF)
sys.path.insert(0, REF)
import main as ref_main  # noqa: E402
with contextlib.redirect_stdout(io.StringIO()):
    names, optima = ref_main.benchmark()
table = {}
for n, v in zip(names, optima):
    try:
        table[n.replace(".", "-")] = float(v)
    except (TypeError, ValueError):
        pass                                   # "(see NOTES)" entries
with open(os.path.join(HERE, "netlib_optima.json"), "w") as fh:
    json.dump(table, fh, indent=0, sort_keys=True)
print(len(table), "optima")
