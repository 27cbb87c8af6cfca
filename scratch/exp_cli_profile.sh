#!/bin/bash
# cProfile of the CLI on the example data (rebuilt from the golden arrays): bash scratch/exp_cli_profile.sh <iterations> <chains>
tmp=$(mktemp -d)
python3 - "$tmp" <<'PY'
import sys, os, numpy as np
G = np.load("tests/golden/binning_lik.npz")
ts, te = G["example_TBP/ts"], G["example_TBP/te"] - 0.5
with open(os.path.join(sys.argv[1], "example.tsv"), "w") as f:
    f.write("id\tts\tte\n")
    for i, (a, b) in enumerate(zip(ts, te)):
        f.write("%d\t%g\t%g\n" % (i, 24.0 - a, 24.0 - b))
PY
python3 -m cProfile -s cumtime LiteRateForward.py -d $tmp/example.tsv -TBP -n $1 -s 1000 -p 1000000 -seed 31 --chains $2 2>&1 | grep -v "^\s*sp\.\|^\s*ex\.\|^[0-9]* -" | grep -A28 "function calls"
rm -rf $tmp
