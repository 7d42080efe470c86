#!/bin/bash
# Side-by-side run on REAL models and reads: needs a reference `flappie` binary built elsewhere
# (OpenBLAS + HDF5) and the real src/models/*.mdl files (git-LFS objects, not in this checkout).
#   tools/parity_real.sh /path/to/reference/flappie /path/to/reference/src/models reads/ [model]
# Compares called bases and qualities record by record; exits non-zero on the first difference.
set -euo pipefail
ref=$1; models=$2; reads=$3; model=${4:-r941_native}
here=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
"$ref" --model "$model" "$reads" | sort > "$tmp/ref.fq.sorted" &
FLAPPIE_MODEL_DIR=$models "$here/flappie_amd/flappie" --model "$model" "$reads" | sort > "$tmp/new.fq.sorted"
wait
python3 - "$tmp/ref.fq.sorted" "$tmp/new.fq.sorted" <<'PY'
import sys
def recs(path):
    lines = [l.rstrip("\n") for l in open(path)]
    heads = [l for l in lines if l.startswith("@") and "{" in l]
    return {h.split()[0]: h for h in heads}, lines
a, la = recs(sys.argv[1]); b, lb = recs(sys.argv[2])
print("records: reference %d, this build %d" % (len(a), len(b)))
sys.exit(0 if sorted(la) == sorted(lb) else 1)
PY
