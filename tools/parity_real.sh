#!/bin/bash
# Side-by-side run on REAL models and reads: needs a reference `flappie` binary built elsewhere
# (OpenBLAS + HDF5) and the real src/models/*.mdl files (git-LFS objects, not in this checkout).
#   tools/parity_real.sh /path/to/reference/flappie /path/to/reference/src/models reads/ [model]
# Compares called bases and qualities read by read (north star: bit-exact base string); prints the mismatching reads.
set -euo pipefail
ref=$1; models=$2; reads=$3; model=${4:-r941_native}
here=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
"$ref" --model "$model" "$reads" > "$tmp/ref.fq" &
FLAPPIE_MODEL_DIR=$models "$here/flappie_amd/flappie" --model "$model" "$reads" > "$tmp/new.fq"
wait
python3 - "$tmp/ref.fq" "$tmp/new.fq" <<'PY'
import sys
def recs(path):
    lines = [l.rstrip("\n") for l in open(path)]
    return {lines[k][1:].split()[0]: (lines[k + 1], lines[k + 3]) for k in range(0, len(lines) - 3, 4) if lines[k].startswith("@")}
a, b = recs(sys.argv[1]), recs(sys.argv[2])
bad = [n for n in a if a[n] != b.get(n)]
print("reads: reference %d, this build %d, differing %d" % (len(a), len(b), len(bad)))
for n in bad[:20]:
    print("  ", n, "bases differ" if n in b and a[n][0] != b[n][0] else ("qualities differ" if n in b else "missing"))
sys.exit(1 if bad or len(a) != len(b) else 0)
PY
