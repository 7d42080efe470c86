#!/bin/bash
# INTEGRATION.md section 1, compiled: the REFERENCE's own flappie.c (and runnie.c) built unchanged from where they lie under
# $REF/src, against this repo's include/ for every header the drop-in layer replaces, and linked against
# -lflappie_host -lffhip instead of layers.c / networks.c / decode.c / flappie_matrix.c / nnfeatures.c / flappie_common.c /
# flappie_structures.c / OpenBLAS.  Nothing is copied; outputs go to a scratch directory (default /tmp/flappie_relink).
# Runs in the build container only (the GPU box has no /root/reference); what it proves is that the symbol set and the
# declarations of include/ are what flappie.c needs -- `nm -u` of the result must show no flappie symbol left undefined.
#
#   tools/relink_check.sh [REF=/root/reference] [OUT=/tmp/flappie_relink]
#
# Headers: -Iinclude comes first, so decode.h / layers.h / networks.h / flappie_matrix.h / flappie_structures.h /
# flappie_common.h / flappie_output.h / fast5_interface.h are OURS; flappie_licence.h, flappie_stdlib.h and util.h (inline
# helpers only) come from $REF/src.  version.h is produced the way the reference's CMakeLists.txt:24-26,50-53 does it
# (configure_file of src/version.h.in with the CPACK version numbers of that file) -- a three-number substitution, done by sed.
set -euo pipefail
REF=${1:-/root/reference}
OUT=${2:-/tmp/flappie_relink}
here=$(cd "$(dirname "$0")/.." && pwd)
[ -f "$REF/src/flappie.c" ] || { echo "reference sources not present at $REF: nothing to check"; exit 0; }
HDF5=${HDF5_PREFIX:-$(for p in /usr /usr/local /opt/conda; do [ -f $p/include/hdf5.h ] && echo $p && break; done)}
[ -n "$HDF5" ] || { echo "no hdf5.h in this image: flappie.c cannot be compiled (fast5_interface.h needs it)"; exit 0; }
mkdir -p "$OUT/gen"
ver() { grep -E "set \(CPACK_PACKAGE_VERSION_$1 " "$REF/CMakeLists.txt" | grep -oE '[0-9]+' | head -1; }
sed -e "s/@CPACK_PACKAGE_VERSION_MAJOR@/$(ver MAJOR)/g" -e "s/@CPACK_PACKAGE_VERSION_MINOR@/$(ver MINOR)/g" \
    -e "s/@CPACK_PACKAGE_VERSION_PATCH@/$(ver PATCH)/g" -e "s/@GIT_COMMIT_HASH@/relink/g" "$REF/src/version.h.in" > "$OUT/gen/version.h"
CFLAGS="-O2 -std=c99 -fgnu89-inline -DUSE_SSE2 -DNDEBUG -D_GNU_SOURCE -w -I$here/include -I$OUT/gen -I$REF/src -I$HDF5/include"
LIBS="-L$here/flappie_amd -lflappie_host -lffhip $HDF5/lib/libhdf5.so -lm -Wl,--enable-new-dtags -Wl,--allow-shlib-undefined -Wl,-rpath,$here/flappie_amd -Wl,-rpath,$HDF5/lib"
rc=0
nm -D --defined-only "$here/flappie_amd/libflappie_host.so" "$here/flappie_amd/libffhip.so" | awk 'NF >= 3 {print $3}' | sort -u > "$OUT/provided.txt"
for prog in flappie runnie; do
    # the translation units INTEGRATION.md keeps from the reference: the program, its fast5 reader and its record writer
    gcc $CFLAGS "$REF/src/$prog.c" "$REF/src/fast5_interface.c" "$REF/src/flappie_output.c" $LIBS -o "$OUT/${prog}_relinked" || { echo "relink of $prog.c FAILED"; rc=1; continue; }
    left=$(nm -u "$OUT/${prog}_relinked" | grep -vE '@|__gmon_start__|_ITM_|__cxa|H5' | awk '{print $2}' | sort -u | tr '\n' ' ')
    # every remaining undefined symbol must be provided by the two libraries
    miss=""
    for s in $left; do
        grep -qx "$s" "$OUT/provided.txt" || miss="$miss $s"
    done
    if [ -n "$miss" ]; then echo "$prog.c links, but these symbols are in neither library:$miss"; rc=1
    else echo "$prog.c from $REF/src compiles against include/ and links against libflappie_host + libffhip ($(echo $left | wc -w) symbols resolved there): $OUT/${prog}_relinked"; fi
done
exit $rc
