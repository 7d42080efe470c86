#!/bin/bash
# usage: tools/dev/build.sh [EXTRA flags for the timing harness]
set -e
cd /root/repo/flappie_amd/csrc
make -j4 2>&1 | grep -E " error |Error " -A8 || true
make -B ../../tools/bin/persist_timing EXTRA="$*" 2>&1 | grep -E " error |Error " -A8 || true
ls -la --time-style=full-iso ../libffhip.so ../../tools/bin/persist_timing | awk '{print $6, $7, $9}'
