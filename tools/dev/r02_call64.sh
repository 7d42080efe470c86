#!/bin/bash
cd $GRAFT_REPO_ROOT
for args in "200 0 256 2 290" "200 0 256 2 600" "200 0 384 2 290" "150 1 256 2 290" "100 0 512 2 100" "200 0 128 2 512" "150 1 384 2 256"; do
  timeout 600 python tools/stress.py $args 2>&1 | grep -v "^iteration" | tail -3 | cut -c1-300
done
