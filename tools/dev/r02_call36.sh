#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python tools/dev/ragged_cost.py
