#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu -rP 2>&1 | grep -E "packed-fp32 probe|passed|failed|Error" | tail -5
