#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python tools/prof_policy.py 2>&1 | grep -v amdgpu.ids | head -45
timeout 300 python tools/prof_collector.py 2>&1 | grep -v amdgpu.ids | head -50
