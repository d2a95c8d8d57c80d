#!/bin/bash
# round 5: counters, kernel-trace summary and the bench line of the final build (tools/collect_profiles.sh)
set -u
ROOTDIR=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOTDIR
bash tools/collect_profiles.sh r5 2>&1 | tail -5
