#!/bin/bash
# Round 5: the kernel traces again (without the two-stream launches in the process), then the bench lines with the committed counters
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
KT_ONLY=1 NO_CONDENSE=1 bash tools/make_profiles.sh
