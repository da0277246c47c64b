#!/bin/bash
# Variants of the look-ahead plan (which target-only piece runs where), A = plain plan, B = variant.
R=${GRAFT_REPO_ROOT:-$(pwd)}
for v in "" "RIED_LA_VGG_FRONT=1" "RIED_LA_DREAL_FRONT=1" "RIED_NO_PRE_DREAL=1" "RIED_LA_VGG_FRONT=1 RIED_NO_PRE_DREAL=1"; do
  echo "== $v"
  env $v timeout -k 10 200 python -u $R/tools/ab_step.py --ab-lookahead 2>&1 | grep "^[AB] [0-9]" || exit 1
done
