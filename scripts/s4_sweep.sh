#!/bin/bash
# round 2: the tile streamer (sample.variant=2) against the walker, with ablations
# (debug.ablate 16: no stores, 32: no arithmetic and no stores)
cd ${GRAFT_REPO_ROOT:-.}
V="--opt sample.variant=2"
scripts/sweep.sh s4 \
  "--opt sample.variant=1" \
  "$V" \
  "$V --opt sample.srows=16" \
  "$V --opt sample.srows=12" \
  "$V --opt sample.depth=3" \
  "$V --opt sample.spread=1" \
  "$V --opt sample.groups=1" \
  "$V --opt debug.ablate=16" \
  "$V --opt debug.ablate=32" \
  "$V --streams 3"
