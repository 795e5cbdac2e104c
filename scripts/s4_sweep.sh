#!/bin/bash
# round 2: tile-streamer sampler (variant 4) against the walker, with ablations
cd ${GRAFT_REPO_ROOT:-.}
V="--opt sample.variant=4"
scripts/sweep.sh s4 \
  "--opt sample.variant=1" \
  "$V --opt sample.srows=32 --opt sample.depth=2 --opt sample.spread=0" \
  "$V --opt sample.srows=32 --opt sample.depth=2" \
  "$V --opt sample.srows=32 --opt sample.depth=4" \
  "$V --opt sample.srows=16 --opt sample.depth=2" \
  "$V --opt sample.srows=64 --opt sample.depth=2" \
  "$V --opt sample.srows=64 --opt sample.depth=4" \
  "$V --opt sample.srows=32 --opt sample.depth=2 --opt sample.hsplit=2" \
  "$V --opt sample.srows=64 --opt sample.depth=2 --opt sample.hsplit=4" \
  "$V --opt sample.srows=32 --opt sample.depth=2 --opt debug.ablate=16" \
  "$V --opt sample.srows=32 --opt sample.depth=2 --opt debug.ablate=32" \
  "$V --opt sample.srows=32 --opt sample.depth=2 --opt debug.ablate=64" \
  "$V --opt sample.srows=32 --opt sample.depth=2 --streams 3"
