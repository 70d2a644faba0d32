#!/bin/bash
# same-box A/B of the pipeline shape: depth (buffer sets) x record streams
for rep in 1 2; do for d in 2 3; do for s in 1 2; do
  python tools/sweep.py H2R_TAG depth$d-streams$s --steps 60 --warmup 6 --pipeline-depth $d --side-streams $s
done; done; done
