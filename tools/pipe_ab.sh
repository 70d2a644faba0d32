#!/bin/bash
# same-box A/B of the pipeline shape: depth (buffer sets) x side streams
for rep in 1 2; do for d in 2 3; do for one in 0 1; do
  export H2R_PIPE_DEPTH=$d; unset H2R_PIPE_ONE_AUX; [ $one = 1 ] && export H2R_PIPE_ONE_AUX=1
  python tools/sweep.py H2R_TAG depth$d-oneaux$one --steps 60 --warmup 6
done; done; done
