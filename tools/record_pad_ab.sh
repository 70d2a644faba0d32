#!/bin/bash
# same box: the record kernel for record strides 65,024 + 256 * pad (library variants -DH2R_RECORD_PAD_UNITS=pad; 2 = shipped)
for rep in 1 2; do
for k in 2 0 1 3 5 7; do
  lib=""; [ $k != 2 ] && lib=$PWD/halo2_rsa_amd/lib/variants/pad$k.so
  H2R_LIB=$lib timeout 200 python tools/record_pad_probe.py 2>&1 | grep "record stride"
done; done
