#!/bin/bash
for L in "" tools/exp/libpfhip_exp4.so ""; do
  echo "== ${L:-shipped}"
  if [ -n "$L" ]; then export PF_LIBPFHIP=$PWD/$L; else unset PF_LIBPFHIP; fi
  python tools/time_s4_layer.py 256 512 16 91:28:2:0 63:10:1:0 73:18:2:0 18:10:1:0 2>&1 | grep -v amdgpu.ids
  python tools/time_s4_layer.py 128 256 16 163:46:3:0 135:28:2:0 2>&1 | grep -v amdgpu.ids
done
