#!/bin/bash
for i in 1 2 3 4 5 6; do
  echo "== run $i: $(python tools/dbg_pair.py 2>&1 | grep failures)"
done
