#!/bin/bash
# A/B two builds of the core inside one GPU session: LSR_LIB_NAME selects the .so.
for rep in 1 2 3; do
  for v in A B; do
    echo -n "variant $v rep $rep: "; LSR_LIB_NAME=liblsr_$v.so python tools/batch_probe.py 2>&1 | grep "B=1 "
  done
done
