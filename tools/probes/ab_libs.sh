#!/bin/bash
# A/B two builds of libegopose_hip.so inside ONE gpurun call (boxes differ by +-5 %): usage ab_libs.sh <libB.so> <python command...>
# runs the command alternately with the in-tree library (A) and with <libB.so> in its place (B), three rounds.
LIBA=egopose_amd/libegopose_hip.so
cp $LIBA /tmp/libA.so; cp $1 /tmp/libB.so; shift
for r in 1 2 3; do
  for v in A B; do
    cp /tmp/lib$v.so $LIBA
    echo "== round $r variant $v"; "$@" 2>&1 | tail -${AB_TAIL:-6}
  done
done
cp /tmp/libA.so $LIBA
