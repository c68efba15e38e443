#!/bin/bash
# A / B / C: the in-tree library against two other builds inside one gpurun call: ab3.sh <libB.so> <libC.so> <command...>
LIBA=egopose_amd/libegopose_hip.so
cp $LIBA /tmp/libA.so; cp $1 /tmp/libB.so; cp $2 /tmp/libC.so; shift; shift
for r in 1 2 3; do
  for v in A B C; do
    cp /tmp/lib$v.so $LIBA
    echo "== round $r variant $v"; "$@" 2>&1 | tail -${AB_TAIL:-1}
  done
done
cp /tmp/libA.so $LIBA
