#!/bin/bash
# A / B / C / ...: the in-tree library against any number of other builds inside one gpurun call (boxes differ by +-5 %):
#   abn.sh libB.so [libC.so ...] -- <command...>      three rounds, the variants in turn; prints the last AB_TAIL lines of each run
LIBA=egopose_amd/libegopose_hip.so
cp $LIBA /tmp/lib_0.so
n=0
while [ "$1" != "--" ]; do n=$((n + 1)); cp $1 /tmp/lib_$n.so; names[$n]=$1; shift; done
shift
names[0]=in-tree
for r in 1 2 3; do
  for v in $(seq 0 $n); do
    cp /tmp/lib_$v.so $LIBA
    echo "== round $r variant $v (${names[$v]})"; "$@" 2>&1 | tail -${AB_TAIL:-1}
  done
done
cp /tmp/lib_0.so $LIBA
