#!/bin/bash
# ON THE GPU BOX: bash tools/ab_run.sh "<python script + args>" tagA tagB ...   (libraries from tools/ab_build.sh; two rounds each)
cmd=$1; shift
for rep in 1 2; do
  for t in "$@"; do
    echo "== $t"
    OCN_LIB=$PWD/tools/_bin/lib_$t.so timeout -k 10 200 python $cmd 2>&1 | tail -${AB_TAIL:-3} || exit 1
  done
done
