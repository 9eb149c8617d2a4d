#!/bin/bash
# usage: tools/build_commit.sh <name> <commit>   -> ab/<name>.so : the library as of <commit> (same-box A/B baseline via SN_LIB; same ABI version required)
set -e
name=$1; commit=$2
root=$(cd "$(dirname "$0")/.." && pwd)
d=$root/ab/_src_$name; rm -rf $d; mkdir -p $d/sanerf-hq_amd/csrc $d/include
cd $root
for f in $(git ls-tree --name-only $commit sanerf-hq_amd/csrc/); do git show $commit:$f > $d/$f; done
git show $commit:include/sanerf_hip.h > $d/include/sanerf_hip.h
cd $d/sanerf-hq_amd/csrc && make -j4 > /dev/null 2>&1
cp $d/sanerf-hq_amd/libsanerf_hip.so $root/ab/$name.so; rm -rf $d
echo built ab/$name.so from $commit
