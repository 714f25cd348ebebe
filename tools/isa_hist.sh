#!/bin/bash
# Instruction histogram of one kernel in a built object: bash tools/isa_hist.sh <object.o> <symbol substring>
# (used to check that a change to a sibling path leaves a hot kernel's code alone)
OD=/opt/rocm/lib/llvm/bin/llvm-objdump
tmp=$(mktemp -d)
cp "$1" $tmp/x.o && (cd $tmp && $OD -d --offloading x.o > /dev/null 2>&1)
co=$(ls $tmp/x.o.*amdgcn* | head -1)
$OD -d $co | awk -v pat="$2" '$0 ~ /^[0-9a-f]+ <.*>:$/ {f = ($0 ~ pat)} f && !/>:$/ {print $1}' | sort | uniq -c | sort -rn
rm -rf $tmp
