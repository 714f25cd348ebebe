#!/bin/bash
# An A/B twin of libuchirp.so that differs in ONE kernel file compiled with extra defines:
#   bash tools/ab_build.sh <name> <kernel file stem> "<-D...>"   ->  ultrasonic-communication_amd/libuchirp_ab_<name>.so
# e.g. bash tools/ab_build.sh nocarry uc_stream_kernel "-DUC_STREAM_CARRY=0"; then on the GPU box
#      bash tools/lib_ab.sh "libuchirp_ab_nocarry.so libuchirp.so" "stream" 3
# (run `make -C ultrasonic-communication_amd libuchirp.so` first: every other object is taken from that build)
set -e
cd "$(dirname "$0")/../ultrasonic-communication_amd"
name="$1"; stem="$2"; defs="$3"
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Xclang -target-feature -Xclang -load-store-opt -mllvm -amdgpu-atomic-optimizer-strategy=None"
/opt/rocm/bin/hipcc $FL $defs -c csrc/$stem.hip -o /tmp/ab_$name.o 2>/dev/null
/opt/rocm/bin/hipcc $FL $defs -DUC_CLOCKSTAMP -c csrc/$stem.hip -o /tmp/ab_$name.clk.o 2>/dev/null
objs=""
for o in csrc/*.o; do
  case "$o" in
    csrc/$stem.o) objs="$objs /tmp/ab_$name.o" ;;
    csrc/$stem.clk.o) objs="$objs /tmp/ab_$name.clk.o" ;;
    *) objs="$objs $o" ;;
  esac
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libuchirp_ab_$name.so $objs -ldl
ls -la libuchirp_ab_$name.so
