#!/bin/bash
# Knock-out builds of stream_kernel (UC_STREAM_KNOCK bits: 1 no input loads in the loop, 2 no FIR arithmetic, 4 no
# transforms, 8 no output stores): what each stage costs beside the others.  Results are WRONG by construction; the
# libraries are never shipped.
# build (container): bash tools/stream_knock.sh build      -> ultrasonic-communication_amd/libuchirp_sk<bits>.so
# run (GPU box):     bash tools/stream_knock.sh run        -> samples/s of every build, two alternations
set -e
cd "$(dirname "$0")/../ultrasonic-communication_amd"
KS="${KS:-1 2 4 8 6 7 15 16 32}"
if [ "$1" = build ]; then
  FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Xclang -target-feature -Xclang -load-store-opt -mllvm -amdgpu-atomic-optimizer-strategy=None"
  for k in $KS; do
    /opt/rocm/bin/hipcc $FL -DUC_STREAM_KNOCK=$k -c csrc/uc_stream_kernel.hip -o /tmp/uc_stream_k$k.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libuchirp_sk$k.so csrc/uc_band_kernel.o csrc/uc_full_kernel.o \
      csrc/uc_iq_kernel.o /tmp/uc_stream_k$k.o csrc/uc_cic_kernel.o csrc/uc_api_core.o csrc/uc_api_rx.o csrc/uc_api_stream.o csrc/uc_api_cic.o csrc/uc_api_clock.o csrc/uc_tables.o
  done
  ls -la libuchirp_sk*.so
else
  cd ..
  for rep in 1 2; do
    for L in libuchirp.so $(for k in $KS; do echo libuchirp_sk$k.so; done); do
      UCHIRP_LIB=$PWD/ultrasonic-communication_amd/$L python3 bench.py --variant stream 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-22s %.4g samples/s  %.4f ms' % ('$L', d['value'], d.get('ms_per_step', 0)))"
    done
  done
fi
