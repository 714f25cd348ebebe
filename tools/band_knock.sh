#!/bin/bash
# Knock-out builds of band_kernel<rx_real> (UC_BAND_KNOCK bits: 2 no pass-1 arithmetic, 4 no pass-2 arithmetic, 8 no
# pruned-pass arithmetic, 16 no exchange 1, 32 no exchange-2 stores, 64 no pruned-pass reads, 128 no window search; the
# no-load build is -DUC_KNOCK_NOLOAD = "L"): what each stage costs beside the others.  Results are WRONG by
# construction; the libraries are never shipped.
# build (container): KS="..." bash tools/band_knock.sh build   -> ultrasonic-communication_amd/libuchirp_bk<bits>.so
# run (GPU box):     KS="..." bash tools/band_knock.sh run
set -e
cd "$(dirname "$0")/../ultrasonic-communication_amd"
KS="${KS:-2 4 8 14 16 32 64 112 128 142 254 L}"
if [ "$1" = build ]; then
  FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Xclang -target-feature -Xclang -load-store-opt -mllvm -amdgpu-atomic-optimizer-strategy=None"
  for k in $KS; do
    if [ "$k" = L ]; then def="-DUC_KNOCK_NOLOAD"; else def="-DUC_BAND_KNOCK=$k"; fi
    /opt/rocm/bin/hipcc $FL $def -c csrc/uc_band_kernel.hip -o /tmp/uc_band_k$k.o 2>/dev/null
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libuchirp_bk$k.so /tmp/uc_band_k$k.o csrc/uc_full_kernel.o \
      csrc/uc_iq_kernel.o csrc/uc_stream_kernel.o csrc/uc_cic_kernel.o csrc/uc_api_core.o csrc/uc_api_rx.o csrc/uc_api_stream.o csrc/uc_api_cic.o csrc/uc_api_clock.o csrc/uc_tables.o
  done
  ls libuchirp_bk*.so
else
  cd ..
  bash tools/lib_ab.sh "libuchirp.so $(for k in $KS; do echo -n "libuchirp_bk$k.so "; done)" rx_real "${REPS:-2}"
fi
