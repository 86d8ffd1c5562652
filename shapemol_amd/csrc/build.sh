#!/bin/bash
# Build libshapemol_hip.so for gfx950 (MI355X).  Usage: build.sh [--report]
set -euo pipefail
cd "$(dirname "$0")"
OUT=../libshapemol_hip.so
# -fno-slp-vectorize: packed fp32 VALU (v_pk_add/mul/fma_f32) beside MFMAs is slower than the scalar forms (+1 % on the step)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-value -Wno-comment -Wno-pass-failed -fno-slp-vectorize"
if [[ "${1:-}" == "--report" ]]; then
  hipcc $FLAGS -o $OUT shapemol_hip.hip shape_encoder.hip train_ops.hip -Rpass-analysis=kernel-resource-usage 2>&1 \
   | grep -E "Function Name|VGPRs:|AGPRs|Scratch|Occupancy|LDS Size|VGPRs Spill" | paste - - - - - - - \
   | sed -E 's/.*Function Name: ([^ ]+).*VGPRs: ([0-9]+).*AGPRs: ([0-9]+).*ScratchSize \[bytes\/lane\]: ([0-9]+).*Occupancy \[waves\/SIMD\]: ([0-9]+).*VGPRs Spill: ([0-9]+).*LDS Size \[bytes\/block\]: ([0-9]+).*/\1 vgpr=\2 agpr=\3 scratch=\4 occ=\5 spill=\6 lds=\7/' \
   | c++filt | cut -c1-160
elif [[ "${1:-}" == "--ablate" ]]; then     # diagnostic: drop phases of the edge kernel at compile time (mask in $2)
  hipcc $FLAGS -DSM_ABLATE=$2 -o ../libshapemol_hip_abl$2.so shapemol_hip.hip shape_encoder.hip train_ops.hip
elif [[ "${1:-}" == "--variant" ]]; then    # experiment builds (tools/ only): build.sh --variant NAME -DFLAG=1 ... -> ../libshapemol_hip_NAME.so (SHAPEMOL_LIB=NAME)
  name=$2; shift 2
  hipcc $FLAGS "$@" -o ../libshapemol_hip_${name}.so shapemol_hip.hip shape_encoder.hip train_ops.hip
elif [[ "${1:-}" == "--stamps-serial" ]]; then
  hipcc $FLAGS -DSM_STAMPS -DSM_STAMPS_SERIAL -o ../libshapemol_hip_stamps.so shapemol_hip.hip shape_encoder.hip train_ops.hip
elif [[ "${1:-}" == "--stamps" ]]; then     # diagnostic build with in-kernel phase stamps (tools/ only)
  hipcc $FLAGS -DSM_STAMPS -o ../libshapemol_hip_stamps.so shapemol_hip.hip shape_encoder.hip train_ops.hip
else
  hipcc $FLAGS -o $OUT shapemol_hip.hip shape_encoder.hip train_ops.hip
fi
