#!/bin/bash
# scratch: resources of the streaming kernels from the small TU; extra flags pass through ($@); asm in /tmp/stream_only.s
cd /root/repo/shapemol_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -Wno-comment -Wno-pass-failed -fno-slp-vectorize --cuda-device-only -S -o /tmp/stream_only.s probe_stream_only.hip "$@" -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|ScratchSize|VGPRs Spill" | paste - - - - | sed -E 's/[^ ]*sm_edge_stream.h:[0-9]+:[0-9]+: remark: //g; s/\[-Rpass-analysis=kernel-resource-usage\]//g; s/_Z18edge_stream_kernelILi//; s/EEv14EdgeStreamArgs//' | cut -c1-200
