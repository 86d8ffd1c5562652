// scratch TU: only the B = 256 streaming kernels, for quick resource / asm checks (not part of the library)
#include "../../include/shapemol_hip.h"
#include "sm_device.h"
#include "sm_edge.h"
#include "sm_edge_bf16.h"
#include "sm_edge16.h"
#include "sm_edge_stream.h"
template __global__ void edge_stream_kernel<128, 8, false>(EdgeStreamArgs);
template __global__ void edge_stream_kernel<128, 8, true>(EdgeStreamArgs);
template __global__ void edge_stream_kernel<128, 32, false>(EdgeStreamArgs);
