// scratch TU: only the exact-mode node kernels, for quick resource / asm checks (not part of the library)
#include "../../include/shapemol_hip.h"
#include "sm_device.h"
#include "sm_edge.h"
#include "sm_edge_bf16.h"
#include "sm_edge16.h"
#include "sm_node.h"
template __global__ void node_chain6_kernel<128>(NodeChainArgs);
template __global__ void node_linear6_kernel<128>(NodeLinArgs);
template __global__ void node_prologue6_kernel<128>(NodePrologueArgs);
