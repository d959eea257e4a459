// One translation unit for the interaction-graph, pair-pass and side-chain kernels: the merged launches of kernels_batch.h
// (several independent kernels of a force pass as workgroup ranges of ONE launch) need all their bodies in one code object.
// (kernels_basic.hip stays on its own: the per-element kernels are compiled without fused-multiply-add contraction.)
#include "igraph_device.h"
#include "pair2_device.h"
#include <cstdio>
#include <cstring>
using namespace up;
#define ST(L) ((hipStream_t)(L)->stream)
#define UPK_LAUNCH_STATUS_DEFINED
static inline int launch_status() { return (int)hipGetLastError(); }
#include "kernels_batch.h"
// launchers without a batch form run an open batch, then the queue of fused per-element ops, before they launch
#undef UPK_FLUSH
#define UPK_FLUSH(L) do { UPK_BATCH_BREAK(L); const int r_ = upk_fuse_flush(L); if (r_) return r_; } while (0)
#include "kernels_igraph.hip"
#include "kernels_pair.hip"
#include "kernels_rotamer.hip"
#include "kernels_batch_impl.h"
