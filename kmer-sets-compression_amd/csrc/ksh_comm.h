// Internal face of the multi-GPU transport (ksh_comm.hip) for the owner-sharded build.
#ifndef KSH_COMM_H_
#define KSH_COMM_H_

#include <hip/hip_runtime.h>

#include <cstddef>

#include "kmersets_hip.h"

namespace ksh {

int comm_rank(const ksh_comm* c);
int comm_world(const ksh_comm* c);
int comm_allgather(ksh_comm* c, const void* d_send, void* d_recv, size_t bytes);
int comm_send(ksh_comm* c, const void* d_buf, size_t bytes, int peer);
int comm_recv(ksh_comm* c, void* d_buf, size_t bytes, int peer);
// the side channel (ksh_comm.hip): operations that run under the context's compute
int comm_side_allgather(ksh_comm* c, const void* d_send, void* d_recv, size_t bytes);
int comm_side_send(ksh_comm* c, const void* d_buf, size_t bytes, int peer);
int comm_side_recv(ksh_comm* c, void* d_buf, size_t bytes, int peer);
hipStream_t comm_side_stream(ksh_comm* c);
int comm_side_join_main(ksh_comm* c);
int comm_side_sync(ksh_comm* c);
void comm_abort(ksh_comm* c);

}  // namespace ksh

#endif
