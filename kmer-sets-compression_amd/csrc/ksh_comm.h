// Internal face of the multi-GPU transport (ksh_comm.hip) for the owner-sharded build.
#ifndef KSH_COMM_H_
#define KSH_COMM_H_

#include <cstddef>

#include "kmersets_hip.h"

namespace ksh {

int comm_rank(const ksh_comm* c);
int comm_world(const ksh_comm* c);
int comm_allgather(ksh_comm* c, const void* d_send, void* d_recv, size_t bytes);
int comm_send(ksh_comm* c, const void* d_buf, size_t bytes, int peer);
int comm_recv(ksh_comm* c, void* d_buf, size_t bytes, int peer);

}  // namespace ksh

#endif
