// fake_rccl.cpp -- TEST INFRASTRUCTURE: a host stand-in for the six RCCL entry points devices.cpp calls, so that the
// multi-device driver (capi.cpp::estimate_maps_devices, devices.cpp::gather_pair_records, compiled for real) runs under
// ThreadSanitizer and AddressSanitizer without a GPU.  The single-process form only: every rank's ncclAllGather is enqueued
// inside one ncclGroupStart / ncclGroupEnd pair on one thread and executed at the group's end ("device" memory is host
// memory, fake_hip.cpp).  Like RCCL, ncclCommInitAll refuses a device twice.  Nothing here is part of the product.
#include <rccl/rccl.h>

#include <cstring>
#include <set>
#include <vector>

struct FakeComm { int rank, world; };
struct FakeOp { const void *send; void *recv; size_t bytes; FakeComm *comm; };
static thread_local std::vector<FakeOp> t_ops;
static thread_local int t_depth = 0;

static ncclResult_t run_ops()
{
  for (const FakeOp &a : t_ops)
    for (const FakeOp &b : t_ops) {
      if (a.bytes != b.bytes || a.comm->world != b.comm->world) return ncclInvalidArgument;
      std::memcpy((char *)b.recv + (size_t)a.comm->rank * a.bytes, a.send, a.bytes);
    }
  if (!t_ops.empty() && (int)t_ops.size() != t_ops[0].comm->world) return ncclInvalidUsage;   // a rank did not call
  t_ops.clear();
  return ncclSuccess;
}

extern "C" {
ncclResult_t ncclCommInitAll(ncclComm_t *comms, int n, const int *devs)
{
  if (!comms || n < 1 || !devs) return ncclInvalidArgument;
  if ((int)std::set<int>(devs, devs + n).size() != n) return ncclInvalidUsage;
  for (int i = 0; i < n; ++i) comms[i] = (ncclComm_t) new FakeComm{i, n};
  return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) { delete (FakeComm *)c; return ncclSuccess; }
ncclResult_t ncclGroupStart() { ++t_depth; return ncclSuccess; }
ncclResult_t ncclGroupEnd() { return --t_depth == 0 ? run_ops() : ncclSuccess; }
ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t type, ncclComm_t comm, hipStream_t)
{
  if (type != ncclChar && type != ncclUint8) return ncclInvalidArgument;
  t_ops.push_back(FakeOp{send, recv, count, (FakeComm *)comm});
  return t_depth ? ncclSuccess : run_ops();
}
const char *ncclGetErrorString(ncclResult_t) { return "fake RCCL"; }
}
