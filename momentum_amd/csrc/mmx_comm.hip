// mmx_comm.hip -- the multi-GPU exchange of the batched-IK path: one all-reduce of three doubles per solve,
// RCCL called directly (include/mmx.h, "Multi-GPU").  librccl is opened at run time: a single-GPU process
// never loads it, and a process that already carries an RCCL (torch ships one under the same soname) shares
// that copy.
#include "../../include/mmx.h"

#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cstring>
#include <new>
#include <string>
#include <vector>

namespace mmx {
int32_t failWith(int32_t code, const std::string& msg); // mmx_capi.hip: sets mmx_last_error()
}

namespace {

// the handful of RCCL entry points this file needs (rccl.h: NCCL_UNIQUE_ID_BYTES = 128, ncclSum = 0, ncclFloat64 = 8)
struct NcclId {
  char internal[MMX_COMM_ID_BYTES];
};
using ncclComm_t = void*;
struct Rccl {
  int (*getUniqueId)(NcclId*) = nullptr;
  int (*commInitRank)(ncclComm_t*, int, NcclId, int) = nullptr;
  int (*commInitAll)(ncclComm_t*, int, const int*) = nullptr;
  int (*allReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*commDestroy)(ncclComm_t) = nullptr;
  int (*commCount)(ncclComm_t, int*) = nullptr; // optional
  const char* (*getErrorString)(int) = nullptr;
  std::string why;
  Rccl() {
    void* h = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (h != nullptr) {
        break;
      }
    }
    if (h == nullptr) {
      why = "librccl.so not found (multi-GPU runs need RCCL)";
      return;
    }
    auto sym = [&](const char* n) { return dlsym(h, n); };
    getUniqueId = reinterpret_cast<decltype(getUniqueId)>(sym("ncclGetUniqueId"));
    commInitRank = reinterpret_cast<decltype(commInitRank)>(sym("ncclCommInitRank"));
    commInitAll = reinterpret_cast<decltype(commInitAll)>(sym("ncclCommInitAll"));
    allReduce = reinterpret_cast<decltype(allReduce)>(sym("ncclAllReduce"));
    commDestroy = reinterpret_cast<decltype(commDestroy)>(sym("ncclCommDestroy"));
    getErrorString = reinterpret_cast<decltype(getErrorString)>(sym("ncclGetErrorString"));
    commCount = reinterpret_cast<decltype(commCount)>(sym("ncclCommCount"));
    if (!getUniqueId || !commInitRank || !commInitAll || !allReduce || !commDestroy) {
      why = "librccl.so lacks an expected entry point";
      getUniqueId = nullptr;
    }
  }
  bool ok() const {
    return getUniqueId != nullptr;
  }
};
const Rccl& rccl() {
  static const Rccl r;
  return r;
}
int32_t ncclFail(int rc, const char* what) {
  const char* msg = rccl().getErrorString != nullptr ? rccl().getErrorString(rc) : "?";
  return mmx::failWith(MMX_ERR_DEVICE, std::string(what) + ": " + msg);
}
constexpr int kNcclSum = 0, kNcclFloat64 = 8;

// (sum error, sum iterations, #failed) of one shard: one workgroup, fixed summation order (deterministic)
__global__ void __launch_bounds__(256) residualNormsKernel(
    int B,
    const double* __restrict__ err,
    const int32_t* __restrict__ iters,
    const int32_t* __restrict__ status,
    double* __restrict__ out) {
  __shared__ double red[3][256];
  double e = 0.0, it = 0.0, bad = 0.0;
  for (int b = threadIdx.x; b < B; b += 256) {
    e += err != nullptr ? err[b] : 0.0;
    it += iters != nullptr ? double(iters[b]) : 0.0;
    bad += (status != nullptr && (status[b] & MMX_SOLVE_ERROR_MASK) != 0) ? 1.0 : 0.0; // (informational bits do not count)
  }
  red[0][threadIdx.x] = e, red[1][threadIdx.x] = it, red[2][threadIdx.x] = bad;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (int(threadIdx.x) < s) {
      for (int k = 0; k < 3; ++k) {
        red[k][threadIdx.x] += red[k][threadIdx.x + s];
      }
    }
    __syncthreads();
  }
  if (threadIdx.x < 3) {
    out[threadIdx.x] = red[threadIdx.x][0];
  }
}

} // namespace

struct mmx_comm {
  ncclComm_t comm = nullptr;
  int32_t world = 1, rank = 0, device = 0;
  double* staging = nullptr; // [3] device doubles + a stream, created on first use by the host-buffer entry point
  hipStream_t stream = nullptr;
};

extern "C" {

int32_t mmx_comm_unique_id(uint8_t id[MMX_COMM_ID_BYTES]) {
  if (id == nullptr) {
    return mmx::failWith(MMX_ERR_INVALID_ARGUMENT, "id is null");
  }
  if (!rccl().ok()) {
    return mmx::failWith(MMX_ERR_UNSUPPORTED, rccl().why);
  }
  NcclId u;
  const int rc = rccl().getUniqueId(&u);
  if (rc != 0) {
    return ncclFail(rc, "ncclGetUniqueId");
  }
  std::memcpy(id, u.internal, MMX_COMM_ID_BYTES);
  return MMX_OK;
}

int32_t mmx_comm_create(const uint8_t id[MMX_COMM_ID_BYTES], int32_t world_size, int32_t rank, int32_t device, mmx_comm** out) {
  if (out == nullptr || id == nullptr) {
    return mmx::failWith(MMX_ERR_INVALID_ARGUMENT, "out / id is null");
  }
  *out = nullptr;
  if (world_size <= 0 || rank < 0 || rank >= world_size) {
    return mmx::failWith(MMX_ERR_INVALID_ARGUMENT, "rank / world size out of range");
  }
  if (!rccl().ok()) {
    return mmx::failWith(MMX_ERR_UNSUPPORTED, rccl().why);
  }
  if (hipSetDevice(device) != hipSuccess) {
    (void)hipGetLastError();
    return mmx::failWith(MMX_ERR_DEVICE, "hipSetDevice failed");
  }
  mmx_comm* c = new (std::nothrow) mmx_comm();
  if (c == nullptr) {
    return mmx::failWith(MMX_ERR_OUT_OF_MEMORY, "host allocation failed");
  }
  NcclId u;
  std::memcpy(u.internal, id, MMX_COMM_ID_BYTES);
  const int rc = rccl().commInitRank(&c->comm, world_size, u, rank);
  if (rc != 0) {
    delete c;
    return ncclFail(rc, "ncclCommInitRank");
  }
  c->world = world_size, c->rank = rank, c->device = device;
  *out = c;
  return MMX_OK;
}

int32_t mmx_comm_create_all(int32_t num_devices, const int32_t* devices, mmx_comm** out) {
  if (out == nullptr || devices == nullptr || num_devices <= 0) {
    return mmx::failWith(MMX_ERR_INVALID_ARGUMENT, "out / devices is null or no device");
  }
  for (int32_t i = 0; i < num_devices; ++i) {
    out[i] = nullptr;
  }
  if (!rccl().ok()) {
    return mmx::failWith(MMX_ERR_UNSUPPORTED, rccl().why);
  }
  // the handles first: a failed host allocation must not leave RCCL communicators behind
  std::vector<mmx_comm*> objs(static_cast<size_t>(num_devices), nullptr);
  for (int32_t i = 0; i < num_devices; ++i) {
    objs[static_cast<size_t>(i)] = new (std::nothrow) mmx_comm();
    if (objs[static_cast<size_t>(i)] == nullptr) {
      for (mmx_comm* c : objs) {
        delete c;
      }
      return mmx::failWith(MMX_ERR_OUT_OF_MEMORY, "host allocation failed");
    }
  }
  std::vector<ncclComm_t> comms(static_cast<size_t>(num_devices), nullptr);
  std::vector<int> devs(devices, devices + num_devices);
  const int rc = rccl().commInitAll(comms.data(), num_devices, devs.data());
  if (rc != 0) {
    for (mmx_comm* c : objs) {
      delete c;
    }
    return ncclFail(rc, "ncclCommInitAll");
  }
  for (int32_t i = 0; i < num_devices; ++i) {
    mmx_comm* c = objs[static_cast<size_t>(i)];
    c->comm = comms[static_cast<size_t>(i)];
    c->world = num_devices, c->rank = i, c->device = devices[i];
    out[i] = c;
  }
  return MMX_OK;
}

int32_t mmx_comm_world_size(const mmx_comm* comm) {
  if (comm == nullptr) {
    return 0;
  }
  // what RCCL itself counts (ncclCommCount) when the library exports it: bench.py asserts it equals --gpus
  if (comm->comm != nullptr && rccl().ok() && rccl().commCount != nullptr) {
    int n = 0;
    if (rccl().commCount(comm->comm, &n) == 0) {
      return n;
    }
  }
  return comm->world;
}
int32_t mmx_comm_rank(const mmx_comm* comm) {
  return comm != nullptr ? comm->rank : -1;
}

int32_t mmx_comm_all_reduce_norms(mmx_comm* comm, double* norms_dev, void* stream) {
  if (comm == nullptr || norms_dev == nullptr) {
    return mmx::failWith(MMX_ERR_INVALID_ARGUMENT, "comm / norms is null");
  }
  const int rc = rccl().allReduce(norms_dev, norms_dev, 3, kNcclFloat64, kNcclSum, comm->comm, static_cast<hipStream_t>(stream));
  if (rc != 0) {
    return ncclFail(rc, "ncclAllReduce");
  }
  return MMX_OK;
}

int32_t mmx_comm_all_reduce_norms_host(mmx_comm* comm, double norms_host[3]) {
  if (comm == nullptr || norms_host == nullptr) {
    return mmx::failWith(MMX_ERR_INVALID_ARGUMENT, "comm / norms is null");
  }
  auto hipFail = [](hipError_t e, const char* what) { return mmx::failWith(MMX_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e)); };
  hipError_t e = hipSetDevice(comm->device);
  if (e != hipSuccess) {
    return hipFail(e, "hipSetDevice");
  }
  if (comm->staging == nullptr) {
    if ((e = hipMalloc(reinterpret_cast<void**>(&comm->staging), 3 * sizeof(double))) != hipSuccess) {
      return hipFail(e, "hipMalloc");
    }
    if ((e = hipStreamCreateWithFlags(&comm->stream, hipStreamNonBlocking)) != hipSuccess) {
      return hipFail(e, "hipStreamCreate");
    }
  }
  if ((e = hipMemcpyAsync(comm->staging, norms_host, 3 * sizeof(double), hipMemcpyHostToDevice, comm->stream)) != hipSuccess) {
    return hipFail(e, "hipMemcpyAsync");
  }
  const int32_t rc = mmx_comm_all_reduce_norms(comm, comm->staging, comm->stream);
  if (rc != MMX_OK) {
    return rc;
  }
  if ((e = hipMemcpyAsync(norms_host, comm->staging, 3 * sizeof(double), hipMemcpyDeviceToHost, comm->stream)) != hipSuccess) {
    return hipFail(e, "hipMemcpyAsync");
  }
  if ((e = hipStreamSynchronize(comm->stream)) != hipSuccess) {
    return hipFail(e, "hipStreamSynchronize");
  }
  return MMX_OK;
}

int32_t mmx_residual_norms(int32_t batch, const double* final_error, const int32_t* iterations, const int32_t* status, double* norms_dev, void* stream) {
  if (norms_dev == nullptr || batch < 0) {
    return mmx::failWith(MMX_ERR_INVALID_ARGUMENT, "norms is null or batch < 0");
  }
  (void)hipGetLastError();
  hipLaunchKernelGGL(residualNormsKernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), batch, final_error, iterations, status, norms_dev);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    return mmx::failWith(MMX_ERR_DEVICE, std::string("residualNormsKernel: ") + hipGetErrorString(e));
  }
  return MMX_OK;
}

void mmx_comm_destroy(mmx_comm* comm) {
  if (comm != nullptr) {
    if (comm->comm != nullptr && rccl().ok()) {
      (void)hipSetDevice(comm->device);
      (void)rccl().commDestroy(comm->comm);
    }
    if (comm->staging != nullptr) {
      (void)hipSetDevice(comm->device);
      (void)hipFree(comm->staging);
      (void)hipStreamDestroy(comm->stream);
    }
    delete comm;
  }
}

} // extern "C"
