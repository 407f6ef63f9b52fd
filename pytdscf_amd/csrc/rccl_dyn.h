// rccl_dyn.h -- RCCL entry points resolved at run time (dlopen), so that the single-GPU
// library has no load-time dependency on librccl.  Types and enums come from RCCL's own header.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <string>

#include "common.h"

namespace mitdvp {

struct RcclApi {
  decltype(&ncclGetUniqueId) get_unique_id = nullptr;
  decltype(&ncclCommInitRank) comm_init_rank = nullptr;
  decltype(&ncclCommDestroy) comm_destroy = nullptr;
  decltype(&ncclAllGather) all_gather = nullptr;
  decltype(&ncclAllReduce) all_reduce = nullptr;
  // point-to-point halo of the site-sharded sweep (shard.hip): grouped ncclSend / ncclRecv between chain neighbours
  decltype(&ncclSend) send = nullptr;
  decltype(&ncclRecv) recv = nullptr;
  decltype(&ncclGroupStart) group_start = nullptr;
  decltype(&ncclGroupEnd) group_end = nullptr;
  decltype(&ncclGetErrorString) error_string = nullptr;

  static const RcclApi& get() {
    static const RcclApi api = load();
    return api;
  }

 private:
  static RcclApi load() {
    void* h = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (h) break;
    }
    if (!h) throw HipError(std::string("cannot load librccl: ") + dlerror());
    RcclApi a;
    auto sym = [&](const char* n) {
      void* p = dlsym(h, n);
      if (!p) throw HipError(std::string("librccl lacks ") + n);
      return p;
    };
    a.get_unique_id = reinterpret_cast<decltype(a.get_unique_id)>(sym("ncclGetUniqueId"));
    a.comm_init_rank = reinterpret_cast<decltype(a.comm_init_rank)>(sym("ncclCommInitRank"));
    a.comm_destroy = reinterpret_cast<decltype(a.comm_destroy)>(sym("ncclCommDestroy"));
    a.all_gather = reinterpret_cast<decltype(a.all_gather)>(sym("ncclAllGather"));
    a.all_reduce = reinterpret_cast<decltype(a.all_reduce)>(sym("ncclAllReduce"));
    a.send = reinterpret_cast<decltype(a.send)>(sym("ncclSend"));
    a.recv = reinterpret_cast<decltype(a.recv)>(sym("ncclRecv"));
    a.group_start = reinterpret_cast<decltype(a.group_start)>(sym("ncclGroupStart"));
    a.group_end = reinterpret_cast<decltype(a.group_end)>(sym("ncclGroupEnd"));
    a.error_string = reinterpret_cast<decltype(a.error_string)>(sym("ncclGetErrorString"));
    return a;
  }
};

inline void rccl_check(ncclResult_t r, const char* what) {
  if (r != ncclSuccess) throw HipError(std::string(what) + ": " + RcclApi::get().error_string(r));
}

}  // namespace mitdvp
