// capi_internal.h -- the opaque handle of include/mitdvp.h as the translation units of the C surface see it
// (capi.hip, shard.hip).
#pragma once
#include <memory>
#include <string>

#include "engine.h"

struct mitdvp_engine {
  std::unique_ptr<mitdvp::Engine> e;
  std::string err;
  int device = 0;
};
