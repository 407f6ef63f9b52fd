// engine_internal.h -- shared by the translation units that implement mitdvp::Engine
// (engine.hip, engine_ops.hip, engine_obs.hip, capi.hip): reduction-scratch layout and the
// Krylov template bodies.
#pragma once
#include "engine.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "small_linalg.h"

namespace mitdvp {


constexpr double KRYLOV_EPS = 1e-12;  // _integrator.py:22

// layout of the reduction scratch (units: zc)
constexpr size_t RED_ALPHA = 0;                                   // [MAXK][NPART] zc
constexpr size_t RED_NRM = RED_ALPHA + (size_t)MAXK * NPART;      // [MAXK][NPART] double
constexpr size_t RED_H = RED_NRM + (size_t)MAXK * NPART / 2 + 1;  // [MAXK][MAXK][NPART] zc
constexpr size_t RED_MISC = RED_H + (size_t)MAXK * MAXK * NPART;  // [4][NPART] zc
constexpr size_t RED_TOTAL = RED_MISC + 4 * (size_t)NPART;

}  // namespace mitdvp
