// mas_forms.hpp -- which kernel forms the batch driver (mas_solver.hpp) can launch: THE place where the product build and the
// test-only host emulation part ways.
//   product (hipcc, gfx950):          the hand-written kernels + mas_solver_forms_hip.inc
//   -DMAS_HOST_EMULATION (g++, tests): no kernels; mas_solver_forms_emulation.inc answers "no such form" everywhere
#pragma once

#if defined(MAS_HOST_EMULATION)
#define MAS_SOLVER_FORMS "mas_solver_forms_emulation.inc"
#else
#include "mas_kernels_hip.hpp"
#include "mas_fused_hip.hpp"
#include "mas_dense_chip.hpp"
#define MAS_SOLVER_FORMS "mas_solver_forms_hip.inc"
#endif
