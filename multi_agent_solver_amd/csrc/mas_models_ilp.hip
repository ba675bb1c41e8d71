// mas_models_ilp.hip -- second translation unit of libmas_hip.so: the models whose kernels are compiled with the machine
// scheduler's max-ILP strategy (see make_solver in mas_solver.hpp).  Same sources, one more compiler option.
#define MAS_TU_MODELS_ILP 1
#include "mas_solver.hpp"
