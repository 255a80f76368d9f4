// so100_kind.hip -- every kernel instantiation of ONE env kind (compile with -DSO100_KIND=1..6; see so100_kernels.hpp).
#include "so100_kernels.hpp"
#ifndef SO100_KIND
#error "compile with -DSO100_KIND=<1..6>"
#endif
template struct so100::KindOps<SO100_KIND>;
