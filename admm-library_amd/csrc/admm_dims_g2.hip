// (n, m) instantiations, group 2 (see admm_dispatch.hpp).  Adding a pair = adding X(n, m) here.
#define ADMM_GROUP_FN launch_group2
#define ADMM_GROUP_LIST dims_group2
#ifdef ADMM_DEV_DIMS      // development builds (tools/dev_variant.sh): one pair per group, seconds to compile
#define ADMM_GROUP_DIMS(X) X(8, 4)
#else
#define ADMM_GROUP_DIMS(X) X(7, 2) X(7, 3) X(8, 2) X(8, 3) X(8, 4) X(9, 3)
#endif
#include "admm_dims_impl.hpp"
