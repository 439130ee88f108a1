// (n, m) instantiations, group 1 (see admm_dispatch.hpp).  Adding a pair = adding X(n, m) here.
#define ADMM_GROUP_FN launch_group1
#define ADMM_GROUP_LIST dims_group1
#ifdef ADMM_DEV_DIMS      // development builds (tools/dev_variant.sh): one pair per group, seconds to compile
#define ADMM_GROUP_DIMS(X) X(6, 3)
#else
#define ADMM_GROUP_DIMS(X) X(5, 1) X(5, 2) X(5, 3) X(6, 1) X(6, 2) X(6, 3) X(6, 4) X(6, 6)
#endif
#include "admm_dims_impl.hpp"
