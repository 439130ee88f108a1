// (n, m) instantiations, group 3 (see admm_dispatch.hpp).  Adding a pair = adding X(n, m) here.
#define ADMM_GROUP_FN launch_group3
#define ADMM_GROUP_LIST dims_group3
#ifdef ADMM_DEV_DIMS      // development builds (tools/dev_variant.sh): one pair per group, seconds to compile
#define ADMM_GROUP_DIMS(X) X(12, 6)
#else
#define ADMM_GROUP_DIMS(X) X(10, 2) X(10, 4) X(12, 3) X(12, 4) X(12, 6)
#endif
#include "admm_dims_impl.hpp"
