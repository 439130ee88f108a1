// (n, m) instantiations, group 0 (see admm_dispatch.hpp).  Adding a pair = adding X(n, m) here.
#define ADMM_GROUP_FN launch_group0
#define ADMM_GROUP_LIST dims_group0
#ifdef ADMM_DEV_DIMS      // development builds (tools/dev_variant.sh): one pair per group, seconds to compile
#define ADMM_GROUP_DIMS(X) X(2, 1)
#else
#define ADMM_GROUP_DIMS(X) X(1, 1) X(2, 1) X(2, 2) X(3, 1) X(3, 2) X(3, 3) X(4, 1) X(4, 2) X(4, 3) X(4, 4)
#endif
#include "admm_dims_impl.hpp"
