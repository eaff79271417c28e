// placeholder, replaced below
#include "hgmm_ctx.h"
using namespace hgmm;
extern "C" int hgmm_tree_build(hgmm_ctx* c, int, double, double, const double*, double, int, double*, double*, double*, int32_t*, int32_t*, double*, int, int*) { return fail(c, HGMM_ERR_STATE, "tree not built yet"); }
extern "C" int hgmm_tree_set_nodes(hgmm_ctx* c, int, const double*, const double*, const double*) { return fail(c, HGMM_ERR_STATE, "tree not built yet"); }
extern "C" int hgmm_tree_set_target(hgmm_ctx* c, const double*, int64_t) { return fail(c, HGMM_ERR_STATE, "tree not built yet"); }
extern "C" int hgmm_tree_reg_estep(hgmm_ctx* c, const double*, const double*, double, double, double*, double*, double*) { return fail(c, HGMM_ERR_STATE, "tree not built yet"); }
extern "C" int hgmm_tree_node_complexity(hgmm_ctx* c, double*) { return fail(c, HGMM_ERR_STATE, "tree not built yet"); }
