// l1pd.hip -- L1RA: primal-dual interior point per coordinate (ral/l1_irls.cpp:228-468, 851-912).
#include "graph.hpp"
#include "kernels.hpp"

namespace irh {

int l1decode_pd_dev(Graph &, int, const double *, int, double *, int *, int) {
    return IROTAVG_ERR_BAD_ARG;  // placeholder, implemented next
}
int run_l1ra(Graph &, int, double, int *, double *, double *) { return IROTAVG_ERR_BAD_ARG; }

}  // namespace irh
