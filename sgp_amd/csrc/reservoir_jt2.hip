// Instantiations of the reservoir layer kernel for 32-wide (padded) reservoirs.
#include "reservoir_impl.h"
namespace sgp_res {
int launch_jt2(const ResArgs& a, int nkx, hipStream_t s) { return launch_nkx<2>(a, nkx, s); }
}
