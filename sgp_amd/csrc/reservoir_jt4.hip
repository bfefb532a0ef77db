// Instantiations of the reservoir layer kernel for 64-wide (padded) reservoirs.
#include "reservoir_impl.h"
namespace sgp_res {
int launch_jt4(const ResArgs& a, int nkx, hipStream_t s) { return launch_nkx<4>(a, nkx, s); }
}
