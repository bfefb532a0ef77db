// Instantiations of the reservoir layer kernel for 16-wide (padded) reservoirs.
#include "reservoir_impl.h"
namespace sgp_res {
int launch_jt1(const ResArgs& a, int nkx, hipStream_t s) { return launch_nkx<1>(a, nkx, s); }
}
