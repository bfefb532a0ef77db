// Instantiations of the reservoir layer kernel for 256-wide (padded) reservoirs.
#include "reservoir_impl.h"
namespace sgp_res {
int launch_jt16(const ResArgs& a, int nkx, hipStream_t s) { return launch_nkx<16>(a, nkx, s); }
}
