// Instantiations of the reservoir layer kernel for 128-wide (padded) reservoirs.
#include "reservoir_impl.h"
namespace sgp_res {
int launch_jt8(const ResArgs& a, int nkx, hipStream_t s) { return launch_nkx<8>(a, nkx, s); }
}
