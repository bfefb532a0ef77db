// Instantiations of the streamed-weights reservoir kernel (R = 256: 16 output tiles, and
// R = 128 with F = 256, the one 8-tile shape whose weights exceed the LDS), one
// translation unit of its own so that it builds in parallel with reservoir_jt16.hip.
#define SGP_RES_STREAM_TU
#include "reservoir_impl.h"
namespace sgp_res {
template <> int launch_stream_ool<8, 64>(const ResArgs& a, hipStream_t s) { return launch_stream<8, 64>(a, s); }
template <> int launch_stream_ool<16, 1>(const ResArgs&, hipStream_t) { return sgp::fail(SGP_EUNSUP, "unreachable"); }
template <> int launch_stream_ool<16, 2>(const ResArgs&, hipStream_t) { return sgp::fail(SGP_EUNSUP, "unreachable"); }
template <> int launch_stream_ool<16, 4>(const ResArgs& a, hipStream_t s) { return launch_stream<16, 4>(a, s); }
template <> int launch_stream_ool<16, 8>(const ResArgs& a, hipStream_t s) { return launch_stream<16, 8>(a, s); }
template <> int launch_stream_ool<16, 16>(const ResArgs& a, hipStream_t s) { return launch_stream<16, 16>(a, s); }
template <> int launch_stream_ool<16, 32>(const ResArgs& a, hipStream_t s) { return launch_stream<16, 32>(a, s); }
template <> int launch_stream_ool<16, 64>(const ResArgs& a, hipStream_t s) { return launch_stream<16, 64>(a, s); }
}
