// placeholder until the wave-per-polynomial family lands
#include "kernels.hpp"
namespace camd {
bool wave_supported(const camd_params&) { return false; }
size_t wave_tables_count(uint32_t) { return 1; }
void wave_fill_tables(uint32_t, double2* t) { t[0] = make_double2(0, 0); }
template <typename T>
int wave_bsk_to_fourier(const T*, double2*, const double2*, const camd_params&, hipStream_t) { return fail(CAMD_EINVAL, "no wave kernel"); }
template <typename T>
int wave_launch_pbs(const PbsLaunch<T>&, hipStream_t) { return fail(CAMD_EINVAL, "no wave kernel"); }
template int wave_bsk_to_fourier<uint64_t>(const uint64_t*, double2*, const double2*, const camd_params&, hipStream_t);
template int wave_bsk_to_fourier<uint32_t>(const uint32_t*, double2*, const double2*, const camd_params&, hipStream_t);
template int wave_launch_pbs<uint64_t>(const PbsLaunch<uint64_t>&, hipStream_t);
template int wave_launch_pbs<uint32_t>(const PbsLaunch<uint32_t>&, hipStream_t);
}
