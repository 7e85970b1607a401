#include "handle.hpp"
namespace lsr {
float nn_pick_cell(size_t, const lsr_handle_s*) { return 0.5f; }
int nn_build_hash(const DeviceCloud&, float, HashGridDev&, BuildScratch&, hipStream_t) { set_last_error("NN not implemented yet"); return LSR_ERR_NOT_IMPLEMENTED; }
int nn_fitness_score(const DeviceCloud&, const float*, const HashGridDev&, double, double*, BuildScratch&, DevBuf<float>&, hipStream_t) { return LSR_ERR_NOT_IMPLEMENTED; }
int nn_search_host(const DeviceCloud&, const float*, const HashGridDev&, int32_t*, float*, BuildScratch&, DevBuf<float>&, hipStream_t) { return LSR_ERR_NOT_IMPLEMENTED; }
}
