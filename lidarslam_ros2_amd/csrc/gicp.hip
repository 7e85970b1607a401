#include "handle.hpp"
namespace lsr {
int gicp_align(lsr_handle_s*, const float*, float*, lsr_result*) { set_last_error("GICP not implemented yet"); return LSR_ERR_NOT_IMPLEMENTED; }
int gicp_get_covariances(lsr_handle_s*, int, double*) { return LSR_ERR_NOT_IMPLEMENTED; }
}
