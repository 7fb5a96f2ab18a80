// stubs.hip -- entry points declared in include/omk.h whose kernels are not written yet return OMK_EUNSUPPORTED.
// (shrinks to nothing as the kernels land)
#include "omk_common.h"
#define OMK_STUB(name, T) extern "C" int name(const T*, omk_stream) { return omk::fail(OMK_EUNSUPPORTED, #name ": not implemented yet"); }
OMK_STUB(omk_selective_state_update, OmkStateUpdate)
OMK_STUB(omk_selective_scan_fwd, OmkSelScanFwd)
OMK_STUB(omk_selective_scan_bwd, OmkSelScanBwd)
OMK_STUB(omk_ssd_scan_fwd, OmkSsdFwd)
OMK_STUB(omk_ssd_scan_bwd, OmkSsdBwd)
extern "C" size_t omk_ssd_scan_fwd_workspace_bytes(const OmkSsdFwd*) { return 0; }
extern "C" size_t omk_ssd_scan_bwd_workspace_bytes(const OmkSsdBwd*) { return 0; }
