// ssd_mfma.hip -- MFMA chunked SSD scan (placeholder until the kernel lands: everything falls to the generic scan)
#include "ssd_scan.h"
namespace omk {
int ssd_mfma_launch(const GScan&, omk_stream) { return OMK_EUNSUPPORTED; }
}
