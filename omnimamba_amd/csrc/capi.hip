// capi.hip -- ABI bookkeeping: version, thread-local error text, launch-error translation.
#include "omk_common.h"

namespace omk {
char* err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}
int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}
char* kern_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}
void kernels_reset() { kern_buf()[0] = 0; }
void kernels_note(const char* fmt, ...) {
  char* b = kern_buf();
  size_t n = strlen(b);
  if (n + 2 >= 512) return;
  if (n) b[n++] = ';';
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(b + n, 512 - n, fmt, ap);
  va_end(ap);
}
int finish_launch(const char* what) {
#ifndef OMK_EMU
  hipError_t e = hipGetLastError();
  // the library holds gfx950 code objects only: on any other device the launch fails to find a kernel image
  if (e == hipErrorNoBinaryForGpu || e == hipErrorInvalidDeviceFunction || e == hipErrorSharedObjectInitFailed || e == hipErrorInvalidImage)
    return fail(OMK_EARCH, "%s: no kernel image for this device (%s): libomnimamba_hip is built for gfx950 (MI355X) only", what, hipGetErrorString(e));
  if (e != hipSuccess) return fail(OMK_ELAUNCH, "%s: launch failed: %s", what, hipGetErrorString(e));
#endif
  (void)what;
  return OMK_OK;
}
}  // namespace omk

extern "C" {
int omk_abi_version(void) { return OMK_ABI_VERSION; }
const char* omk_last_error(void) { return omk::err_buf(); }
const char* omk_ssd_last_kernels(void) { return omk::kern_buf(); }
size_t omk_sizeof(const char* n) {
#define OMK_SZ(S) if (strcmp(n, #S) == 0) return sizeof(S)
  if (!n) return 0;
  OMK_SZ(OmkTensor); OMK_SZ(OmkAddNormFwd); OMK_SZ(OmkAddNormBwd); OMK_SZ(OmkNormGatedFwd); OMK_SZ(OmkNormGatedBwd);
  OMK_SZ(OmkConv1dFwd); OMK_SZ(OmkConv1dBwd); OMK_SZ(OmkConv1dUpdate); OMK_SZ(OmkStateUpdate); OMK_SZ(OmkSelScanFwd);
  OMK_SZ(OmkSelScanBwd); OMK_SZ(OmkNormLinear); OMK_SZ(OmkLoraAdd); OMK_SZ(OmkLoraUpBwd); OMK_SZ(OmkSsdFwd); OMK_SZ(OmkSsdBwd); OMK_SZ(OmkCrossEntropy); OMK_SZ(OmkSample);
#undef OMK_SZ
  return 0;
}
int omk_is_emulated(void) {
#ifdef OMK_EMU
  return 1;
#else
  return 0;
#endif
}
}
