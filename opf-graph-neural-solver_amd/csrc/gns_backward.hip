// placeholder: the fused reverse pass is added next
#include "gns_kernels.h"
extern "C" int gns_backward(const gns_config*, const void*, const float*, int64_t, const void*, size_t, const float*, const float*,
                            const float*, const float*, float*, void*, size_t, void*) {
  return GNS_EUNSUPPORTED;
}
