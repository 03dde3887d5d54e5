// Entry points declared in include/dfepe.h whose kernels are not written yet: they refuse loudly.
#include "dfepe_common.h"

extern "C" int dfepe_cheirality(const float*, const float*, const float*, int, int, float, float*, int*, int*, void*) {
  return DFEPE_ERR_UNSUPPORTED;
}
extern "C" int dfepe_epi_metrics(int, const float*, const float*, const float*, int, int, float, float, float*, void*) {
  return DFEPE_ERR_UNSUPPORTED;
}
