// Entry points declared in include/dfepe.h whose kernels are not written yet: they refuse loudly.
#include "dfepe_common.h"

extern "C" int dfepe_cheirality(const float*, const float*, const float*, int, int, float, float*, int*, int*, void*) {
  return DFEPE_ERR_UNSUPPORTED;
}
extern "C" int dfepe_epi_metrics(int, const float*, const float*, const float*, int, int, float, float, float*, void*) {
  return DFEPE_ERR_UNSUPPORTED;
}
extern "C" int dfepe_w8pt_bwd(const float*, const float*, const float*, int, int, unsigned, float, float, float, const float*,
                              const float*, const float*, const float*, const float*, float*, void*) {
  return DFEPE_ERR_UNSUPPORTED;
}
extern "C" int dfepe_floss_fwd(const float*, int, int, const float*, const float*, int, const float*, const float*,
                               const float*, int, float, float*, float*, void*) {
  return DFEPE_ERR_UNSUPPORTED;
}
extern "C" int dfepe_floss_bwd(const float*, int, int, const float*, const float*, int, const float*, const float*,
                               const float*, int, float, const float*, const float*, float*, void*) {
  return DFEPE_ERR_UNSUPPORTED;
}
extern "C" int dfepe_pose_fwd(const float*, int, int, const float*, const float*, const float*, float*, float*, float*,
                              float*, int*, void*) {
  return DFEPE_ERR_UNSUPPORTED;
}
extern "C" int dfepe_pose_bwd(const float*, int, int, const float*, const float*, const float*, const float*, float*, void*) {
  return DFEPE_ERR_UNSUPPORTED;
}
