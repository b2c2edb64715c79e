// NIST P-384 kernels and launchers: ops_nist.inc instantiated for CurveP384.
#define ECGPU_NIST_CURVE CurveP384
#define ECGPU_NIST_OPS_FN ecgpu_ops_p384
#define ECGPU_NIST_MSM_FN ecgpu_msm_p384
#include "ops_nist.inc"
