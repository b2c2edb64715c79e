// NIST P-256 kernels and launchers: ops_nist.inc instantiated for CurveP256.
#define ECGPU_NIST_CURVE CurveP256
#define ECGPU_NIST_OPS_FN ecgpu_ops_p256
#define ECGPU_NIST_MSM_FN ecgpu_msm_p256
#include "ops_nist.inc"
