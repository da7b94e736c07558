// C-ABI entry points of the BFVI sweep: argument checks, then kernel-family dispatch.
//   z_dim, h_dim <= 32, K <= 32 : register-chained f32 MFMA kernels (sweep_mfma.hip)
//   z_dim = h_dim = 256 with a fragment pack (gtf_frag) : wide MFMA kernels (sweep_wide.hip;
//                                 stand-alone z_next: trans_wide.hip)
//   everything else             : generic LDS-tiled fp32 kernels    (sweep_simt.hip)
// MDMM_FORCE_GENERIC=1 in the environment pins the generic family (A/B runs, cross-checks).
#include <stdlib.h>
#include "sweep_internal.h"

static bool force_generic() {
  const char* v = getenv("MDMM_FORCE_GENERIC");
  return v && v[0] == '1';
}

// the fused KL term exists in the wide K = 1 kernels only: any other route would drop it silently
static bool kld_lost(const mdmm_sweep_t* a) {
  if (!(a->kld_out || a->kld_scale_dev)) return false;
  if (force_generic()) return true;
  return !mdmm_sweep_kld_fused(a);
}

extern "C" int mdmm_bfvi_sweep_fwd(const mdmm_sweep_t* args, void* stream) {
  int rc = mdmm_sweep_check_args(args, 0);
  if (rc) return rc;
  if (kld_lost(args)) return MDMM_E_ARG;
  if (!force_generic()) {
    if (args->trans_only && args->gtf_frag) {
      rc = mdmm_wide_trans(args, 0, (hipStream_t)stream);
      if (rc != MDMM_UNSUPPORTED) return rc;
    }
    rc = mdmm_wide_sweep_fwd(args, (hipStream_t)stream);
    if (rc != MDMM_UNSUPPORTED) return rc;
    rc = mdmm_mfma_sweep_fwd(args, (hipStream_t)stream);
    if (rc != MDMM_UNSUPPORTED) return rc;
  }
  return mdmm_simt_sweep_fwd(args, (hipStream_t)stream);
}

extern "C" int mdmm_bfvi_sweep_bwd(const mdmm_sweep_t* args, void* stream) {
  int rc = mdmm_sweep_check_args(args, 1);
  if (rc) return rc;
  if (kld_lost(args)) return MDMM_E_ARG;
  if (!force_generic()) {
    if (args->trans_only && args->gtf_frag) {
      rc = mdmm_wide_trans(args, 1, (hipStream_t)stream);
      if (rc != MDMM_UNSUPPORTED) return rc;
    }
    if (args->wide_ws && mdmm_wide_bwd_supported(args)) return mdmm_wide_sweep_bwd(args, (hipStream_t)stream);
    rc = mdmm_mfma_sweep_bwd(args, (hipStream_t)stream);
    if (rc != MDMM_UNSUPPORTED) return rc;
  }
  return mdmm_simt_sweep_bwd(args, (hipStream_t)stream);
}

extern "C" int mdmm_sweep_bwd_mode(const mdmm_sweep_t* args) {
  if (!args || force_generic()) return 0;
  if (mdmm_wide_bwd_supported(args)) return 2;
  return mdmm_mfma_bwd_supported(args);
}

extern "C" int mdmm_sweep_dw_width(int D, int H) { return mdmm_mfma_dw_width(D, H); }

extern "C" int64_t mdmm_sweep_dw_rows(const mdmm_sweep_t* args) {
  if (!args || force_generic()) return 0;
  if (mdmm_wide_bwd_supported(args)) return 1;
  return mdmm_mfma_dw_rows(args);
}
