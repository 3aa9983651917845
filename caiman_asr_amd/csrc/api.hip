// Library identity + error plumbing for the C-ABI (include/caiman_rnnt.h).
#include "common.h"

namespace caiman {

static thread_local char g_err[512] = {0};

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return CAIMAN_ERR_LAUNCH;
  }
  return CAIMAN_OK;
}

}  // namespace caiman

// 5: + caiman_proj_gemm, caiman_lstm_weight_images, caiman_lstm_grad_deliver, caiman_lstm_resident_xcd_roles; caiman_lstm_prepare accepts R = NULL and gate_layout bit 1
// 6: + caiman_specaug_geometry, caiman_lstm_last_states, caiman_embedding_grad, caiman_slab_accumulate, caiman_wgrad_tn_covers_remainder,
//    caiman_debug_occupy_cus; caiman_lstm_grad_item_t.slabs (was reserved); caiman_wgrad_tn sums the uncovered rows into the last slab;
//    caiman_lstm_wave_fwd / _bwd serve B > 32 at every resident hidden size (32-row slices)
extern "C" int caiman_abi_version(void) { return 6; }
extern "C" const char* caiman_last_error(void) { return caiman::g_err; }
extern "C" int caiman_built_for_gfx950(void) { return 1; }
