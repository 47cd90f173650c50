// Fused MFMA DyGFormer kernel — placeholder until the kernel lands (reports "unsupported" so the
// dispatcher uses the generic path).
#include "dygformer_layout.h"

namespace dygnn {

size_t fused_packed_floats(const Dims&) { return 0; }
bool fused_supported(const Dims&) { return false; }

int pack_fused(const Dims&, const PackedLayout&, const dygnn_dygformer_weights*, float*, hipStream_t) { return DYGNN_OK; }

int forward_fused(const Dims&, const PackedLayout&, const dygnn_dygformer_weights*, const float*, const dygnn_csr*, const float*,
                  const float*, const int64_t*, const int64_t*, const double*, int64_t, float*, float*, char*,
                  const WorkspaceLayout&, const dygnn_dygformer_taps*, hipStream_t) {
    set_error("fused kernel not built");
    return DYGNN_E_UNSUPPORTED;
}

}  // namespace dygnn
