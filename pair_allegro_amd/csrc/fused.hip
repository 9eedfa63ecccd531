// Fused MFMA path -- placeholder until the register-chain kernel lands.
#include "engine.h"

namespace ahip {
bool fused_model_supported(const Model &, std::string *why) { if (why) *why = "fused kernels not built yet"; return false; }
bool fused_run(Model &, const ComputeArgs &, std::string *why) { if (why) *why = "fused kernels not built yet"; return false; }
void fused_free(Model &) {}
}  // namespace ahip
