// Fused read-set backward for gfx950 (placeholder until the kernel lands: fails loudly, never falls back).
#include "pmt_device.hpp"

extern "C" int pmt_backward(const PmtModel* model_host, const PmtModel* model_dev, const float* theta, const float* phi,
                            const float* packed, const PmtBatch* batch, const PmtOutputGrads* dout, const float* stash,
                            float* grad_theta, float* grad_phi, float* grad_variant_embed, void* stream) {
    return PMT_E_UNSUPPORTED;
}
