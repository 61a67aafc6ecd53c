// placeholder until the MFMA kernels land (next commit)
#include "blobnet.h"
int blobnet_prepare_mfma(covahip_ctx *, covahip_blobnet *, const float *) { return COVAHIP_OK; }
void blobnet_release_mfma(covahip_ctx *, covahip_blobnet *) {}
int blobnet_forward_mfma(covahip_ctx *ctx, covahip_blobnet *m, const uint8_t *d_stack, int batch, float *d_logits,
                         uint8_t *d_mask) {
    return blobnet_forward_naive(ctx, m, d_stack, batch, d_logits, d_mask);
}
