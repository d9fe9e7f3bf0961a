/* TX (DUC) side: state holder.  The kernels are added with the TX path. */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct ddc_tx_state {
    uint32_t n_channels = 0;
};

inline int ddc_tx_state_create(ddc_tx_state *s, uint32_t n_channels, hipStream_t)
{
    s->n_channels = n_channels;
    return 0;
}
inline void ddc_tx_state_destroy(ddc_tx_state *) {}
