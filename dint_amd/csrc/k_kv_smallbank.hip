// k_kv_smallbank.hip -- the smallbank instantiation of the kv pass kernels (k_kv_dev.h); host side in k_kv.hip.
#include "k_kv_dev.h"

template void launch_kv_passes<DINT_WL_SMALLBANK>(kv_multi_args &, uint32_t, uint32_t, hipStream_t, hipEvent_t *, const dint_kv_knobs &, bool,
                                     const kv_multi_args *, uint32_t);
template int kv_piece_residency<DINT_WL_SMALLBANK>(int);
