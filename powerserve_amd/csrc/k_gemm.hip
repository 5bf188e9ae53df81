// MFMA mat-mul for batched activations (prefill / tree verify).  Placeholder: returns -1 ("shape not
// covered") so callers fall back to column groups through the GEMV until the MFMA kernel lands.
#include "ps_internal.h"
int psk_gemm(hipStream_t, int, const psk_gemv_args &, ps_act, int, int64_t, int64_t) { return -1; }
