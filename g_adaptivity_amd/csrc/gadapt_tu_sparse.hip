// gadapt_tu_sparse.hip - generic CSR-row message-passing primitives (spmm, sddmm, edge softmax, ...) for the conv variants besides
// GRAND / GRAND_plus that get_conv builds (src/GNN.py:108-124).  One translation unit of libgadapt_hip.so (see gadapt_internal.h).
#include "gadapt_internal.h"
#include "gadapt_sparse.inc"
