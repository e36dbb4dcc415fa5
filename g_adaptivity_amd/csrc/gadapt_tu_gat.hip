// gadapt_tu_gat.hip - fused GAT_plus block (src/GRAND_plus.py:386-416 inside the layer loop of src/GNN.py:273-296).  One translation
// unit of libgadapt_hip.so (see gadapt_internal.h).
#include "gadapt_internal.h"
#include "gadapt_gat.inc"
