// conv_rows_k1.hip -- the 1x1 instantiations of conv_rows.hip as a translation unit of their own (parallel compilation).
#define ROWS_TU_KS1 1
#include "conv_rows.hip"
