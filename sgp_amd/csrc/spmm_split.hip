// Split-fp16 hop, standard form: 16 waves x 16 rows x 7 chunks of 32 columns (sgp_spmm_split_f32; kernel and commentary:
// spmm_split_impl.h).
#define SGP_SPLIT_NW 16
#define SGP_SPLIT_NCH 7
#define SGP_SPLIT_SMAX 768
#define SGP_SPLIT_CR 3
#define SGP_SPLIT_NAME(x) sgp_spmm_split_##x
#include "spmm_split_impl.h"
