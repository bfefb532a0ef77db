// Split-fp16 hop, WIDE form: 8 waves x 16 rows x 14 chunks = 448 columns per wave, two waves per SIMD (its A fragments
// take 112 registers).  Serves operators whose rows are longer than the standard form's 224 columns -- the reference's
// full large-scale graphs (config/largescale/sgp_pv.yaml / sgp_cer.yaml with experiments/run_largescale_sgp.py:167-170:
// ~740 / ~495 entries per row): half as many accumulating passes, half the staged rows per result row, at two instead of
// four waves per SIMD (sgp_spmm_split_wide_f32; same kernel source: spmm_split_impl.h).
#define SGP_SPLIT_NW 8
#define SGP_SPLIT_NCH 14
#define SGP_SPLIT_SMAX 896
#define SGP_SPLIT_CR 6
#define SGP_SPLIT_VSTAGE 1
#define SGP_SPLIT_NAME(x) sgp_spmm_split_wide_##x
#include "spmm_split_impl.h"
