// placeholder -- replaced by the real solver TU in the next milestone
#include "psfm_internal.h"
psfm_status psfm_solve_frame(psfm_ctx*, const PsfmTrackDims&, const float*, const float*, const float*, const uint8_t*, int,
                             psfm_solve_stats*, hipStream_t) { psfm_set_error("solver not built"); return PSFM_ERR_SOLVER; }
psfm_status psfm_solve_batch(psfm_ctx*, const double*, const double*, const double*, const double*, const float*, int64_t, int,
                             int, double*, psfm_solve_stats*, hipStream_t) { psfm_set_error("solver not built"); return PSFM_ERR_SOLVER; }
