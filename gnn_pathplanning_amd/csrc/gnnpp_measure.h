// Measurement-only tuning keys of -DGNNPP_MEASURE builds (tools/ab_bench.py builds
// libgnnpp_measure.so next to the product library).  These knobs make results WRONG by construction
// (kernel phases are skipped), which is why they are not part of include/gnnpp.h and do not exist
// in libgnnpp.so: there gnnpp_set_tuning() rejects the keys with GNNPP_ERR_ARG.
#ifndef GNNPP_MEASURE_H_
#define GNNPP_MEASURE_H_
#define GNNPP_TUNE_FILTER_ABLATE   3  /* bit mask of filter phases to skip: 1 shifts, 2 contraction,
                                         4 GSO staging, 8 epilogue; 0 = the real kernel            */
#define GNNPP_TUNE_ENCODER_STOP    4  /* split-f16 encoder: return after phase 1 staging, 2 L0, 3 L1,
                                         4 L2, 5 L3, 6 L4; 0 = whole encoder                       */
#endif
