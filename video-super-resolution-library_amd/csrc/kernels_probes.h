// kernels_probes.h -- TIMING PROBES of the filter stage (development builds, -DRAISR_HIP_DEV, only; OUTPUT WRONG).
// Included by device_abi.hip before kernels_filter.h; filter_phase expands the two hooks below where a probe macro is defined.
// What they answered (docs/EXPERIMENTS.md I.4): the ceiling of ANY scheme that shares coefficient rows between pixels, and what the
// window reads cost.  Build:  scripts/build_exp.sh reuse4 -DRAISR_HIP_DEV -DRAISR_EXP_COEF_REUSE=4   /   ... nowin -DRAISR_HIP_DEV -DRAISR_EXP_NO_WINDOW
#pragma once

#ifdef RAISR_EXP_NO_WINDOW
// every step reads step 0's window values, so the compiler keeps them in registers: eight LDS reads per row instead of 128
#define RAISR_PROBE_LDS_F(p, s) (*reinterpret_cast<const float*>((p)))
#endif

#if defined(RAISR_EXP_COEF_REUSE) || defined(RAISR_EXP_NO_WINDOW)
#ifndef RAISR_EXP_COEF_REUSE
#define RAISR_EXP_COEF_REUSE 1
#endif
// On the compiler's own schedule.  RAISR_EXP_COEF_REUSE = n: coefficients are fetched for every n-th step only and reused for the steps
// between -- what a key-chunked stage could gain at most, with its sort, scattered window reads and scattered stores for free.
// (Expands inside filter_phase's row loop: load_q(bucket, q, step), chain, tap[], sH, prow, RAISR_COL, A16 are the names of that scope.)
#define RAISR_PROBE_FILTER_STEPS                                                                           \
        {                                                                                                  \
            float qr[8] = {0, 0, 0, 0, 0, 0, 0, 0};                                                        \
            _Pragma("unroll")                                                                              \
            for (int s = 0; s < 16; s++) {                                                                 \
                if (s % RAISR_EXP_COEF_REUSE == 0) load_q(sH[prow * TW + RAISR_COL(s)], qr, s);               \
                float x[8];                                                                                \
                _Pragma("unroll")                                                                          \
                for (int ch = 0; ch < 8; ch++) x[ch] = RAISR_LDS_F(tap[ch], s);                            \
                A16[s] = chain(x, qr);                                                                     \
            }                                                                                              \
        }
#endif
